"""Summarise a `rocprofv3 --pmc <COUNTER> --kernel-trace` sqlite database: per kernel, calls and mean counter value.
usage: python tools/pmc_summary.py <results.db> [kernel-substring]"""
import collections
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    ik, ic, iv = cols.index("kernel_name"), cols.index("counter_name"), cols.index("value")
    agg = collections.defaultdict(list)
    for r in db.execute("select * from counters_collection"):
        if pat in r[ik]:
            agg[(r[ik][:90], r[ic])].append(r[iv])
    print("%-90s %-12s %6s %16s %16s %16s" % ("kernel", "counter", "calls", "mean", "min", "max"))
    for (k, c), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("%-90s %-12s %6d %16.3f %16.3f %16.3f" % (k, c, len(v), sum(v) / len(v), min(v), max(v)))


if __name__ == "__main__":
    main()
