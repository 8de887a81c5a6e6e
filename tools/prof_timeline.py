"""Timeline of the last N kernel launches of a rocprofv3 --kernel-trace database: start offset, duration and the gap to the
previous kernel's end (us).  usage: prof_timeline.py <results.db> [N=40] [name filter for the anchor kernel]"""
import sqlite3, sys
db = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end from kernels order by start"))
rows = rows[-n:]
t0 = rows[0][1]; prev = None
print("%-60s %10s %9s %9s" % ("kernel", "start_us", "dur_us", "gap_us"))
for name, s, e in rows:
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    print("%-60s %10.1f %9.2f %9.2f" % (name[:60], (s - t0) / 1e3, (e - s) / 1e3, gap))
    prev = e
