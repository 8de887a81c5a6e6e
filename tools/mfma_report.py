"""Matrix-pipe evidence for the solver kernels from rocprofv3 counter passes (tools/run_r02_profiles.sh).
usage: python tools/mfma_report.py <pmc1.db> <pmc3.db>
  pmc1: SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES   (+ kernel trace)
  pmc3: GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_MFMA                                               (+ kernel trace)
Per kernel (means per launch): duration, workgroups' waves, MFMA instructions, f64 flops on the matrix pipe (MOPS x 512),
matrix-pipe busy cycles (64 per v_mfma_f64_16x16x4_f64), achieved TFLOP/s, and the busy fraction of the matrix pipes
 (a) of the whole chip (256 CUs x 4 pipes x kernel cycles) and (b) of the CUs the kernel can occupy at all
     (min(256, waves / 4) CUs -- a single-workgroup kernel owns one CU).
Kernel cycles = duration x 2.4 GHz (the counters' own clock domain is not uniform across blocks: GRBM_GUI_ACTIVE is summed
over the 8 XCDs; it is printed for reference)."""
import collections
import sqlite3
import sys


def counters(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    ik, ic, iv = cols.index("kernel_name"), cols.index("counter_name"), cols.index("value")
    agg = collections.defaultdict(list)
    for r in db.execute("select * from counters_collection"):
        agg[(r[ik], r[ic])].append(r[iv])
    dur = {}
    for name, n, avg in db.execute("select name, count(*), avg(end-start) from kernels group by name"):
        dur[name] = (n, avg)
    return {k: sum(v) / len(v) for k, v in agg.items()}, dur


def main():
    c1, d1 = counters(sys.argv[1])
    c3, d3 = counters(sys.argv[2])
    names = sorted({k for k, _ in c1} | {k for k, _ in c3}, key=lambda k: -d1.get(k, (0, 0))[1] * d1.get(k, (0, 0))[0])
    print("%-52s %6s %9s %8s %9s %12s %12s %9s %10s %10s %14s" % ("kernel", "calls", "avg us", "waves", "MFMAs", "f64 flops", "busy cycles", "TFLOP/s",
                                                                   "chip frac", "own-CU frac", "GRBM_GUI_ACTIVE"))
    for k in names:
        if "svin::" not in k:
            continue
        n, avg = d1.get(k, d3.get(k, (0, 0.0)))
        mops = c1.get((k, "SQ_INSTS_VALU_MFMA_MOPS_F64"), 0.0)
        busy = c1.get((k, "SQ_VALU_MFMA_BUSY_CYCLES"), 0.0)
        waves = c3.get((k, "SQ_WAVES"), 0.0)
        insts = c3.get((k, "SQ_INSTS_MFMA"), 0.0)
        gui = c3.get((k, "GRBM_GUI_ACTIVE"), 0.0)
        cyc = avg * 2.4          # ns x 2.4 GHz
        flops = mops * 512.0
        cus = max(1.0, min(256.0, waves / 4.0))
        short = k.replace("svin::", "").split("(")[0]
        print("%-52s %6d %9.2f %8.0f %9.0f %12.0f %12.0f %9.4f %10.5f %10.4f %14.0f" %
              (short[:52], n, avg / 1e3, waves, insts, flops, busy, flops / max(avg, 1e-9) / 1e3, busy / max(cyc * 4 * 256, 1.0),
               busy / max(cyc * 4 * cus, 1.0), gui))


if __name__ == "__main__":
    main()
