# (gpurun helper) launch timeline of a config-#3 iteration (both streams)
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/c3tl -o w -- python $GRAFT_REPO_ROOT/tools/cfg3time.py > /tmp/c3tl.log 2>&1
python - /tmp/c3tl/w_results.db <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/r06/cfg3_timeline.txt
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end, queue_id from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "k_post_solve" in r[0]][-3]
rows = rows[idx - 14:idx + 14]
t0 = rows[0][1]
for name, s, e, qd in rows:
    print("%-40s q%-3s %9.1f %9.1f %8.2f" % (name.replace("svin::","").replace("void ","")[:40], qd, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
PY
