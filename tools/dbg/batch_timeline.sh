# (gpurun helper) timeline of the batched solve (lanes given by $1, B = 16)
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp
for l in $1; do
SVIN_BATCH_LANES=$l rocprofv3 --kernel-trace --stats -d /tmp/bt$l -o b -- python $GRAFT_REPO_ROOT/tools/batchtime.py 16 > /tmp/log$l 2>&1
grep "aggregate\|x_one" /tmp/log$l
python - /tmp/bt$l/b_results.db <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/r06/e_timeline_$l.txt
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print(cols)
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = list(c.execute("select name, start, end, %s from kernels order by start" % q))
rows = rows[-150:]
t0 = rows[0][1]
for name, s, e, qd in rows:
    print("%-44s q%-4s %9.1f %9.1f %8.2f" % (name.replace("svin::","").replace("void ","")[:44], qd, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
PY
done
