OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python - $OUT/bench.json <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "pcie", d["config"].get("pcie_inclusive"))
for k in ("config3","config4_single_gpu"):
    r=d.get(k,{}); print(k, r.get("value"), json.dumps(r.get("roofline"))[:600])
print("sliding", json.dumps(d.get("sliding_window"))[:1500])
print("config5", d.get("config5"), d.get("config5_6dof"))
print("cpu", d.get("cpu_baseline",{}).get("threads"))
P
SVIN_PACK_TIMING=1 SVIN_MARG_TIMING=1 timeout 300 python tools/margtime.py 2>&1 | grep "policy\|pack\]" | tail -6
