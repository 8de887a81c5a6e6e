# round-3 GPU call: smoke, GPU test-suite, bench, in-kernel timeline of k_chol_solve_lds
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc $?"; tail -3 $OUT/smoke.log
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
SVIN_BA_LIB=$PWD/build/variants/choltiming.so SVIN_CHOL_TIMING=1 timeout 300 python tools/choltime.py > $OUT/choltime.txt 2>&1
tail -6 $OUT/choltime.txt
python - $OUT/bench.json <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("value", d["value"], "ms_per_step", d["ms_per_step"], "config3", d.get("config3",{}).get("value"), "config4", d.get("config4_single_gpu",{}).get("value"))
except Exception as e:
    print("bench parse failed", e)
P
