# in-kernel timeline of k_chol_solve_lds (timing variant), parity smoke + quick bench of the product build
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "config2 or config3 or full_size or landmark_quality or determin" 2>&1 | tail -2
SVIN_BA_LIB=$PWD/build/variants/choltiming.so SVIN_CHOL_TIMING=1 timeout 300 python tools/choltime.py > $OUT/choltime.txt 2>&1
tail -14 $OUT/choltime.txt
timeout 300 python tools/choltime.py 2>&1 | tail -1
timeout 600 python bench.py --no-cpu-baseline --no-extras > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json,sys
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print('value', d['value'], 'ms_per_step', d['ms_per_step'])"
