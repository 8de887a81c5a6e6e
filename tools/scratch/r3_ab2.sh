# A/B of the whole iteration (config #2): committed baseline variant vs the product build, bench.py without extras
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
for i in 1 2; do
  SVIN_BA_LIB=$PWD/build/variants/base.so timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base', d['value'], d['ms_per_step'])"
  timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new ', d['value'], d['ms_per_step'])"
done
