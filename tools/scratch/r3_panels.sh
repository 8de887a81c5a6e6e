OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "wide or config4 or panel or sharded or rccl" 2>&1 | tail -2
for i in 1 2; do
  echo -n "base "; SVIN_WIDE_BENCH=1 SVIN_BA_LIB=$PWD/build/variants/base.so timeout 300 python tools/widetime.py 2>&1 | tail -2 | tr '\n' ' '; echo
  echo -n "new  "; SVIN_WIDE_BENCH=1 timeout 300 python tools/widetime.py 2>&1 | tail -2 | tr '\n' ' '; echo
done
