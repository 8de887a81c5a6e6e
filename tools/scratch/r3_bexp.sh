OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
for e in 1 2 4 7; do echo "exp $e"; SVIN_BA_LIB=$PWD/build/variants/bexp$e.so SVIN_CHOL_TIMING=1 timeout 300 python tools/choltime.py 2>&1 | grep "chol cycles\|back: w0 far\|back: w0 prod"; done
