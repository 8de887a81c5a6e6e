OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
for v in choltiming $2; do echo "variant $v"; SVIN_BA_LIB=$PWD/build/variants/$v.so SVIN_CHOL_TIMING=1 timeout 300 python tools/choltime.py 2>&1 | grep "chol cycles\|per wave\|w0 pivot start\|w0 past the wait\|w0 at look"; done
