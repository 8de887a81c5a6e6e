OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
SVIN_BA_LIB=$PWD/build/variants/choltiming.so SVIN_CHOL_TIMING=1 timeout 300 python tools/choltime.py > $OUT/choltime.txt 2>&1
tail -15 $OUT/choltime.txt
