# A/B of the solve phase: committed baseline variant vs the product build, alternating, plus the coarse timeline
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
for i in 1 2 3; do
  echo -n "base "; SVIN_BA_LIB=$PWD/build/variants/base.so timeout 120 python tools/choltime.py 2>&1 | tail -1
  echo -n "new  "; timeout 120 python tools/choltime.py 2>&1 | tail -1
done
SVIN_BA_LIB=$PWD/build/variants/choltiming.so SVIN_CHOL_TIMING=1 timeout 300 python tools/choltime.py > $OUT/choltime.txt 2>&1
grep "chol cycles\|back:" $OUT/choltime.txt
