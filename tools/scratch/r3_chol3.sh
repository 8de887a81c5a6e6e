OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
SVIN_BA_LIB=$PWD/build/variants/choltiming.so SVIN_CHOL_TIMING=1 timeout 300 python tools/choltime.py > $OUT/choltime.txt 2>&1
tail -15 $OUT/choltime.txt
bash tools/r3_prof.sh $1 | head -3
