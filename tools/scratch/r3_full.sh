# full GPU test-suite + rocprof kernel table + chol timeline
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
SVIN_BA_LIB=$PWD/build/variants/choltiming.so SVIN_CHOL_TIMING=1 timeout 300 python tools/choltime.py > $OUT/choltime.txt 2>&1
tail -14 $OUT/choltime.txt
bash tools/r3_prof.sh $1
