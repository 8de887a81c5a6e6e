# round-3 GPU call 1: full GPU test-suite, the headline bench, in-kernel timeline of k_chol_solve_lds (before)
OUT=$PWD/gpurun_out/r3a; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
SVIN_BA_LIB=$PWD/build/variants/choltiming.so SVIN_CHOL_TIMING=1 timeout 300 python tools/choltime.py > $OUT/choltime.txt 2>&1
cat $OUT/choltime.txt | tail -12
