OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q -s > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
grep -n "exact chain\|vs exact\|passed\|failed\|FAILED\|Error" $OUT/pytest.log | tail -20
