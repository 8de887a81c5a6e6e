OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_abi.py -x -q -k "config3 or rig_v2 or euroc or wide or ll or sonar" 2>&1 | grep "passed\|failed"
SVIN_BA_LIB=$PWD/build/variants/llt.so timeout 200 python tools/cfg3time.py 2>&1 | grep "ll wave 0\|ll backsub"
