# config #3 (d = 270, k_chol_solve_ll): parity tests that use it, kernel times of the committed baseline variant and the product build
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_abi.py -x -q -k "config3 or rig_v2 or euroc or wide or ll or sonar" 2>&1 | tail -3
for i in 1 2; do
  echo -n "base "; SVIN_BA_LIB=$PWD/build/variants/base.so timeout 200 python tools/cfg3time.py 2>&1 | tail -1
  echo -n "new  "; timeout 200 python tools/cfg3time.py 2>&1 | tail -1
done
timeout 200 python tools/cfg3time.py 2>&1 | tail -3 | head -2
