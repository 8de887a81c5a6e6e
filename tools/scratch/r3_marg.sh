OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "marg or sequence or sliding or window" 2>&1 | tail -2
SVIN_MARG_TIMING=1 timeout 300 python tools/margtime.py 2>&1 | grep "\[marg\]\|\[pack\]\|per frame\|ms" | tail -8
