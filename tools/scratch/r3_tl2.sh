OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/trace -o b -- python $REPO/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/trace.log
python $REPO/tools/prof_timeline.py $OUT/trace/b_results.db 100000 > $OUT/timeline.txt 2>&1
cd $REPO; rm -rf $OUT/trace
