# kernel-trace of the headline command; summary table under gpurun_out/$1
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o b -- python $REPO/bench.py --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
python $REPO/tools/prof_summary.py $OUT/trace/b_results.db > $OUT/bench_kernel_stats.txt 2>&1
cd $REPO
rm -rf $OUT/trace
head -14 $OUT/bench_kernel_stats.txt
