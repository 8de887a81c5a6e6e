OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/trace -o b -- python $REPO/tools/margtime.py > $OUT/marg.log 2>&1
python $REPO/tools/prof_summary.py $OUT/trace/b_results.db > $OUT/kernel_stats.txt 2>&1
cd $REPO; rm -rf $OUT/trace
head -16 $OUT/kernel_stats.txt
