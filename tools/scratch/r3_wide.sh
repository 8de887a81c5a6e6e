OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
SVIN_WIDE_BENCH=${SVIN_WIDE_BENCH:-0} rocprofv3 --kernel-trace -d $OUT/trace -o b -- python $REPO/tools/widetime.py > $OUT/wide.log 2>&1
python $REPO/tools/prof_timeline.py $OUT/trace/b_results.db 400 > $OUT/timeline.txt 2>&1
python $REPO/tools/prof_summary.py $OUT/trace/b_results.db > $OUT/kernel_stats.txt 2>&1
cd $REPO; rm -rf $OUT/trace
tail -4 $OUT/wide.log
