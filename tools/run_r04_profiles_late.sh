# round 4, late set (after the speed / bias chain elimination): the bench line, config #4 and the two sliding windows per kernel.
# Run on the GPU box through gpurun from the repo root; outputs under gpurun_out/r04late.
set -x
OUT=$PWD/gpurun_out/r04late; mkdir -p $OUT
REPO=$PWD
python bench.py > $OUT/bench_v3.json 2> $OUT/bench_v3.err
cd /tmp && export TMPDIR=/tmp
SVIN_WIDE_BENCH=1 rocprofv3 --kernel-trace --stats -d $OUT/c4t -o b -- python $REPO/tools/widetime.py > $OUT/c4_trace.log 2>&1
python $REPO/tools/prof_summary.py $OUT/c4t/b_results.db > $OUT/config4_bench_kernel_stats.txt 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/slide -o b -- python $REPO/tools/slidetime.py --short > $OUT/slide.log 2>&1
python $REPO/tools/prof_summary.py $OUT/slide/b_results.db > $OUT/sliding_window_kernel_stats.txt 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/slide2 -o b -- python $REPO/tools/slidetime.py --short --rig_v2 > $OUT/slide_rig_v2.log 2>&1
python $REPO/tools/prof_summary.py $OUT/slide2/b_results.db > $OUT/sliding_window_rig_v2_kernel_stats.txt 2>&1
cd $REPO
rm -rf $OUT/c4t $OUT/slide $OUT/slide2
ls -la $OUT
