# (gpurun helper) the wide window (config #4 shape) with / without the early speed-bias chain, tests of the wide paths, kernel table
mkdir -p gpurun_out/r06
SVIN_WIDE_BENCH=1 python tools/widetime.py 2>&1 | grep "solve(5)"
SVIN_NO_SB_EARLY=1 SVIN_WIDE_BENCH=1 python tools/widetime.py 2>&1 | grep "solve(5)"
python -m pytest tests -x -q -m gpu -k "wide or config4 or config3 or reduced_solve" 2>&1 | grep "passed\|failed\|^E   " | head
cd /tmp && export TMPDIR=/tmp
SVIN_WIDE_BENCH=1 rocprofv3 --kernel-trace --stats -d /tmp/wd -o w -- python $GRAFT_REPO_ROOT/tools/widetime.py > /tmp/wd.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/wd/w_results.db > $GRAFT_REPO_ROOT/gpurun_out/r06/g_wide_kernel_stats.txt 2>&1
head -12 $GRAFT_REPO_ROOT/gpurun_out/r06/g_wide_kernel_stats.txt | cut -c1-150
