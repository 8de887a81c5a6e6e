# round-6 scratch run: the block-pair Schur form on the wide window (tests, timing against the tile form, kernel table, in-kernel stamps)
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wide_window_panels or config4_full_size or native_rccl" > gpurun_out/r06/a_tests.txt 2>&1
SVIN_WIDE_BENCH=1 python tools/widetime.py > gpurun_out/r06/a_wide_new.txt 2>&1
SVIN_WIDE_BENCH=1 python tools/widetime.py --old > gpurun_out/r06/a_wide_old.txt 2>&1
SVIN_BA_LIB=$PWD/build/variants/blkt.so SVIN_WIDE_BENCH=1 python tools/widetime.py 2>&1 | grep "blocks block\|rows block" | head -12 > gpurun_out/r06/a_stamps.txt
cd /tmp && export TMPDIR=/tmp
SVIN_WIDE_BENCH=1 rocprofv3 --kernel-trace --stats -d /tmp/c4t -o b -- python $GRAFT_REPO_ROOT/tools/widetime.py > $GRAFT_REPO_ROOT/gpurun_out/r06/a_trace.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/c4t/b_results.db > $GRAFT_REPO_ROOT/gpurun_out/r06/a_config4_kernel_stats.txt 2>&1
cd $GRAFT_REPO_ROOT
grep -n "passed\|failed\|rror" gpurun_out/r06/a_tests.txt | head -5; tail -2 gpurun_out/r06/a_wide_new.txt; tail -2 gpurun_out/r06/a_wide_old.txt; cat gpurun_out/r06/a_stamps.txt; head -14 gpurun_out/r06/a_config4_kernel_stats.txt
