# round-6 scratch run: k_schur_rows against SVIN_BLK_ROUNDS (workgroups per place), kernel table per setting
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wide_window_panels or config4_full_size" > gpurun_out/r06/b_tests.txt 2>&1
grep -n "passed\|failed\|rror" gpurun_out/r06/b_tests.txt | head -5
cd /tmp && export TMPDIR=/tmp
for r in ${ROUNDS:-1 2 3 4}; do
  SVIN_BLK_ROUNDS=$r SVIN_WIDE_BENCH=1 rocprofv3 --kernel-trace --stats -d /tmp/c4r$r -o b -- python $GRAFT_REPO_ROOT/tools/widetime.py > $GRAFT_REPO_ROOT/gpurun_out/r06/b_trace_$r.log 2>&1
  python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/c4r$r/b_results.db > $GRAFT_REPO_ROOT/gpurun_out/r06/b_stats_$r.txt 2>&1
  echo "rounds $r: $(grep 'solve(5)' $GRAFT_REPO_ROOT/gpurun_out/r06/b_trace_$r.log | tail -1 | cut -c1-70)"
  grep "k_schur_rows\|k_reduce_panel_slabs\|k_blocks_slots" $GRAFT_REPO_ROOT/gpurun_out/r06/b_stats_$r.txt | cut -c1-130
done
