# round-6 scratch run: k_blocks_slots against the number of accumulator copies (occupancy against LDS-atomic conflicts)
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-cop2 cop1}; do
  SVIN_BA_LIB=$GRAFT_REPO_ROOT/build/variants/$v.so SVIN_WIDE_BENCH=1 rocprofv3 --kernel-trace --stats -d /tmp/c4$v -o b -- python $GRAFT_REPO_ROOT/tools/widetime.py > /dev/null 2>&1
  echo $v; python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/c4$v/b_results.db 2>&1 | grep "k_blocks_slots\|k_blocks_pose_reduce" | cut -c1-130
done
