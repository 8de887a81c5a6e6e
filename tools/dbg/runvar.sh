cd /tmp && export TMPDIR=/tmp
for v in bskipupd bskiptail; do
  SVIN_BA_LIB=$GRAFT_REPO_ROOT/build/variants/$v.so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/sl_$v -o b -- python $GRAFT_REPO_ROOT/tools/slidetime.py --short --rig_v2 > /tmp/sl_$v.log 2>&1
  echo "== $v"; python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/sl_$v/b_results.db 2>&1 | grep "chol_solve_lds<2>\|border" | cut -c1-140
done
