# round-6 scratch run: batch kernel durations against the number of windows in ONE lane
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp
for b in 2 4 8; do
SVIN_BATCH_LANES=1 rocprofv3 --kernel-trace --stats -d /tmp/bf$b -o b -- python $GRAFT_REPO_ROOT/tools/batchtime.py $b > /tmp/logf$b 2>&1
echo "B = $b, one lane"; grep "x_one" /tmp/logf$b
python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/bf$b/b_results.db | grep "_batch" | cut -c1-60,78-140
done
