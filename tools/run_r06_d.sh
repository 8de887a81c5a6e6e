# round-6 scratch run: the batched solve under rocprof (kernel table of B = 16)
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_batch.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | grep "^E  \|passed\|failed" | cut -c1-300 | head -20
for l in 1 2 4 8; do echo "lanes $l"; SVIN_BATCH_LANES=$l python tools/batchtime.py 16 64 2>&1 | grep "aggregate\|x_one" ; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/bt -o b -- python $GRAFT_REPO_ROOT/tools/batchtime.py 16 > $GRAFT_REPO_ROOT/gpurun_out/r06/d_batch.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/bt/b_results.db > $GRAFT_REPO_ROOT/gpurun_out/r06/d_batch_kernel_stats.txt 2>&1
head -24 $GRAFT_REPO_ROOT/gpurun_out/r06/d_batch_kernel_stats.txt | cut -c1-150
