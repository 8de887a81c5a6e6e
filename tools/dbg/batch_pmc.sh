# (gpurun helper) hardware counters of the batched kernels, B = 16 in ONE lane (no other lane's kernels beside them)
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp
export SVIN_BATCH_LANES=1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace -d /tmp/bp1 -o b -- python $GRAFT_REPO_ROOT/tools/batchtime.py 16 > /tmp/bp1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAVES --kernel-trace -d /tmp/bp2 -o b -- python $GRAFT_REPO_ROOT/tools/batchtime.py 16 > /tmp/bp2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/bp3 -o b -- python $GRAFT_REPO_ROOT/tools/batchtime.py 16 > /tmp/bp3.log 2>&1
for k in k_schur_dense_batch k_post_solve_batch "k_schur_dense<" "k_post_solve<"; do
  for d in bp1 bp2 bp3; do python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/$d/b_results.db "$k" | tail -n +2 | cut -c1-40,88-150; done
done > $GRAFT_REPO_ROOT/gpurun_out/r06/batch_pmc.txt 2>&1
