# round-3 counter passes of the solver kernels (run on the GPU box through gpurun from the repo root); outputs under gpurun_out/r03pmc
OUT=$PWD/gpurun_out/r03pmc; mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace -d $OUT/pmc1 -o b -- python $REPO/tools/solver_kernels.py > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS --kernel-trace -d $OUT/pmc2 -o b -- python $REPO/tools/solver_kernels.py > $OUT/pmc2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_MFMA --kernel-trace -d $OUT/pmc3 -o b -- python $REPO/tools/solver_kernels.py > $OUT/pmc3.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -d $OUT/pmc4 -o b -- python $REPO/tools/solver_kernels.py > $OUT/pmc4.log 2>&1
for p in pmc1 pmc2 pmc3 pmc4; do python $REPO/tools/pmc_summary.py $OUT/$p/b_results.db > $OUT/$p.txt 2>&1; done
python $REPO/tools/mfma_report.py $OUT/pmc1/b_results.db $OUT/pmc3/b_results.db > $OUT/mfma_report.txt 2>&1
cd $REPO
rm -rf $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4
grep "chol_solve_lds" $OUT/pmc*.txt
