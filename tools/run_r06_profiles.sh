# round-6 profile set (run on the GPU box through gpurun from the repo root); outputs under gpurun_out/r06prof.
# PMC passes use --kernel-trace only (never a sys / hip / memory-copy trace next to --pmc).
set -x
OUT=$PWD/gpurun_out/r06prof; mkdir -p $OUT
REPO=$PWD
# 0. the bench line itself (all sub-records)
python bench.py > $OUT/bench_v1.json 2> $OUT/bench_v1.err
cd /tmp && export TMPDIR=/tmp
# 1. kernel trace of the headline command (the bench line that goes with it is written next to it) + the launch timeline with gaps
rocprofv3 --kernel-trace --stats -d $OUT/trace -o b -- python $REPO/bench.py --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
python $REPO/tools/prof_summary.py $OUT/trace/b_results.db > $OUT/bench_kernel_stats.txt 2>&1
python $REPO/tools/iter_gaps.py $OUT/trace/b_results.db > $OUT/bench_iteration_gaps.txt 2>&1
# 2. K1 (Jacobian evaluation) on the HBM-resident batch: kernel-trace average, then FETCH_SIZE / WRITE_SIZE in separate passes
rocprofv3 --kernel-trace --stats -d $OUT/k1t -o b -- python $REPO/tools/k1_bench.py > $OUT/k1_trace.log 2>&1
python $REPO/tools/prof_summary.py $OUT/k1t/b_results.db > $OUT/k1_kernel_stats.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/k1f -o b -- python $REPO/tools/k1_bench.py > $OUT/k1f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/k1w -o b -- python $REPO/tools/k1_bench.py > $OUT/k1w.log 2>&1
python $REPO/tools/pmc_summary.py $OUT/k1f/b_results.db k_eval_reproj > $OUT/k1_fetch.txt 2>&1
python $REPO/tools/pmc_summary.py $OUT/k1w/b_results.db k_eval_reproj > $OUT/k1_write.txt 2>&1
# 3. config #4 (bench window): per-kernel times, then the MFMA flops k_schur_rows executes (and the round-5 tile form beside it)
SVIN_WIDE_BENCH=1 rocprofv3 --kernel-trace --stats -d $OUT/c4t -o b -- python $REPO/tools/widetime.py > $OUT/c4_trace.log 2>&1
python $REPO/tools/prof_summary.py $OUT/c4t/b_results.db > $OUT/config4_bench_kernel_stats.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace -d $OUT/c4m -o b -- python $REPO/tools/config4_mfma.py > $OUT/c4m.log 2>&1
python $REPO/tools/config4_mfma.py --summarise $OUT/c4m/b_results.db $OUT/config4_mfma.json k_schur_rows > $OUT/c4m_summary.log 2>&1
SVIN_PANELS_OLD=1 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace -d $OUT/c4mo -o b -- python $REPO/tools/config4_mfma.py > $OUT/c4mo.log 2>&1
python $REPO/tools/config4_mfma.py --summarise $OUT/c4mo/b_results.db $OUT/config4_mfma_panels_old.json k_schur_panels > $OUT/c4mo_summary.log 2>&1
# 4. solver counters (MFMA / VALU / LDS busy, waits) on the current kernels
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace -d $OUT/pmc1 -o b -- python $REPO/tools/solver_kernels.py > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS --kernel-trace -d $OUT/pmc2 -o b -- python $REPO/tools/solver_kernels.py > $OUT/pmc2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_MFMA --kernel-trace -d $OUT/pmc3 -o b -- python $REPO/tools/solver_kernels.py > $OUT/pmc3.log 2>&1
for p in pmc1 pmc2 pmc3; do python $REPO/tools/pmc_summary.py $OUT/$p/b_results.db > $OUT/$p.txt 2>&1; done
python $REPO/tools/mfma_report.py $OUT/pmc1/b_results.db $OUT/pmc3/b_results.db > $OUT/mfma_report.txt 2>&1
# 5. the sliding windows (SVIn's operating mode): per-kernel times
rocprofv3 --kernel-trace --stats -d $OUT/slide -o b -- python $REPO/tools/slidetime.py --short > $OUT/slide.log 2>&1
python $REPO/tools/prof_summary.py $OUT/slide/b_results.db > $OUT/sliding_window_kernel_stats.txt 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/slide2 -o b -- python $REPO/tools/slidetime.py --short --rig_v2 > $OUT/slide_rig_v2.log 2>&1
python $REPO/tools/prof_summary.py $OUT/slide2/b_results.db > $OUT/sliding_window_rig_v2_kernel_stats.txt 2>&1
# 6. the batched solve: one launch per stage and lane whatever B is (kernel table of B = 16, four lanes)
rocprofv3 --kernel-trace --stats -d $OUT/bat -o b -- python $REPO/tools/batchtime.py 16 > $OUT/batch.log 2>&1
python $REPO/tools/prof_summary.py $OUT/bat/b_results.db > $OUT/batch16_kernel_stats.txt 2>&1
# 7. config #3: per-kernel times and the HBM traffic of its Schur kernel + slab sum ("take a FETCH_SIZE / WRITE_SIZE pass first")
rocprofv3 --kernel-trace --stats -d $OUT/c3t -o b -- python $REPO/tools/cfg3time.py > $OUT/c3_trace.log 2>&1
python $REPO/tools/prof_summary.py $OUT/c3t/b_results.db > $OUT/config3_kernel_stats.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/c3f -o b -- python $REPO/tools/cfg3time.py > $OUT/c3f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/c3w -o b -- python $REPO/tools/cfg3time.py > $OUT/c3w.log 2>&1
(python $REPO/tools/pmc_summary.py $OUT/c3f/b_results.db k_schur_dense; python $REPO/tools/pmc_summary.py $OUT/c3f/b_results.db k_reduce_slabs; python $REPO/tools/pmc_summary.py $OUT/c3w/b_results.db k_schur_dense; python $REPO/tools/pmc_summary.py $OUT/c3w/b_results.db k_reduce_slabs) > $OUT/config3_schur_traffic.txt 2>&1
cd $REPO
rm -rf $OUT/trace $OUT/k1t $OUT/k1f $OUT/k1w $OUT/c4t $OUT/c4m $OUT/c4mo $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/slide $OUT/slide2 $OUT/bat $OUT/c3t $OUT/c3f $OUT/c3w
ls -la $OUT
