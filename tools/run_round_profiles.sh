set -x
cd /root/repo
python bench.py > gpurun_out/bench_v9.json 2> gpurun_out/bench_v9.err
python bench.py --workload posegraph --steps 10 --warmup 2 > gpurun_out/bench_pg4_v5.json 2> gpurun_out/bench_pg4_v5.err
python bench.py --workload posegraph --six-dof --steps 10 --warmup 2 > gpurun_out/bench_pg6_v5.json 2> gpurun_out/bench_pg6_v5.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_v9 -o b -- python /root/repo/bench.py --no-cpu-baseline > /root/repo/gpurun_out/prof_v9.log 2>&1
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_pg4_v5 -o b -- python /root/repo/bench.py --workload posegraph --steps 10 --warmup 2 --no-cpu-baseline > /root/repo/gpurun_out/prof_pg4_v5.log 2>&1
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_pg6_v5 -o b -- python /root/repo/bench.py --workload posegraph --six-dof --steps 10 --warmup 2 --no-cpu-baseline > /root/repo/gpurun_out/prof_pg6_v5.log 2>&1
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_slide -o b -- python /root/repo/tools/margtime.py --rig-v2 > /root/repo/gpurun_out/prof_slide.log 2>&1
cd /root/repo
for d in prof_slide prof_v9 prof_pg4_v5 prof_pg6_v5; do python tools/prof_summary.py gpurun_out/$d/b_results.db > gpurun_out/$d.txt 2>&1; done
tail -c 600 gpurun_out/bench_v9.json; echo; tail -c 1500 gpurun_out/bench_pg4_v5.json; echo; tail -c 800 gpurun_out/bench_pg6_v5.json
