# round-6: the whole GPU suite, the CPU-visible smoke, and the default bench line
mkdir -p gpurun_out/r06
python -m pytest tests -x -q -m gpu > gpurun_out/r06/full_tests.txt 2>&1; tail -3 gpurun_out/r06/full_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06/smoke.txt 2>&1; tail -1 gpurun_out/r06/smoke.txt
python bench.py > gpurun_out/r06/bench_a.json 2> gpurun_out/r06/bench_a.err; tail -c 1500 gpurun_out/r06/bench_a.json
