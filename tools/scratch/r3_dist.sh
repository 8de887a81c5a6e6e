# sharded path (1-rank RCCL) tests + bench with the sharded sub-record + kernel table of config #4 (one GPU)
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
REPO=$PWD
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
timeout 900 python bench.py --force-sharded > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python - $OUT/bench.json <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "config3", d.get("config3",{}).get("value"), "config4", d.get("config4_single_gpu",{}).get("value"))
print(json.dumps(d.get("sharded_config4"), indent=1)[:2500])
P
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace4 -o b -- python $REPO/tools/widetime.py > $OUT/widetime.log 2> $OUT/trace4.log
python $REPO/tools/prof_summary.py $OUT/trace4/b_results.db > $OUT/config4_kernel_stats.txt 2>&1
cd $REPO; rm -rf $OUT/trace4
tail -5 $OUT/widetime.log; head -16 $OUT/config4_kernel_stats.txt
