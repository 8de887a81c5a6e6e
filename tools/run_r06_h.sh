# round-6 scratch run: config #3 against the number of chunks per slab workgroup
for c in 1 2 3 4; do echo "chunks $c"; SVIN_SLAB_CHUNKS=$c python tools/cfg3time.py 2>&1 | grep "config3\|kernel ms" | tail -2; done
cd /tmp && export TMPDIR=/tmp
for c in 1 2; do
SVIN_SLAB_CHUNKS=$c rocprofv3 --kernel-trace --stats -d /tmp/c3$c -o c -- python $GRAFT_REPO_ROOT/tools/cfg3time.py > /tmp/c3$c.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/c3$c/c_results.db | head -9 | cut -c1-150
done
