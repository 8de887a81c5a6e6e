"""profiles/rNN_k1_pmc.json from the two counter passes of tools/run_r06_profiles.sh (pmc_summary.py outputs of FETCH_SIZE and
WRITE_SIZE for the batched k_eval_reproj launch).  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-counts by a
factor of two (MI355X_MICROARCH.md, HBM / rocprofv3 section): traffic = 2 x fetch + write.
usage: python tools/k1_pmc_json.py <k1_fetch.txt> <k1_write.txt> <replicas> <algorithmic bytes per launch> <out.json>"""
import json, sys


def mean_of(path, counter):
    for line in open(path):
        t = line.split()
        if counter in t and "k_eval_reproj" in line:
            return float(t[t.index(counter) + 2])
    raise SystemExit("no %s line in %s" % (counter, path))


fetch, write = mean_of(sys.argv[1], "FETCH_SIZE"), mean_of(sys.argv[2], "WRITE_SIZE")
replicas, alg = int(sys.argv[3]), float(sys.argv[4])
out = {"kernel": "k_eval_reproj<true,false>", "workload": "%d replicas of config #2 (%.2f M observations)" % (replicas, 20000 * replicas / 1e6),
       "fetch_size_kib": fetch, "fetch_correction": 2.0, "write_size_kib": write,
       "traffic_bytes_per_launch": 1024.0 * (2.0 * fetch + write), "algorithmic_bytes_per_launch": alg,
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python tools/k1_bench.py "
                 "(tools/run_r06_profiles.sh)"}
json.dump(out, open(sys.argv[5], "w"), indent=1)
print(json.dumps(out, indent=1))
