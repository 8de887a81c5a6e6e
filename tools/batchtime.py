"""The `batched` sub-record of bench.py on its own (svin_ba_solve_prepared_batch: B configs[1] windows per launch sequence)."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
sizes = tuple(int(a) for a in sys.argv[1:]) or (16, 64)
print(json.dumps(bench.batched_record(0, sizes=sizes), indent=1))
