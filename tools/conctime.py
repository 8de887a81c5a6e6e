"""The `concurrent` record of bench.py on its own."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
print(json.dumps(bench.concurrent_record(0), indent=1))
