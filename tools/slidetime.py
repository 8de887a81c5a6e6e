"""The `sliding_window` record of bench.py on its own (SVIn's operating mode: addStates, ~1 000 addObservation, optimize(10),
applyMarginalizationStrategy(5, 3) per frame); --cpu adds the oracle beside it."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
rec = bench.sliding_window_record(0, with_oracle="--cpu" in sys.argv)
print(json.dumps(rec, indent=1))
