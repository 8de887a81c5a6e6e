"""The `sliding_window` record of bench.py on its own (SVIn's operating mode: addStates, ~1 000 addObservation, optimize(10),
applyMarginalizationStrategy(5, 3) per frame); --cpu adds the oracle beside it, --rig_v2 takes the stereo_rig_v2 + sonar + depth rig."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
rec = bench.sliding_window_record(0, with_oracle="--cpu" in sys.argv, rig="rig_v2" if "--rig_v2" in sys.argv else "euroc")
if "--short" in sys.argv:
    print(rec["workload"][:40], "back to back %.3f ms / frame" % rec["ms_per_frame"], {k: round(v, 3) for k, v in rec["ms"].items()})
    fs = rec["frames_spaced"]
    print("   frames spaced: %.3f ms / frame (host %.3f), marginalisation job %.3f ms" % (fs["ms_per_frame"], fs["host_ms_per_frame"], fs["marginalisation_job_ms"]),
          {k: round(v, 3) for k, v in fs["ms"].items()})
else:
    print(json.dumps(rec, indent=1))
