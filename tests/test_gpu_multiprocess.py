"""Two PROCESSES, one landmark-sharded window (the reference has no counterpart, Estimator.cpp:889 num_threads).

Until an 8-GPU node runs it, no communicator with more than one rank has executed the sharded solve; the emulation in
test_gpu_parity.py uses two threads of ONE process.  Here two processes share GPU 0, each with its own handle, its own
torch.distributed rank (gloo, 127.0.0.1) and an all-reduce callback that stages the device buffer through host memory.
BASELINE configs[3] at full size (64 KF / 50 000 landmarks / 500 000 residuals, d = 960): both ranks must reproduce the
plain single-GPU solve (computed in the same process) to 1e-9, take the same number of iterations, and stop together on
the time-limit vote."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(tmp_path, size, timeout):
    port, world = free_port(), 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    outs = [str(tmp_path / ("rank%d.json" % r)) for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "helpers", "sharded_two_proc.py"), str(r), str(world),
                               str(port), outs[r]] + [str(x) for x in size], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    logs = []
    try:
        for p in procs:
            logs.append(p.communicate(timeout=timeout)[0].decode(errors="replace"))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, logs[r][-3000:])
    return [json.load(open(o)) for o in outs]


@pytest.mark.parametrize("size", [(64, 50000, 500000)])
def test_two_process_sharded_config4(gpu_lib, tmp_path, size):
    res = run_ranks(tmp_path, size, timeout=600)
    for r in res:
        print(r)
        assert r["iterations"] == r["ref_iterations"] == 3 and r["successful"] == r["ref_successful"]
        assert abs(r["final_cost"] - r["ref_final_cost"]) <= 1e-9 * r["ref_final_cost"]
        assert r["pose_diff"] < 1e-9 and r["speed_bias_diff"] < 1e-9
        assert r["limit_termination"] == 2 and 2 <= r["limit_iterations"] <= 3
    assert res[0]["final_cost"] == res[1]["final_cost"] and res[0]["limit_iterations"] == res[1]["limit_iterations"]


def test_bench_sharded_record_two_ranks_on_one_gpu(gpu_lib):
    """bench.py's N > 1 sub-record (`sharded_config4`: rank launch through torch.distributed.run, shard, three collectives per
    iteration, slowest-rank timing, the JSON hand-over) produced by two ranks that share GPU 0 -- --sharded-transport=stage, gloo
    over host-staged buffers -- so that the first multi-GPU run of the driver is not also the first run of this code.  The
    record must describe the same solve as the plain one-GPU optimisation of that window."""
    sys.path.insert(0, ROOT)
    import bench
    from svin_amd import synthetic as syn
    from svin_amd.estimator import Estimator
    rec = bench.run_sharded_children(2, False, transport="stage", steps=2)
    print({k: rec.get(k) for k in ("value", "ms_per_iteration", "iterations_per_step", "final_cost", "transport", "error", "stderr_tail")})
    assert "error" not in rec, rec
    assert rec["n_gpus"] == 2 and rec["landmarks_per_rank"] == 25000 and rec["transport"].startswith("gloo")
    est = Estimator(0)
    syn.feed(est, syn.make_window(P=64, L=50000, n_obs=500000, seed=20250629, frame_dt=0.25))
    est.optimize(5)
    s = est.summary()
    assert rec["iterations_per_step"] == s["iterations"]
    assert abs(rec["final_cost"] - s["final_cost"]) <= 1e-9 * s["final_cost"]
