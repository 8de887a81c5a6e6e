"""The landmark-sharded solve on MORE THAN ONE GPU (the reference has no counterpart, Estimator.cpp:889 num_threads):
native RCCL communicators with 2 and -- where the node has them -- 4 and 8 ranks, one process per GPU, started exactly as the
driver starts `bench.py --gpus N` (torch.distributed.run, 127.0.0.1).  These tests SKIP on a one-GPU box (the builder's lease
and, so far, the driver's test box): they exist so that the first multi-rank RCCL run of this code happens in a test and not
inside the driver's timed scaling command ("what's missing" 1).

 * BASELINE configs[3] at full size (64 KF / 50 000 landmarks / 500 000 residuals, d = 960), three iterations: every rank
   reproduces the one-GPU solve of the whole window (run on its own GPU) to 1e-8 in the poses and speed / bias states and in
   the final cost, with the same iteration and step counts;
 * the time-limit stop vote: the ranks stop together after the minimum number of iterations;
 * a rank that dies before the first collective: the launcher ends with an error within the bound instead of hanging (the
   remaining ranks sit in ncclAllReduce; torch.distributed.run takes the group down), and bench.py's own launcher of the
   sharded sub-record -- process group of its own, killed as a whole on a time-out -- returns an error record, not a number."""
import glob
import json
import os
import signal
import socket
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_count():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:   # noqa: BLE001
        return 0


needs_two = pytest.mark.skipif(device_count() < 2, reason="needs at least two GPUs (RCCL communicator with more than one rank)")


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(world, out_prefix, size=(), env_extra=None, timeout=900):
    """torch.distributed.run with `world` ranks in a process group of its own; returns (return code or None on a time-out, log, seconds)"""
    env = {k: v for k, v in os.environ.items()
           if not k.startswith("TORCHELASTIC") and k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                             "ROLE_RANK", "ROLE_WORLD_SIZE", "MASTER_PORT", "GROUP_WORLD_SIZE")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env.update(env_extra or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "helpers", "sharded_multi_gpu.py"), out_prefix] + [str(x) for x in size]
    t0 = time.time()
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, start_new_session=True)
    try:
        log = proc.communicate(timeout=timeout)[0].decode(errors="replace")
        return proc.returncode, log, time.time() - t0
    except subprocess.TimeoutExpired:
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(proc.pid, sig)
            except OSError:
                break
            try:
                proc.communicate(timeout=10)
                break
            except subprocess.TimeoutExpired:
                continue
        return None, "timed out", time.time() - t0


def worlds():
    n = device_count()
    return [w for w in (2, 4, 8) if w <= n] or [2]


@needs_two
@pytest.mark.parametrize("world", worlds())
def test_sharded_config4_on_several_gpus(gpu_lib, tmp_path, world):
    prefix = str(tmp_path / "res")
    rc, log, secs = launch(world, prefix)
    assert rc == 0, "the ranks failed (rc %s after %.0f s):\n%s" % (rc, secs, log[-4000:])
    res = [json.load(open(f)) for f in sorted(glob.glob(prefix + ".rank*.json"))]
    assert len(res) == world and sorted(r["device"] for r in res) == list(range(world))   # one GPU per rank
    for r in res:
        print(r)
        assert r["iterations"] == r["ref_iterations"] == 3 and r["successful"] == r["ref_successful"]
        assert abs(r["final_cost"] - r["ref_final_cost"]) <= 1e-8 * r["ref_final_cost"]
        assert r["pose_diff"] < 1e-8 and r["speed_bias_diff"] < 1e-8
        assert r["limit_termination"] == 2 and 2 <= r["limit_iterations"] <= 3
        assert r["allreduce_us"] > 0
    assert len({r["final_cost"] for r in res}) == 1 and len({r["limit_iterations"] for r in res}) == 1   # bit-identical on every rank


@needs_two
def test_a_dead_rank_is_an_error_not_a_hang(gpu_lib, tmp_path):
    """rank 1 exits before the first collective of the solve (a small window: what matters is the launcher's behaviour)"""
    rc, log, secs = launch(2, str(tmp_path / "dead"), size=(12, 2000, 20000), env_extra={"SVIN_TEST_KILL_RANK": "1"}, timeout=300)
    print("launcher rc %s after %.0f s" % (rc, secs))
    assert rc is not None, "the launcher hung for 300 s with a dead rank"
    assert rc != 0
    assert not glob.glob(str(tmp_path / "dead") + ".rank*.json")   # nobody reported a result


@needs_two
def test_bench_launcher_reports_a_dead_rank(gpu_lib, monkeypatch):
    """bench.py's own launcher of the sharded sub-record (run_sharded_children: process group of its own, killed as a whole when
    the bound passes): a rank that dies gives an error record"""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setenv("SVIN_BENCH_KILL_RANK", "1")
    monkeypatch.setenv("SVIN_BENCH_SHARDED_TIMEOUT", "240")
    t0 = time.time()
    rec = bench.run_sharded_children(2, False, transport="rccl", steps=1)
    print(rec, "after %.0f s" % (time.time() - t0))
    assert "error" in rec and "value" not in rec
    assert time.time() - t0 < 300


@needs_two
def test_bench_sharded_record_native_rccl(gpu_lib):
    """the sub-record the driver's SCALE run carries, produced by two ranks on two GPUs over RCCL; its numbers reach `summary`"""
    sys.path.insert(0, ROOT)
    import bench
    rec = bench.run_sharded_children(2, False, transport="rccl", steps=2)
    print({k: rec.get(k) for k in ("value", "ms_per_iteration", "iterations_per_step", "final_cost", "transport", "allreduce_GBps", "error")})
    assert "error" not in rec, rec
    assert rec["n_gpus"] == 2 and rec["landmarks_per_rank"] == 25000 and rec["transport"] == "RCCL"
    assert rec["allreduce_GBps"]["frac_of_link"] is not None and rec["k1_roofline_per_gpu"]["frac"] > 0
    s = bench.summary_of({"sharded_config4": rec})
    assert s["sharded_config4_its"] == round(rec["value"], 4)
    assert s["sharded_allreduce_frac_of_link"] is not None and s["sharded_k1_frac_per_gpu"] is not None
