"""One rank of the landmark-sharded solve on SEVERAL GPUs (tests/test_gpu_multigpu.py): one process per GPU, started by
torch.distributed.run, native RCCL on the solver's stream (svin_ba_set_distributed_rccl) -- the transport the driver's
`bench.py --gpus N` uses.  BASELINE configs[3] at full size unless a size is given.
usage (under torch.distributed.run): sharded_multi_gpu.py <out-prefix> [P L n_obs]
environment: SVIN_TEST_KILL_RANK=r -- rank r leaves (exit code 17) after the communicator is up and before the first collective
of the solve: the other ranks block in ncclAllReduce until the launcher takes the group down."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def pose_diff(Ta, Tb):
    dq = Ta[3:] - Tb[3:] * np.sign(Ta[3:] @ Tb[3:])
    return max(np.linalg.norm(Ta[:3] - Tb[:3]) / max(1.0, np.linalg.norm(Tb[:3])), np.linalg.norm(dq))


def main():
    out = sys.argv[1]
    P, L, n_obs = (int(x) for x in sys.argv[2:5]) if len(sys.argv) >= 5 else (64, 50000, 500000)
    rank, world, local_rank = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    import datetime
    import torch
    import torch.distributed as dist
    from svin_amd import distributed as sd
    from svin_amd import synthetic as syn
    from svin_amd.estimator import Estimator
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", timeout=datetime.timedelta(minutes=10))
    spec = syn.make_window(P=P, L=L, n_obs=n_obs, seed=20250629, frame_dt=0.25)
    iters = 3
    ref = Estimator(local_rank)              # the plain one-GPU solve of the whole window, on this rank's GPU
    f_ref, _ = syn.feed(ref, spec)
    ref.optimize(iters)
    s_ref = ref.summary()
    est = Estimator(local_rank)              # this rank's share: all states and factors, its range of landmarks
    f, _ = syn.feed(est, sd.shard_spec(spec, rank, world))
    sd.init_rccl(est, rank, world)
    if os.environ.get("SVIN_TEST_KILL_RANK") == str(rank):
        os._exit(17)
    est.optimize(iters)
    s = est.summary()
    worst = max(pose_diff(est.get_T_WS(a), ref.get_T_WS(b)) for a, b in zip(f, f_ref))
    sb = max(float(np.max(np.abs(est.get_speed_and_bias(a) - ref.get_speed_and_bias(b)))) for a, b in zip(f, f_ref))
    # the time limit is over when the first iteration ends: the ranks vote with the evaluation's all-reduce and stop together
    est.set_time_limit(1e-6, 2)
    est.optimize(10)
    s_lim = est.summary()
    # the collective on its own: the [lower(S) | g | h] message of this window
    d = 15 * P
    ar_us = est.bench_allreduce(d * (d + 1) // 2 + 3 * d, 10)
    res = dict(rank=rank, world=world, device=torch.cuda.current_device(), iterations=s["iterations"], ref_iterations=s_ref["iterations"],
               successful=s["successful"], ref_successful=s_ref["successful"], final_cost=s["final_cost"], ref_final_cost=s_ref["final_cost"],
               pose_diff=float(worst), speed_bias_diff=sb, limit_termination=s_lim["termination"], limit_iterations=s_lim["iterations"],
               allreduce_us=ar_us)
    with open("%s.rank%d.json" % (out, rank), "w") as fh:
        json.dump(res, fh)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
