"""One rank of the two-PROCESS landmark-sharded solve (tests/test_gpu_multiprocess.py): everything of the multi-process path
except RCCL itself -- per-process device selection, torch.distributed bring-up, factors dealt by creation number, the three
all-reduces per iteration (here: GPU buffer staged through host memory, gloo), the stop vote.  Both ranks share GPU 0.
usage: sharded_two_proc.py <rank> <world> <port> <out.json> [P L n_obs]"""
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
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    P, L, n_obs = (int(x) for x in sys.argv[5:8]) if len(sys.argv) >= 8 else (64, 50000, 500000)
    import torch
    import torch.distributed as dist
    from svin_amd import distributed as sd
    from svin_amd import synthetic as syn
    from svin_amd.estimator import Estimator
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    torch.cuda.set_device(0)
    spec = syn.make_window(P=P, L=L, n_obs=n_obs, seed=20250629, frame_dt=0.25)
    iters = 3
    ref = Estimator(0)                       # the plain single-GPU solve of the whole window
    f_ref, _ = syn.feed(ref, spec)
    ref.optimize(iters)
    s_ref = ref.summary()
    est = Estimator(0)                       # this rank's share: all states and factors, its range of landmarks
    f, _ = syn.feed(est, sd.shard_spec(spec, rank, world))
    cb = sd.make_torch_allreduce(device="stage")
    est.set_distributed(rank, world, cb)
    est.optimize(iters)
    s = est.summary()
    worst = max(pose_diff(est.get_T_WS(a), ref.get_T_WS(b)) for a, b in zip(f, f_ref))
    sb = max(float(np.max(np.abs(est.get_speed_and_bias(a) - ref.get_speed_and_bias(b)))) for a, b in zip(f, f_ref))
    # the time limit is over when the first iteration ends: the ranks vote with the evaluation's all-reduce and stop together,
    # after the minimum number of iterations (a rank leaving on its own clock would leave the other in the next collective)
    est.set_time_limit(1e-6, 2)
    est.optimize(10)
    s_lim = est.summary()
    res = dict(rank=rank, iterations=s["iterations"], ref_iterations=s_ref["iterations"], successful=s["successful"],
               ref_successful=s_ref["successful"], final_cost=s["final_cost"], ref_final_cost=s_ref["final_cost"],
               pose_diff=float(worst), speed_bias_diff=sb, limit_termination=s_lim["termination"], limit_iterations=s_lim["iterations"])
    with open(out, "w") as fh:
        json.dump(res, fh)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
