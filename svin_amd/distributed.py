"""Landmark-sharded multi-GPU solve: host-side plumbing.

One process per GPU.  Every rank holds all states (poses, speed/bias, extrinsics) and the factors
between them; landmarks -- with *all* their observations -- are partitioned into contiguous ranges, one
per rank.  Per Gauss-Newton iteration each rank eliminates its own landmarks and accumulates a partial
reduced camera system; ONE all-reduce (sum, FP64, the lower triangle + three vectors: d (d + 1) / 2 + 3 d doubles) makes the system identical on every
rank, which then solves it redundantly and back-substitutes only its own landmarks.  Two more tiny
all-reduces carry the trust-region scalars.  The collective is RCCL over xGMI (`torch.distributed`
backend "nccl"); on CPU the same code path runs over gloo for the tests.
"""
import ctypes as C
import threading

import numpy as np

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p)


def shard_bounds(n_items, rank, world):
    """contiguous, balanced [lo, hi) range of `n_items` for `rank`"""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_spec(spec, rank, world):
    """Returns a copy of a synthetic WindowSpec that keeps only this rank's landmark range."""
    import copy
    lo, hi = shard_bounds(spec.L, rank, world)
    keep = (spec.obs_lm >= lo) & (spec.obs_lm < hi)
    out = copy.copy(spec)
    out.lm_true = spec.lm_true[lo:hi].copy()
    out.lm_init = spec.lm_init[lo:hi].copy()
    out.obs_lm = spec.obs_lm[keep] - lo
    out.obs_frame = spec.obs_frame[keep].copy()
    out.obs_cam = spec.obs_cam[keep].copy()
    out.obs_uv = spec.obs_uv[keep].copy()
    out.obs_size = spec.obs_size[keep].copy()
    return out


def init_rccl(est, rank, world, group=None, device=None):
    """Switches `est` to the landmark-sharded mode with native RCCL: rank 0 creates the ncclUniqueId, torch.distributed
    (any backend; only this 128-byte hand-shake goes through it) broadcasts it, every rank joins the communicator."""
    import torch
    import torch.distributed as dist
    from . import estimator
    uid = estimator.rccl_unique_id() if rank == 0 else bytes(128)
    backend = dist.get_backend(group)
    dev = device if device is not None else ("cuda" if backend == "nccl" else "cpu")
    t = torch.tensor(list(uid), dtype=torch.uint8, device=dev)
    dist.broadcast(t, src=0, group=group)
    est.set_distributed_rccl(rank, world, bytes(t.cpu().tolist()))


class _DevArray:
    """exposes a raw device pointer through __cuda_array_interface__ so that torch can alias it"""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def make_torch_allreduce(group=None, device="cuda"):
    """all-reduce callback backed by torch.distributed: device "cuda" (nccl == RCCL on ROCm, in place on the GPU buffer),
    "stage" (GPU buffer staged through host memory for a CPU backend such as gloo) or "cpu" (host buffer, gloo)."""
    import torch
    import torch.distributed as dist

    def fn(ptr, count, op, _user):
        try:
            if device == "cuda":
                t = torch.as_tensor(_DevArray(ptr, count), device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX, group=group)
                torch.cuda.synchronize()
            elif device == "stage":
                # the solver's buffer lives on the GPU, the collective runs on a CPU backend (gloo): device -> host, reduce,
                # host -> device.  What a run without RCCL (two processes sharing ONE GPU in the tests) has to do.
                t = torch.as_tensor(_DevArray(ptr, count), device="cuda")
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX, group=group)
                t.copy_(h)
                torch.cuda.synchronize()
            else:
                buf = (C.c_double * count).from_address(ptr)
                a = np.frombuffer(buf, dtype=np.float64)
                t = torch.from_numpy(a)
                dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX, group=group)
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            print("all-reduce callback failed:", e)
            return 1
    return ALLREDUCE_FN(fn)


class ThreadAllReduce:
    """In-process stand-in for RCCL used by the single-GPU emulation test: `world` solver threads (one
    estimator handle each, same GPU) meet at a barrier; rank 0 combines the device buffers with torch."""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world

    def callback(self, rank):
        import torch

        def fn(ptr, count, op, _user):
            try:
                self.slots[rank] = torch.as_tensor(_DevArray(ptr, count), device="cuda")
                self.barrier.wait()
                if rank == 0:
                    stack = torch.stack([t.clone() for t in self.slots])
                    red = stack.sum(0) if op == 0 else stack.max(0).values
                    for t in self.slots:
                        t.copy_(red)
                    torch.cuda.synchronize()
                self.barrier.wait()
                return 0
            except Exception as e:
                print("thread all-reduce failed:", e)
                return 1
        return ALLREDUCE_FN(fn)
