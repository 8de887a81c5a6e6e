"""N > 1 path on CPU: world_size-2 gloo processes check the landmark-sharded exchange.

Each rank builds the oracle's problem for ITS landmark range (rank 0 additionally owns the non-landmark
factors), linearises it, embeds the partial reduced camera system into the global ordering and sums it over
the ranks with the same all-reduce callback the GPU solver uses (svin_amd.distributed.make_torch_allreduce,
here over gloo on host memory).  The sum must equal the unsharded reduced system.
"""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import orc
    from svin_amd import synthetic as syn
    from svin_amd import distributed as sd
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        spec = syn.make_window(P=4, L=90, n_obs=900, seed=17)
        def describe(e, frames, bid):
            f, k, ix = C.c_uint64(), C.c_int(), C.c_int()
            assert e.L.orc_describe_block(e.h, int(bid), C.byref(f), C.byref(k), C.byref(ix))
            return (frames.index(int(f.value)), int(k.value), int(ix.value))

        # reference: the full problem
        full = orc.OracleEstimator()
        f_full, _ = syn.feed(full, spec)
        lin_full = full.map().linearize(0.0)
        # this rank's shard
        est = orc.OracleEstimator()
        f_shard, _ = syn.feed(est, sd.shard_spec(spec, rank, world))
        m = est.map()
        if rank != 0:  # only rank 0 owns the factors between states
            for rid in m.residual_ids():
                if m.residual_kind(rid) != 0:
                    m.remove_residual(rid)
        lin = m.linearize(0.0)
        # embed into the global ordering; blocks are matched by (frame index, kind, sensor index)
        d = lin_full["d"]
        off_full = {describe(full, f_full, b): int(o) for b, o in zip(lin_full["cam_ids"], lin_full["cam_off"])}
        dims = np.diff(np.r_[lin["cam_off"], lin["d"]])
        idx = np.concatenate([off_full[describe(est, f_shard, b)] + np.arange(k)
                              for b, k in zip(lin["cam_ids"], dims)]).astype(int) if lin["d"] else np.zeros(0, int)
        buf = np.zeros(d * d + d + 1)
        S = buf[:d * d].reshape(d, d)
        S[np.ix_(idx, idx)] = lin["S"]
        buf[d * d:d * d + d][idx] = lin["g"]
        buf[-1] = lin["cost"]
        cb = sd.make_torch_allreduce(device="cpu")
        rc = cb(buf.ctypes.data, buf.size, 0, None)
        assert rc == 0
        sd_ = np.sqrt(np.abs(np.diag(lin_full["S"])))
        dS = np.max(np.abs(S - lin_full["S"]) / np.outer(sd_, sd_))
        dg = np.max(np.abs(buf[d * d:d * d + d] - lin_full["g"]) / sd_)
        dc = abs(buf[-1] - lin_full["cost"]) / lin_full["cost"]
        # max-reduce path
        mx = np.array([float(rank)])
        cb(mx.ctypes.data, 1, 1, None)
        q.put((rank, float(dS), float(dg), float(dc), float(mx[0])))
    finally:
        dist.destroy_process_group()


def test_landmark_sharded_exchange_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, dS, dg, dc, mx in res:
        assert dS < 1e-9 and dg < 1e-9 and dc < 1e-12, (rank, dS, dg, dc)
        assert mx == 1.0


def test_shard_bounds_cover_everything():
    sys.path.insert(0, ROOT)
    from svin_amd import distributed as sd
    for n in (0, 1, 7, 50000):
        for world in (1, 2, 3, 8):
            spans = [sd.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
