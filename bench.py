#!/usr/bin/env python3
"""Headline benchmark: Gauss-Newton iterations/s of the sliding-window BA backend + Jacobian-eval roofline.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): synthetic stereo+IMU window, 10 keyframes / 2 000 landmarks /
20 000 reprojection residuals / 9 IMU factors, seed 20250629, FP64.  One *step* = one
`Estimator::optimize(10)` trust-region solve from the seeded perturbed initial state with the inputs
already resident in HBM (upload and download are excluded and reported separately); `value` =
Gauss-Newton iterations completed / wall time of the K solves (device-synchronised, max over ranks).

N > 1: `--gpus N` launches N ranks itself (torch.distributed.run, one process per GPU) unless it already runs under
such a launcher, in which case WORLD_SIZE must equal N.  The headline stays config #2 with one replica window per rank
(the window is far below the size where sharding pays, SURVEY.md 8(e): weak scaling, no data-path collective); the
line additionally carries `sharded_config4`: ONE 64-keyframe / 50 000-landmark window whose landmarks are split over
the ranks, the reduced camera system all-reduced by RCCL on the solver's stream every iteration.

Sub-records in the same JSON line (rank 0, measured after the headline region): `config3` (sonar + depth + per-frame
extrinsics), `config4_single_gpu`, `config5` (pose graph), `cpu_baseline` at 1 / 2 / all host threads.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# cpu_baseline: the oracle's OpenMP regions are short (one evaluation, one Schur elimination per iteration); with the default
# passive wait policy the workers sleep between them and TWO threads came out slower than one (42 against 25 ms per iteration
# where spinning workers give 17).  Has to be in the environment before the first OpenMP runtime of the process starts.
os.environ.setdefault("OMP_WAIT_POLICY", "active")

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_CEILING_GBS = 6290.0   # measured copy ceiling quoted by the same guide (report both fractions)
F64_MFMA_PEAK_TFLOPS = 78.6  # 256 CUs x 4 SIMDs x 2048 flop / 64 cycles (v_mfma_f64_16x16x4_f64, tools/ubench) x 2.4 GHz


def snapshot_init(est, fids, lids, spec):
    return dict(T=[spec.T_WS_init[k] for k in range(len(fids))], sb=[spec.sb_init[k] for k in range(len(fids))],
                lm=[spec.lm_init[l] for l in range(len(lids))], T0=est.get_T_WS(fids[0]).copy())


def reset_state(est, fids, lids, snap):
    est.set_T_WS(fids[0], snap["T0"])
    for k, f in enumerate(fids):
        if k > 0:
            est.set_T_WS(f, snap["T"][k])
        est.set_speed_and_bias(f, snap["sb"][k])
    for l, lid in enumerate(lids):
        est.set_landmark(lid, snap["lm"][l])
    if hasattr(est, "invalidate_preintegration"):
        est.invalidate_preintegration()


def timed_solves(est, fids, lids, snap, steps, warmup, iters, sync):
    """`steps` solves from the same start; returns (per-step seconds, per-step iterations, last summary)"""
    times, its, last = [], [], None
    for k in range(warmup + steps):
        reset_state(est, fids, lids, snap)
        est.prepare()                       # pack + upload (untimed: inputs resident in HBM)
        sync()
        t0 = time.perf_counter()
        est.solve_prepared(iters)           # returns device-synchronised
        sync()
        dt = time.perf_counter() - t0
        last = est.summary()
        est.finish()
        if k >= warmup:
            times.append(dt)
            its.append(last["iterations"])
    return times, its, last


def cpu_baseline(spec, budget_s=6.0):
    """Oracle (CPU restatement of the reference's Ceres path) on the same window: optimize(10) from the same start, at
    1 thread, 2 threads (what the pipeline uses, ThreadedKFVio.cpp:1086) and all host cores."""
    from oracle import orc
    from svin_amd import synthetic as syn
    try:
        lib = orc.lib(orc.build(native=True, out="/tmp/liborc_native.so"))
    except Exception:
        lib = orc.lib()
    nproc = os.cpu_count() or 1
    rows = {}
    for nt in sorted({1, 2, nproc}):
        iters, total, runs = 0, 0.0, 0
        while total < budget_s and runs < 200:
            est = orc.OracleEstimator(L=lib)
            syn.feed(est, spec)
            t0 = time.perf_counter()
            est.optimize(10, nt, False)
            total += time.perf_counter() - t0
            iters += est.summary()["iterations"]
            runs += 1
        rows[nt] = dict(value=iters / total, runs=runs, iterations=iters, seconds=total)
    one = rows[1]
    return dict(value=one["value"], unit="GN iterations/s", cores=1, kind="port", nproc=nproc,
                threads={str(nt): r["value"] for nt, r in rows.items()},
                sample="%d x optimize(10) on the full config-#2 window (%d iterations, %.1f s) per thread count; oracle/ C++ "
                       "restatement of the reference's Ceres path (analytic Jacobians, landmark Schur, dense Cholesky), g++ -O3 "
                       "-march=native, OMP_WAIT_POLICY=active; the 1-thread figure is the baseline -- the 2-thread one is what the pipeline's setting "
                       "(ThreadedKFVio.cpp:1086) gives the port's OpenMP loops (residual evaluation, Schur elimination; the dense solve stays "
                       "serial), not a measurement of Ceres at num_threads = 2; host has %d logical cores" % (one["runs"], one["iterations"], one["seconds"], nproc))


_REAL_STDOUT = None


def protect_stdout():
    """The contract is ONE JSON line on stdout.  RCCL prints a version banner through C stdio (flushed at exit, i.e. after
    anything Python printed), and other libraries may do the like: from here on file descriptor 1 is stderr for everybody,
    and emit() writes the line to the real stdout."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    line = (json.dumps(obj) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, line)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args):
    """`python bench.py --gpus N` outside a launcher: start N ranks (one per GPU) and pass their output through"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def init_distributed(args):
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node %d, or let bench.py spawn the "
                         "ranks itself)" % (args.gpus, world, args.gpus))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    dist = None
    if getattr(args, "sharded_child", False) and getattr(args, "sharded_transport", "rccl") == "stage":
        # the ranks of the sharded sub-record on however many GPUs there are (one will do), their collectives over gloo with the
        # device buffers staged through host memory: every line of the N > 1 record's code except RCCL itself
        import datetime
        import torch.distributed as dist
        local_rank = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=20))
        return rank, world, local_rank, dist
    if world > 1 or getattr(args, "force_sharded", False):
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        else:
            # (rank 0 measures the sub-records alone while the others wait at the final barrier: minutes, not seconds)
            import datetime
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(minutes=45))
    return rank, world, local_rank, dist


def reduce_time_and_count(dist, total_t, total_it):
    import torch
    if dist is None:
        return total_t, total_it
    tt = torch.tensor([total_t], dtype=torch.float64, device="cuda")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    ti = torch.tensor([float(total_it)], dtype=torch.float64, device="cuda")
    dist.all_reduce(ti, op=dist.ReduceOp.SUM)
    return float(tt.item()), int(ti.item())


def main_posegraph(args):
    """BASELINE configs[4]: 5,000-keyframe global pose-graph relinearise + solve.  A step = one optimize4DoFPoseGraph
    pass (PoseGraph.cpp:226-385) over the whole graph from the drifted SVIn poses; the value is Levenberg-Marquardt
    iterations (relinearise + solve + candidate evaluation) per second of device time, inputs resident in HBM.  The
    path does not shard: N > 1 runs replicas."""
    import torch
    rank, world, local_rank, dist = init_distributed(args)
    out = posegraph_record(args, rank, world, local_rank, dist, args.steps, args.warmup, cpu=not args.no_cpu_baseline and world == 1)
    if rank == 0:
        emit(out)
    if dist is not None:
        dist.destroy_process_group()
    return out


def posegraph_record(args, rank, world, local_rank, dist, steps, warmup, cpu):
    import torch
    from svin_amd import synthetic_pg as spg
    from svin_amd.posegraph import PoseGraph
    spec = spg.make_pose_graph(n=5000, laps=20, loop_every=25, seed=7 + rank)

    def one_step():
        g = PoseGraph(local_rank, six_dof=args.six_dof)
        earliest, cur = spg.feed(g, spec)
        s = g.optimize(earliest, cur)
        part = g.partition()
        g.close()
        return s, part

    for _ in range(warmup):
        one_step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    total_t, total_it, dense_t, dense_n, last, part = 0.0, 0, 0.0, 0, None, None
    for _ in range(steps):
        last, part = one_step()
        total_t += last["solve_seconds"]
        total_it += last["iterations"]
        dense_t += part["dense_solve_seconds"]
        dense_n += part["dense_solves"]
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    total_t, total_it = reduce_time_and_count(dist, total_t, total_it)
    if rank != 0:
        return None
    value = total_it / total_t
    d = part["separator_unknowns"]
    flops = d ** 3 / 3.0 + 2.0 * d * d     # Cholesky + the two triangular solves of the separator system
    ach = flops / (dense_t / max(dense_n, 1)) / 1e12
    out = {
        "metric": "pose-graph Levenberg-Marquardt iterations/sec on a 5,000-keyframe loop-closure graph",
        "value": value, "unit": "LM iterations/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": 1e3 * total_t / steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "configs[4]: pose_graph loop closure, 5000 keyframes / %d loop edges, %s, seed 7, one "
                               "optimize pass per step" % (len(spec.loops), "6-DoF" if args.six_dof else "4-DoF"),
                   "iterations_per_step": total_it / (steps * world), "initial_cost": last["initial_cost"],
                   "final_cost": last["final_cost"], "partition": part, "parallelism": "replicas x%d" % world},
        "roofline": {"bound": "mfma", "achieved": ach, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": ach / F64_MFMA_PEAK_TFLOPS, "traffic": None, "kernel": "k_sb_factor / _forward / _load / _back (speed / bias chain, cyclic reduction) + k_big_chol_chain + k_big_back on the kept rows",
                     "launch_ms": 1e3 * dense_t / max(dense_n, 1), "flops_per_launch": flops,
                     "note": "dense root of %d unknowns (loop cover + level-2 cuts): latency-bound (serial 16-column "
                             "pivots), not throughput-bound; see DESIGN.md 9" % d},
    }
    if cpu:
        from oracle import orc
        c = orc.OraclePoseGraph(six_dof=args.six_dof, envelope=True)
        earliest, cur = spg.feed(c, spec)
        t0 = time.perf_counter()
        sc = c.optimize(earliest, cur)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = dict(value=sc["iterations"] / dt, unit="LM iterations/s", cores=1, kind="port",
                                   sample="one optimize pass of the same graph (%d iterations, %.1f s), oracle/ C++ "
                                          "restatement with an envelope Cholesky in natural order -- NOT Ceres' "
                                          "supernodal SuiteSparse factorisation, which would be markedly faster"
                                          % (sc["iterations"], dt))
        out["config"]["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
    return out


def window_record(name, spec, device, steps, warmup, iters):
    """one single-GPU sub-record: `steps` x optimize(iters) of a synthetic window, inputs resident"""
    import torch
    from svin_amd import synthetic as syn
    from svin_amd.estimator import Estimator
    est = Estimator(device)
    fids, lids = syn.feed(est, spec)
    snap = snapshot_init(est, fids, lids, spec)
    times, its, last = timed_solves(est, fids, lids, snap, steps, warmup, iters, torch.cuda.synchronize)
    facs = est.eval_factors() if spec.P <= 16 else []
    kinds = {}
    for f in facs:
        kinds[str(f["kind"])] = kinds.get(str(f["kind"]), 0) + 1
    rec = dict(workload=name, value=sum(its) / sum(times), unit="GN iterations/s", steps=steps,
               ms_per_iteration=1e3 * sum(times) / max(sum(its), 1), median_ms_per_step=1e3 * float(np.median(times)),
               iterations_per_step=sum(its) / steps, initial_cost=last["initial_cost"], final_cost=last["final_cost"],
               upload_ms=1e3 * last["upload_time"], P=spec.P, L=spec.L, N=spec.N)
    if kinds:
        rec["factor_kinds"] = kinds   # 0 imu 1 pose prior 2 speed/bias prior 3 relative extrinsics 4 sonar 5 depth
    # the kernels of one iteration on their own (HIP events on the solver's stream) and the roofline of the dominant ones
    try:
        ev, bu, so = est.bench_kernel_times(5)
        d = 6 * spec.P * (1 + (2 if "rig v2" in name else 0)) + 9 * spec.P     # poses (+ per-frame extrinsics) + speed/bias
        chol_flops = d ** 3 / 3.0 + 2.0 * d * d
        n_l = np.bincount(spec.obs_lm, minlength=spec.L).astype(float)
        schur_flops = float(np.sum(3.0 * (6.0 * n_l) ** 2))      # sum_l (6 n_l x 3)(3 x 6 n_l), lower triangle
        rec["kernel_ms"] = {"eval_reproj": ev, "build_normal_equations": bu, "chol_solve_backsub": so}
        rec["roofline"] = {
            "bound": "mfma", "unit": "TFLOP/s", "peak": F64_MFMA_PEAK_TFLOPS, "d": d,
            "solve": {"kernel": "k_chol_solve_lds" if d <= 176 else ("k_chol_solve_lds with border rows" if d <= 200 else "k_chol_solve_ll") if d <= 272 else ("k_sb_factor / _forward / _load / _back (speed / bias chain, cyclic reduction) + k_big_chol_chain + k_big_back on the kept rows"),
                      "launch_ms": so, "flops": chol_flops, "achieved": chol_flops / (so * 1e-3) / 1e12,
                      "frac": chol_flops / (so * 1e-3) / 1e12 / F64_MFMA_PEAK_TFLOPS},
            "schur": {"kernel": "k_schur_dense" if spec.P <= 20 else "k_panels_landmarks + k_blocks_slots + k_schur_rows + k_blocks_pose_reduce (+ small factors, slab sum)", "launch_ms": bu, "flops": schur_flops,
                      "achieved": schur_flops / (bu * 1e-3) / 1e12, "frac": schur_flops / (bu * 1e-3) / 1e12 / F64_MFMA_PEAK_TFLOPS},
            "note": "algorithmic flops (Cholesky d^3/3 + two triangular solves; Schur complement sum_l 3 (6 n_l)^2) over the launch time: "
                    "both are latency-bound chains at these sizes (DESIGN.md 3), the fractions say how far from the matrix pipes' rate"}
        rec["roofline"]["achieved"] = rec["roofline"]["solve"]["achieved"] if so >= bu else rec["roofline"]["schur"]["achieved"]
        rec["roofline"]["frac"] = rec["roofline"]["achieved"] / F64_MFMA_PEAK_TFLOPS
        rec["roofline"]["kernel"] = rec["roofline"]["solve"]["kernel"] if so >= bu else rec["roofline"]["schur"]["kernel"]
        rec["roofline"]["traffic"] = None
        if spec.P > 20:
            # MFMA flops k_schur_rows executes (rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 x 512) over the algorithmic count
            # for THIS window: measured under the profiler (tools/run_r06_profiles.sh), quoted when it describes the same window
            try:
                with open(os.path.join(ROOT, "profiles", "r06_config4_mfma.json")) as fh:
                    mf = json.load(fh)
                if abs(mf["algorithmic_flops_per_launch"] - schur_flops) < 1e-6 * schur_flops:
                    rec["roofline"]["executed_over_algorithmic"] = mf["executed_over_algorithmic"]
                    rec["roofline"]["executed_mfma_flops_per_launch"] = mf["executed_mfma_flops_per_launch"]
                    rec["roofline"]["executed_kernel"] = mf.get("kernel")
                    rec["roofline"]["executed_source"] = "profiles/r06_config4_mfma.json"
            except (OSError, KeyError, ValueError):
                pass
    except Exception as ex:
        rec["roofline"] = {"error": repr(ex)}
    return rec, est


def sliding_window_record(device, with_oracle=True, rig="euroc"):
    """SVIn's operating mode: a window fed frame by frame -- addStates, ~1 000 addObservation, optimize(10),
    applyMarginalizationStrategy(5 keyframes, 3 IMU frames) -- EVERY library call of a frame timed, host work and PCIe included,
    the oracle beside it.  Two passes: frames back to back (throughput: the device part of the marginalisation, which its call
    only enqueues, is waited for by whichever call of the next frame needs the stream first -- addStates) and frames arriving
    apart as in SVIn (20 Hz: the handle is idle when a frame arrives; what a front end sees as latency)."""
    from svin_amd import synthetic as syn
    from svin_amd.estimator import Estimator

    def run(est, spec, spaced=False):
        rows, timing = [], {}

        def on_frame(k, fid):
            t0 = time.perf_counter()
            est.optimize(10)
            t1 = time.perf_counter()
            ok, removed = est.apply_marginalization(5, 3)
            t2 = time.perf_counter()
            s = est.summary()
            if spaced:
                est.wait_idle()     # the next frame is 50 ms away: the marginalisation job has long run when it arrives
            rows.append(dict(optimize=t1 - t0, marginalise=t2 - t1, job=time.perf_counter() - t2, iterations=s["iterations"],
                             upload=s.get("upload_time", 0.0), solve=s.get("solve_time", 0.0), download=s.get("download_time", 0.0),
                             removed=len(removed)))
        syn.feed(est, spec, on_frame=on_frame, timing=timing)
        for r, a, b, c in zip(rows, timing["add_states_s"], timing["set_states_s"], timing["add_observations_s"]):
            r.update(add_states=a, set_states=b, add_observations=c)
            r["all"] = a + b + c + r["optimize"] + r["marginalise"]
        return rows[4:]      # steady state: the window is full from the fifth frame on
    spec = syn.make_window(P=24, L=2400, n_obs=24000, seed=7, rig=rig, keyframe_every=2, frame_dt=0.25,
                           **({"sonar": True, "depth": True} if rig == "rig_v2" else {}))
    calls = ("add_states", "set_states", "add_observations", "optimize", "marginalise")

    def summary(rows):
        med = lambda key: float(np.median([r[key] for r in rows]))   # noqa: E731
        out = {("%s_call" % c if c in ("optimize", "marginalise") else c): 1e3 * med(c) for c in calls}
        out.update(pack_upload=1e3 * med("upload"), device_solve=1e3 * med("solve"), read_back=1e3 * med("download"))
        return out
    est_b2b = Estimator(device)
    rows = run(est_b2b, spec)
    paths = est_b2b.path_counters()
    del est_b2b
    frame = float(np.mean([r["all"] for r in rows]))
    med = lambda key: float(np.median([r[key] for r in rows]))   # noqa: E731
    rec = dict(workload="sliding window (%s), 5 keyframes + 3 IMU frames, ~1000 new observations per frame, addStates + "
                        "addObservation + optimize(10) + applyMarginalizationStrategy per frame, %d steady-state frames"
                        % ("EuRoC stereo rig, fixed extrinsics" if rig == "euroc" else
                           "SVIn stereo_rig_v2: 2 cameras with variable extrinsics + sonar + depth", len(rows)),
               ms_per_frame=1e3 * frame, frames_per_s=1.0 / frame,
               ms=summary(rows),
               iterations_per_frame=float(np.mean([r["iterations"] for r in rows])), paths=paths,
               optimize_marginalise_add_observations_ms=1e3 * (med("optimize") + med("marginalise") + med("add_observations")),
               window="device-resident (svin_amd/csrc/resident.hpp): per frame the host sends the ~1000 new observation records, the "
                      "removed ones and the state tables; one kernel rebuilds the landmark-major table, the marginalisation job's tables "
                      "are gathered on the device, landmark points / qualities are fetched when asked for",
               note="ms_per_frame = mean over the steady-state frames of ALL library calls of a frame, frames back to back: "
                    "marginalise_call returns once the host policy has enqueued M1-M3, and add_states (which synchronises the stream for "
                    "its IMU propagation) is where the next frame waits for that job.  optimize_marginalise_add_observations_ms is the "
                    "sum rounds 3 and 4 quoted as the frame time (medians of those three calls only).")
    srows = run(Estimator(device), spec, spaced=True)
    sframe = float(np.median([r["all"] for r in srows]))
    smed = lambda key: float(np.median([r[key] for r in srows]))   # noqa: E731
    rec["frames_spaced"] = dict(ms_per_frame=1e3 * sframe, ms=summary(srows), host_ms_per_frame=1e3 * (sframe - smed("solve")),
                                marginalisation_job_ms=1e3 * (smed("marginalise") + smed("job")),
                                note="the handle idle when a frame arrives (svin_ba_wait_idle after the marginalisation, outside the "
                                     "frame's calls): ms_per_frame = median of all calls of a frame = what a 20 Hz front end waits for; "
                                     "marginalisation_job_ms = call + device job, which overlaps the front end's work on the next frame")
    if rig == "rig_v2":
        # the dominant kernel of this record's marginalisation job: the eigen-solve of the prior (M3).  Timed on a prior of this very
        # window's steady state (117 unknowns, tests/golden/prior_matrices.npz) through the solver's own entry point; the flops are
        # the textbook count of tridiagonalisation (4/3 n^3) + divide and conquer (~4/3 n^3 without deflation) + back-transformation
        # (2 n^3): a chain of n dependent steps and log2 n merge levels in ONE workgroup, latency-bound by construction
        try:
            A = np.load(os.path.join(ROOT, "tests", "golden", "prior_matrices.npz"))["rig_v2_n117"]
            n = A.shape[0]
            ms = min(Estimator.debug_sym_eig(A)[2] for _ in range(3))
            flops = (4.0 / 3.0 + 4.0 / 3.0 + 2.0) * n ** 3
            # bound = "latency": the textbook flops over the launch time say how long the dependent chain is, not how well a pipe is
            # used -- no `frac` against the MFMA peak is quoted for it
            rec["roofline"] = dict(bound="latency", kernel="k_marg_final_dc (eigen-solve of the prior, svin_amd/csrc/symeig.hpp)", n=n, launch_ms=ms,
                                   flops=flops, achieved=flops / (ms * 1e-3) / 1e12, peak=None, unit="TFLOP/s",
                                   frac=None, traffic=None,
                                   note="one workgroup, n dependent Householder steps + log2 n merge levels: latency-bound (DESIGN.md); "
                                        "the Jacobi solve of rounds 1-4 took 2.1 ms on this matrix.  In the steady state of this window "
                                        "the prior has full numerical rank and k_marg_final_chol (Cholesky factor + a certificate that the "
                                        "rank rule drops nothing, ~0.06 ms at this size) answers in its place; this solve runs when the "
                                        "certificate fails (rank-deficient priors of the first frames)")
        except Exception as ex:   # noqa: BLE001
            rec["roofline"] = {"error": repr(ex)}
    if with_oracle:
        from oracle import orc
        orows = run(orc.OracleEstimator(), spec)
        of = float(np.mean([r["all"] for r in orows]))
        rec["cpu_baseline"] = dict(ms_per_frame=1e3 * of, frames_per_s=1.0 / of, cores=1, kind="port",
                                   sample="the same %d frames (all calls) through oracle/ (1 thread)" % len(orows))
        rec["speedup_vs_cpu_baseline"] = of / frame
    return rec


def batched_record(device, sizes=(16, 32, 64), steps=12, warmup=3):
    """B independent configs[1] windows (their own seeds) through svin_ba_solve_prepared_batch: ONE launch sequence per trust-region
    round, the window as a grid dimension ("independent replicas processing different windows", on one GPU).  Per step
    every window starts from its own initial state, is packed and uploaded untimed (inputs resident in HBM), and the batch call is
    timed; aggregate = the iterations of all windows per step / that time.  `one_window` = the same measurement with B = 1 (the
    ordinary path: a batch of one is not batched)."""
    import torch
    from svin_amd import estimator as E
    from svin_amd import synthetic as syn
    from svin_amd.estimator import Estimator
    out = {"unit": "GN iterations/s", "workload": "B configs[1] windows (10 KF / 2000 landmarks / 20000 residuals, seeds 20250629 + 7 k), optimize(10) per step"}
    B_max = max(sizes)
    ws = []
    for k in range(B_max):
        spec = syn.make_window(seed=20250629 + 7 * k)
        est = Estimator(device)
        fids, lids = syn.feed(est, spec)
        ws.append((est, fids, lids, snapshot_init(est, fids, lids, spec)))
    for B in (1,) + tuple(sizes):
        times, its, nb = [], [], 0
        for k in range(warmup + steps):
            for est, fids, lids, snap in ws[:B]:
                reset_state(est, fids, lids, snap)
                est.prepare()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nb = E.solve_prepared_batch([w[0] for w in ws[:B]], 10)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            n_it = sum(w[0].summary()["iterations"] for w in ws[:B])
            for w in ws[:B]:
                w[0].finish()
            if k >= warmup:
                times.append(dt)
                its.append(n_it)
        rec = {"aggregate": sum(its) / sum(times), "windows_batched": nb, "ms_per_step": 1e3 * sum(times) / len(times),
               "iterations_per_step": sum(its) / len(its), "us_per_window_iteration": 1e6 * sum(times) / sum(its)}
        if B == 1:
            out["one_window"] = rec
        else:
            rec["x_one_window"] = rec["aggregate"] / out["one_window"]["aggregate"]
            out["B%d" % B] = rec
    return out


def concurrent_record(device, n_handles=8, steps=12):
    """SVIn runs the estimator and pose_graph side by side (okvis_ros/launch/svin_stereorig_v2.xml:17-34): one svin_ba handle
    working through the sliding window while one svin_pg handle optimises the configs[4] graph from another thread, each on its
    own stream -- and n_handles independent svin_ba handles solving configs[1] windows from n_handles threads (what one MI355X
    delivers in aggregate).  Times are the library's own (taken inside the calls: a Python thread returning from a call may
    wait for the interpreter lock, the device does not)."""
    import threading
    from svin_amd import synthetic as syn
    from svin_amd import synthetic_pg as spg
    from svin_amd.estimator import Estimator
    from svin_amd.posegraph import PoseGraph
    wspec = syn.make_window(P=24, L=2400, n_obs=24000, seed=7, rig="euroc", keyframe_every=2, frame_dt=0.25)
    pspec = spg.make_pose_graph(n=5000, laps=20, loop_every=25, seed=7)

    def slide(stop=None):
        est, solves = Estimator(device), []

        def on_frame(k, fid):
            est.optimize(10)
            s = est.summary()
            if k >= 4:
                solves.append(s["solve_time"] / max(s["iterations"], 1))
            est.apply_marginalization(5, 3)
        syn.feed(est, wspec, on_frame=on_frame)
        if stop is not None:
            stop.set()
        return float(np.median(solves))

    def posegraph(stop=None, reps=3):
        out = []
        while (stop is None and len(out) < reps) or (stop is not None and not (stop.is_set() and len(out) >= 1)):
            g = PoseGraph(device)
            earliest, cur = spg.feed(g, pspec)
            s = g.optimize(earliest, cur)
            g.close()
            out.append(s["solve_seconds"] / max(s["iterations"], 1))
            if len(out) > 50:
                break
        return float(np.median(out))
    alone_ba, alone_pg = slide(), posegraph()
    res, stop = {}, threading.Event()
    ta = threading.Thread(target=lambda: res.__setitem__("ba", slide(stop)))
    tb = threading.Thread(target=lambda: res.__setitem__("pg", posegraph(stop)))
    tb.start(); ta.start(); ta.join(); tb.join()
    rec = dict(pair=dict(sliding_window_ms_per_iteration={"alone": 1e3 * alone_ba, "beside_pose_graph": 1e3 * res["ba"],
                                                          "slowdown": res["ba"] / alone_ba},
                         pose_graph_ms_per_iteration={"alone": 1e3 * alone_pg, "beside_sliding_window": 1e3 * res["pg"],
                                                      "slowdown": res["pg"] / alone_pg}))
    # n independent windows of configs[1], one handle and one thread each
    spec = syn.make_window(seed=20250629)
    ests = []
    for _ in range(n_handles):
        e = Estimator(device)
        fids, lids = syn.feed(e, spec)
        ests.append((e, fids, lids, snapshot_init(e, fids, lids, spec)))

    # Every step: all threads reset and upload their window (Python, serialised by the interpreter lock), meet at a barrier,
    # then all call svin_ba_solve_prepared at once (the lock is released inside the call): the solves really overlap.
    gate = threading.Barrier(n_handles)
    walls = []

    def work(i, out, barrier):
        e, fids, lids, snap = ests[i]
        t_solve, its = 0.0, 0
        for k in range(steps + 2):
            reset_state(e, fids, lids, snap)
            e.prepare()
            if barrier is not None:
                barrier.wait()
                t0 = time.perf_counter()
            e.solve_prepared(10)
            if barrier is not None:
                barrier.wait()
                if i == 0 and k >= 2:
                    walls.append(time.perf_counter() - t0)
            s = e.summary()
            if k >= 2:
                t_solve += s["solve_time"]; its += s["iterations"]
        out[i] = (t_solve, its)
    one = {}
    work(0, one, None)
    many = {}
    th = [threading.Thread(target=work, args=(i, many, gate)) for i in range(n_handles)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    per = [many[i][1] / many[i][0] for i in range(n_handles)]
    its_step = sum(many[i][1] for i in range(n_handles)) / float(steps)
    rec["handles_%d" % n_handles] = dict(
        one_handle_alone=one[0][1] / one[0][0], per_handle_when_all_run=float(np.mean(per)), slowest_handle=float(min(per)),
        aggregate=its_step / float(np.median(walls)), unit="GN iterations/s",
        sum_of_per_handle_rates=float(sum(per)),
        note="every step all handles start optimize(10) of their own configs[1] window together (barrier) on their own threads "
             "and streams; aggregate = iterations of all handles per step / wall time from the barrier until the last handle has "
             "returned (median over the steps, Python barrier overhead included); per-handle rates from the time inside "
             "svin_ba_solve_prepared")
    return rec


def summary_of(out):
    """<= 1.5 KB: one number per sub-record (GN / LM iterations per second, milliseconds per frame, roofline fractions)"""
    def get(*path):
        cur = out
        for k in path:
            if not isinstance(cur, dict) or k not in cur:
                return None
            cur = cur[k]
        return round(cur, 4) if isinstance(cur, float) else cur
    return {
        "config2_its": get("value"), "cpu_baseline_its": get("cpu_baseline", "value"),
        "config3_its": get("config3", "value"), "config4_its": get("config4_single_gpu", "value"),
        "config4_executed_over_algorithmic": get("config4_single_gpu", "roofline", "executed_over_algorithmic"),
        "sharded_config4_its": get("sharded_config4", "value"),
        "sharded_allreduce_frac_of_link": get("sharded_config4", "allreduce_GBps", "frac_of_link"),
        "sharded_allreduce_share_of_iteration": get("sharded_config4", "allreduce_us", "share_of_iteration"),
        "sharded_k1_frac_per_gpu": get("sharded_config4", "k1_roofline_per_gpu", "frac"),
        "sharded_speedup_vs_one_gpu": get("sharded_config4", "speedup_vs_one_gpu"),
        "config5_its": get("config5", "value"), "config5_6dof_its": get("config5_6dof", "value"),
        "sliding_ms": get("sliding_window", "ms_per_frame"), "sliding_spaced_ms": get("sliding_window", "frames_spaced", "ms_per_frame"),
        "sliding_optimize_call_ms": get("sliding_window", "frames_spaced", "ms", "optimize_call"),
        "marg_job_ms": get("sliding_window", "frames_spaced", "marginalisation_job_ms"),
        "sliding_cpu_ms": get("sliding_window", "cpu_baseline", "ms_per_frame"),
        "sliding_rig_v2_ms": get("sliding_window_rig_v2", "ms_per_frame"),
        "sliding_rig_v2_spaced_ms": get("sliding_window_rig_v2", "frames_spaced", "ms_per_frame"),
        "marg_job_rig_v2_ms": get("sliding_window_rig_v2", "frames_spaced", "marginalisation_job_ms"),
        "sliding_rig_v2_cpu_ms": get("sliding_window_rig_v2", "cpu_baseline", "ms_per_frame"),
        "prior_eigen_solve_n117_ms": get("sliding_window_rig_v2", "roofline", "launch_ms"),
        "handles8_aggregate_its": get("concurrent", "handles_8", "aggregate"),
        "handles8_x": (round(out["concurrent"]["handles_8"]["aggregate"] / out["concurrent"]["handles_8"]["one_handle_alone"], 3)
                       if get("concurrent", "handles_8", "aggregate") and get("concurrent", "handles_8", "one_handle_alone") else None),
        "batched16_aggregate_its": get("batched", "B16", "aggregate"), "batched16_x": get("batched", "B16", "x_one_window"),
        "batched32_aggregate_its": get("batched", "B32", "aggregate"), "batched32_x": get("batched", "B32", "x_one_window"),
        "batched64_aggregate_its": get("batched", "B64", "aggregate"), "batched64_x": get("batched", "B64", "x_one_window"),
        "k1_frac_b2b": get("roofline", "frac"), "k1_frac_survey_bytes": get("roofline", "frac_survey_bytes"),
        "k1_frac_4GB": get("roofline", "replicas_1024", "frac"),
        "pcie_inclusive_its": get("config", "pcie_inclusive", "value"),
    }


XGMI_LINK_GBS = 153.0   # per xGMI link and direction (MI355X_MICROARCH.md: 7 links x ~153 GB/s per GPU, point to point)


def sharded_config4(rank, world, local_rank, dist, steps, warmup, iters=5):
    """ONE config-#4 window (64 KF / 50 000 landmarks / 500 000 residuals) with its landmarks split over the ranks: every
    rank holds all states, its landmark range and every world-th small factor; per iteration one RCCL all-reduce of
    [lower triangle of S | g | h] (d (d + 1) / 2 + 3 d doubles) and two small ones of trust-region scalars, all enqueued on
    the solver's stream (svin_ba_set_distributed_rccl)."""
    import torch
    from svin_amd import synthetic as syn
    from svin_amd import distributed as sd
    from svin_amd.estimator import Estimator
    spec = syn.make_window(P=64, L=50000, n_obs=500000, seed=20250629, frame_dt=0.25)
    mine = sd.shard_spec(spec, rank, world)
    est = Estimator(local_rank)
    fids, lids = syn.feed(est, mine)
    staged = dist.get_backend() != "nccl"     # --sharded-transport=stage: gloo, device buffers staged through host memory
    if staged:
        keep_alive = sd.make_torch_allreduce(device="stage")
        est.set_distributed(rank, world, keep_alive)
    else:
        sd.init_rccl(est, rank, world)
    snap = snapshot_init(est, fids, lids, mine)
    tdev = "cpu" if staged else "cuda"

    def sync():
        torch.cuda.synchronize()
        dist.barrier()
    times, its, last = timed_solves(est, fids, lids, snap, steps, warmup, iters, sync)
    t = torch.tensor(times, dtype=torch.float64, device=tdev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)          # per step: the slowest rank
    times = [float(v) for v in t.cpu()]
    d = 64 * 15
    n_sys = d * (d + 1) // 2 + 3 * d
    # the collectives on their own (HIP events on the solver's stream): the system message, and the scalar message
    # (the staged transport has no stream-ordered collective to time: the message costs are the gloo + PCIe round trips)
    ar_us = est.bench_allreduce(n_sys, 20) if not staged else float("nan")
    ar_small_us = est.bench_allreduce(24, 50) if not staged else float("nan")
    tt = torch.tensor([ar_us, ar_small_us], dtype=torch.float64, device=tdev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    ar_us, ar_small_us = float(tt[0]), float(tt[1])
    # NCCL's conventions: algorithm bandwidth = bytes / time; bus bandwidth = algbw x 2 (n - 1) / n, the rate every link of a
    # ring carries -- what the per-link xGMI figure bounds
    if staged:
        ar_us = ar_small_us = None
    algbw = 8.0 * n_sys / (ar_us * 1e-6) / 1e9 if ar_us else None
    busbw = (algbw * 2.0 * (world - 1) / world if world > 1 else 0.0) if algbw is not None else None
    # K1 on this rank's share (HBM-resident replicas of its observation set), like the headline roofline object
    k1_ms, k1_bytes = est.bench_jacobian_eval(16, 10)
    k1 = torch.tensor([k1_bytes / (k1_ms * 1e-3) / 1e9], dtype=torch.float64, device=tdev)
    dist.all_reduce(k1, op=dist.ReduceOp.MIN)
    ms_it = 1e3 * sum(times) / max(sum(its), 1)
    return dict(workload="configs[3]: ONE window, 64 KF / 50000 landmarks / 500000 residuals, landmarks sharded over %d GPUs, "
                         "small factors dealt frame by frame, optimize(%d) per step" % (world, iters),
                value=sum(its) / sum(times), unit="GN iterations/s", n_gpus=world, steps=steps, warmup=warmup, scaling="strong",
                ms_per_iteration=ms_it, median_ms_per_step=1e3 * float(np.median(times)),
                iterations_per_step=sum(its) / steps, final_cost=last["final_cost"], initial_cost=last["initial_cost"],
                landmarks_per_rank=mine.L, residuals_per_rank=mine.N,
                allreduce_bytes_per_iteration=8 * n_sys + 8 * (24 + 8),
                allreduce_us=({"system_message": ar_us, "scalar_message": ar_small_us, "per_iteration": ar_us + 2.0 * ar_small_us,
                               "share_of_iteration": (ar_us + 2.0 * ar_small_us) * 1e-3 / ms_it} if ar_us else None),
                allreduce_GBps={"algbw": algbw, "busbw": busbw, "xgmi_link_peak": XGMI_LINK_GBS,
                                "frac_of_link": busbw / XGMI_LINK_GBS if (world > 1 and busbw is not None) else None},
                k1_roofline_per_gpu={"achieved": float(k1[0]), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": float(k1[0]) / HBM_PEAK_GBS,
                                     "note": "slowest rank, 16 HBM-resident replicas of its observation share"},
                transport="gloo, device buffers staged through host memory (--sharded-transport=stage: all ranks may share one GPU)" if staged
                          else "RCCL",
                collective="ncclAllReduce (RCCL), FP64 sum, in place, on the solver's HIP stream; 3 per iteration: "
                           "[lower(S) | g | h], [8 dogleg sums | the ranks' (gradient max, failure flag) pairs], [8 cost / step sums | stop vote]")


def run_sharded_children(world, force_sharded, transport="rccl", steps=None):
    """starts `world` fresh ranks of `bench.py --sharded-child` and returns rank 0's record (or what went wrong)"""
    env = {k: v for k, v in os.environ.items()
           if not k.startswith("TORCHELASTIC") and k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                             "ROLE_RANK", "ROLE_WORLD_SIZE", "MASTER_PORT", "GROUP_WORLD_SIZE")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world == 1:
        env["SVIN_FORCE_DISTRIBUTED"] = "1"   # a one-rank communicator still takes the sharded code path
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__), "--sharded-child", "--gpus", str(world)]
    if force_sharded:
        cmd.append("--force-sharded")
    cmd += ["--sharded-transport", transport]
    if steps is not None:
        cmd += ["--steps", str(steps)]
    timeout = float(os.environ.get("SVIN_BENCH_SHARDED_TIMEOUT", "420"))
    # the launcher and its ranks in a process group of their own: on a timeout the whole group goes (killing the launcher alone
    # would leave its ranks spinning on the GPUs the remaining sub-records are measured on)
    import signal
    import types
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
    try:
        so, se = proc.communicate(timeout=timeout)
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
        return {"error": "the sharded ranks did not finish within %.0f s (process group killed)" % timeout}
    r = types.SimpleNamespace(stdout=so, stderr=se, returncode=proc.returncode)
    for line in r.stdout.decode(errors="replace").splitlines()[::-1]:
        if line.startswith("SHARDED_JSON:"):
            rec = json.loads(line[len("SHARDED_JSON:"):])
            if r.returncode != 0:
                rec["launcher_returncode"] = r.returncode
            return rec
    return {"error": "the sharded ranks ended with return code %d and no record" % r.returncode,
            "stderr_tail": r.stderr.decode(errors="replace")[-1500:]}


def main_sharded_child(args):
    """one rank of the sharded config-#4 sub-record (started by run_sharded_children)"""
    rank, world, local_rank, dist = init_distributed(args)
    if os.environ.get("SVIN_BENCH_KILL_RANK") == str(rank):   # tests/test_gpu_multigpu.py: a rank that dies before the first collective
        os._exit(17)
    try:
        rec = sharded_config4(rank, world, local_rank, dist, args.steps if args.steps != 30 else 20, 3)
    except Exception as ex:   # noqa: BLE001
        rec = {"error": repr(ex)}
    if rank == 0:
        os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, ("SHARDED_JSON:" + json.dumps(rec) + "\n").encode())
    if "error" in rec:
        os._exit(3)    # (no further collective: the other ranks may be anywhere)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--copies", type=int, default=256, help="window replicas for the HBM-resident Jacobian-eval roofline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the config3 / config4 / config5 / sharded sub-records")
    ap.add_argument("--workload", default="window", choices=["window", "posegraph"],
                    help="window = BASELINE configs[1] (the north-star metric, default); posegraph = configs[4], the "
                         "global pose-graph optimisation (SURVEY 8(f) N1), reported as its own line")
    ap.add_argument("--six-dof", action="store_true", help="posegraph: optimize6DoFPoseGraph instead of the 4-DoF one")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the sharded config-#4 sub-record even with one rank (one-rank RCCL communicator: every collective "
                         "really runs): the only way to exercise that code path on a 1-GPU box")
    ap.add_argument("--sharded-transport", default="rccl", choices=["rccl", "stage"],
                    help="collectives of the sharded config-#4 sub-record: rccl (native, on the solver's stream) or stage (gloo over "
                         "host-staged buffers: the ranks may then share one GPU -- how a 1-GPU box runs the N > 1 record's code)")
    ap.add_argument("--sharded-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.sharded_child:
        protect_stdout()
        if args.gpus == 1:
            args.force_sharded = True
        return main_sharded_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    protect_stdout()
    if args.workload == "posegraph":
        return main_posegraph(args)

    import torch
    rank, world, local_rank, dist = init_distributed(args)

    from svin_amd import synthetic as syn
    from svin_amd.estimator import Estimator

    spec = syn.make_window(seed=20250629 + rank)  # config #2 (each rank: its own seeded replica)
    est = Estimator(local_rank)
    fids, lids = syn.feed(est, spec)
    snap = snapshot_init(est, fids, lids, spec)

    timed_solves(est, fids, lids, snap, 0, args.warmup, 10, torch.cuda.synchronize)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t_region = time.perf_counter()
    times, its, last = timed_solves(est, fids, lids, snap, args.steps, 0, 10, torch.cuda.synchronize)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t_region = time.perf_counter() - t_region
    total_t, total_it = reduce_time_and_count(dist, sum(times), sum(its))

    out = None
    if rank == 0:
        value = total_it / total_t
        med = float(np.median(times))
        out = {
            "metric": "Gauss-Newton iterations/sec on 10-KF/2k-landmark window; Jacobian-eval HBM GB/s",
            "value": value, "unit": "GN iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * total_t / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1]: synthetic stereo+IMU 10 KF / 2000 landmarks / 20000 reprojection residuals "
                                   "/ 9 IMU factors, seed 20250629, optimize(10) per step",
                       "iterations_per_step": total_it / (args.steps * world), "final_cost": last["final_cost"],
                       "initial_cost": last["initial_cost"], "upload_ms": 1e3 * last["upload_time"],
                       "download_ms": 1e3 * last["download_time"], "parallelism": "replicas x%d" % world,
                       # median of >= 30 runs; `value` itself is total iterations / total time of the K steps
                       "median_ms_per_step": 1e3 * med, "value_at_median": (its[0] / med) * world,
                       "min_ms_per_step": 1e3 * min(times), "max_ms_per_step": 1e3 * max(times),
                       "timed_region_s_incl_untimed_upload": t_region,
                       # the boundary hands over host buffers: the same solves with pack + upload and the read-back of states,
                       # landmarks and qualities counted (never `value`, which is measured with the inputs resident in HBM)
                       "pcie_inclusive": {"value": its[-1] / (med + last["upload_time"] + last["download_time"]), "unit": "GN iterations/s",
                                          "upload_ms": 1e3 * last["upload_time"], "download_ms": 1e3 * last["download_time"]}},
        }
        # roofline of the dominant streaming kernel (K1 reprojection residual + Jacobian evaluation) on an
        # HBM-resident batch of replicas; HIP events on the kernel's own stream
        # `achieved` is taken from 20 launches enqueued BACK TO BACK between one pair of events (the write-back of launch i
        # overlaps launch i + 1: nothing of a launch's stores can still sit in the 256 MiB Infinity Cache when the clock stops,
        # except for the last of the twenty); the per-launch bracket of rounds 1-3 is reported beside it
        ms_each, ms, nbytes = est.bench_jacobian_eval_b2b(args.copies, 20)
        ach = nbytes / (ms * 1e-3) / 1e9
        ms1, nbytes1 = est.bench_jacobian_eval(1, 50)
        # a second point far beyond the Infinity Cache: 1 024 replicas = 4 GB of Jacobians per launch
        def device_copy_gbps(total_bytes):
            """a plain device-to-device copy moving `total_bytes` (half read, half written), back to back: what the memory
            system of this box gives a two-stream kernel at that footprint"""
            n = int(total_bytes // 16)
            a = torch.empty(n, dtype=torch.float64, device="cuda")
            b = torch.empty_like(a)
            for _ in range(3):
                b.copy_(a)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                b.copy_(a)
            e1.record()
            torch.cuda.synchronize()
            return 16.0 * n / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e9
        try:
            msb_each, msb, nbytesb = est.bench_jacobian_eval_b2b(1024, 10)
            big = {"replicas": 1024, "bytes_per_launch": nbytesb, "launch_ms_back_to_back": msb, "launch_ms_per_launch_events": msb_each,
                   "GBps": nbytesb / (msb * 1e-3) / 1e9, "frac": nbytesb / (msb * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "device_copy_same_footprint_GBps": device_copy_gbps(nbytesb),
                   "note": "beyond ~1.5 GB of footprint this box's memory system slows down for every kernel: a plain device copy "
                           "falls from ~5.4 (1 GB) to ~5.2 (4 GB) and ~4.4 TB/s (8 GB), K1 from 6.2-6.4 to 4.3; at 1 GB the 180 MB of "
                           "K1's inputs also stay in the 256 MiB Infinity Cache from launch to launch (tools/k1_sweep.py, DESIGN.md 5)"}
        except Exception as ex:   # noqa: BLE001
            big = {"error": repr(ex)}
        try:
            copy_1gb = device_copy_gbps(nbytes)
        except Exception:   # noqa: BLE001
            copy_1gb = None
        # HBM bytes per launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of
        # tools/k1_bench.py, gfx950 FETCH_SIZE x2 correction; committed summary profiles/*_k1_pmc.json).  Counters
        # cannot be read from inside this process, so the committed measurement is quoted when it describes the same
        # launch (same replica count and algorithmic bytes); otherwise null.
        traffic, traffic_from = None, None
        for name in ("r06_k1_pmc.json", "r05_k1_pmc.json", "r04_k1_pmc.json", "r02_k1_pmc.json", "r01_k1_pmc.json"):
            try:
                with open(os.path.join(ROOT, "profiles", name)) as fh:
                    pmc = json.load(fh)
                if abs(pmc["algorithmic_bytes_per_launch"] - nbytes) < 1e-6 * nbytes:
                    traffic = pmc["traffic_bytes_per_launch"]
                    traffic_from = "profiles/" + name
                    break
            except (OSError, KeyError, ValueError):
                pass
        n_res = spec.N * args.copies
        out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                           "traffic": traffic, "traffic_source": traffic_from, "kernel": "k_eval_reproj", "launch_ms": ms,
                           "timing": "20 launches back to back between one pair of HIP events on the kernel's stream",
                           "bytes_per_launch": nbytes, "replicas": args.copies,
                           "per_launch_events": {"launch_ms": ms_each, "achieved": nbytes / (ms_each * 1e-3) / 1e9,
                                                 "frac": nbytes / (ms_each * 1e-3) / 1e9 / HBM_PEAK_GBS},
                           "replicas_1024": big,
                           "frac_of_copy_ceiling": ach / HBM_COPY_CEILING_GBS,
                           "device_copy_same_footprint_GBps": copy_1gb,
                           # SURVEY 8(d) prices a residual at 191.2 B (fixed extrinsics); this layout stores the landmark
                           # index explicitly (+4 B) and `achieved` counts it -- the figure without it:
                           "achieved_survey_bytes": 191.2 * n_res / (ms * 1e-3) / 1e9,
                           "frac_survey_bytes": 191.2 * n_res / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "single_window_cache_resident": {"launch_ms": ms1, "GBps": nbytes1 / (ms1 * 1e-3) / 1e9}}
        ev, bu, so = est.bench_kernel_times(20)
        out["kernel_ms"] = {"eval_reproj": ev, "build_normal_equations": bu, "chol_solve_backsub": so}
        d = 150
        out["solver"] = {"kernel": "k_chol_solve_lds", "d": d, "launch_ms": so, "flops": d ** 3 / 3.0 + 2.0 * d * d,
                         "achieved_tflops": (d ** 3 / 3.0 + 2.0 * d * d) / (so * 1e-3) / 1e12, "peak_tflops": F64_MFMA_PEAK_TFLOPS,
                         "note": "latency-bound pivot chain at d = 150, see DESIGN.md"}
    del est
    if not args.no_extras:
        extras = {}
        if (world > 1 or args.force_sharded) and not os.environ.get("SVIN_BENCH_NO_SHARDED") and rank == 0:
            # The sharded window runs in ranks of its OWN (one fresh process per GPU, started from here): a collective that
            # never returns, an exception on one rank or a rank that dies -- under a launcher the death of any rank ends all
            # of them, headline line included -- then costs this sub-record and nothing else.  The ranks of the headline
            # measurement wait at the final barrier meanwhile (their GPUs are idle).
            extras["sharded_config4"] = run_sharded_children(world, args.force_sharded, args.sharded_transport)
        if rank == 0:
            try:
                spec3 = syn.make_window(P=10, L=4000, n_obs=40000, seed=20250629, rig="rig_v2", sonar=True, depth=True)
                extras["config3"], e3 = window_record("configs[2]: rig v2 stereo+IMU+sonar+depth, per-frame extrinsics, 10 KF / 4000 "
                                                      "landmarks / 40000 residuals, optimize(10) per step", spec3, local_rank, 10, 2, 10)
                del e3
                spec4 = syn.make_window(P=64, L=50000, n_obs=500000, seed=20250629, frame_dt=0.25)
                extras["config4_single_gpu"], e4 = window_record("configs[3] on ONE GPU: 64 KF / 50000 landmarks / 500000 residuals "
                                                                 "(d = 960), optimize(5) per step", spec4, local_rank, 5, 1, 5)
                del e4, spec4
                if "value" in extras.get("sharded_config4", {}):
                    extras["sharded_config4"]["speedup_vs_one_gpu"] = extras["sharded_config4"]["value"] / extras["config4_single_gpu"]["value"]
            except Exception as ex:   # a sub-record must never take the headline line down
                extras["error"] = repr(ex)
        if rank == 0:
            try:
                extras["sliding_window"] = sliding_window_record(local_rank, with_oracle=not args.no_cpu_baseline)
                extras["sliding_window_rig_v2"] = sliding_window_record(local_rank, with_oracle=not args.no_cpu_baseline, rig="rig_v2")
            except Exception as ex:
                extras["sliding_window"] = {"error": repr(ex)}
        if rank == 0:
            try:
                extras["concurrent"] = concurrent_record(local_rank)
            except Exception as ex:
                extras["concurrent"] = {"error": repr(ex)}
            try:
                extras["batched"] = batched_record(local_rank)
            except Exception as ex:
                extras["batched"] = {"error": repr(ex)}
        if rank == 0:
            try:
                pg = posegraph_record(argparse.Namespace(six_dof=False), 0, 1, local_rank, None, 2, 1, cpu=not args.no_cpu_baseline)
                extras["config5"] = {k: pg[k] for k in ("value", "unit", "ms_per_step", "cpu_baseline") if k in pg}
                if "speedup_vs_cpu_baseline" in pg["config"]:
                    extras["config5"]["speedup_vs_cpu_baseline"] = pg["config"]["speedup_vs_cpu_baseline"]
                extras["config5"].update(workload=pg["config"]["workload"], iterations_per_step=pg["config"]["iterations_per_step"],
                                         ms_per_iteration=pg["ms_per_step"] / pg["config"]["iterations_per_step"])
            except Exception as ex:
                extras["config5"] = {"error": repr(ex)}
            try:
                pg6 = posegraph_record(argparse.Namespace(six_dof=True), 0, 1, local_rank, None, 2, 1, cpu=not args.no_cpu_baseline)
                extras["config5_6dof"] = {k: pg6[k] for k in ("value", "unit", "ms_per_step", "cpu_baseline") if k in pg6}
                if "speedup_vs_cpu_baseline" in pg6["config"]:
                    extras["config5_6dof"]["speedup_vs_cpu_baseline"] = pg6["config"]["speedup_vs_cpu_baseline"]
                extras["config5_6dof"].update(workload=pg6["config"]["workload"], iterations_per_step=pg6["config"]["iterations_per_step"],
                                              ms_per_iteration=pg6["ms_per_step"] / pg6["config"]["iterations_per_step"])
            except Exception as ex:
                extras["config5_6dof"] = {"error": repr(ex)}
            out.update(extras)
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(syn.make_window(seed=20250629))
            out["config"]["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        out["summary"] = summary_of(out)   # LAST key: the headline of every sub-record survives a truncated tail of this line
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
