#!/usr/bin/env python
"""bench.py — sliding-window solves/sec on the BASELINE.json workload.

One "step" = one Estimator::SolveOptimization (BuildLocalMap + kNN/plane features + newest-frame GN +
<= 10 dogleg iterations + marginalization; Estimator.cc:1648-2438) on a steady-state window snapshot of
synthetic HDL-64E data (64 rings, ~133 k points/scan), window_size 15 / opt_window_size 5, with the
clouds already resident in HBM.

N > 1 (default): every rank owns an independent window (weak scaling, no data-path collective — SURVEY.md
§8e: the all-reduce of normal equations does not pay at this factor count); value = N * K / max-over-ranks
time.  `--shard-factors` instead solves ONE window cooperatively (factors sharded, RCCL all-reduce of the
moments per linearisation; strong scaling).

Prints ONE JSON line (rank 0).  Extra objects: `roofline` for the dominant kernel (HIP events on the
estimator's stream), `cpu_baseline` = the CPU oracle timed on this box's host cores on the same window,
`batched` = lio_est_batch: B in {8, 64, 512} copies of the window solved per launch chain — solves/s and the
device time, algorithmic bytes and roofline fraction of every stage (the single-window path is latency-bound; this is the path's
throughput mode),
`keyframe_batch` = BASELINE.json configs[4] at --keyframes keyframes (N = 1 only).

`--workload keyframes` makes configs[4] the bench line itself: a step = one refinement of all --keyframes keyframes,
the keyframe list sharded over the ranks, one all-gather of the poses per step (strong scaling).

Launching.  `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment re-executes itself as
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same flags>`
(the command the driver uses), after checking that N GPUs are visible.  Under a launcher, WORLD_SIZE must equal --gpus.
At N > 1 the ONE line carries, besides the weak-scaling value, `sharded` (one window, factors sharded, in-library RCCL
all-reduce per linearisation: solves/s and microseconds per all-reduce) and `keyframes` (configs[4], strong scaling,
in-library all-gather), each with `rccl_world` as the communicator reports it.
`--dry-launch` runs the same multi-rank control flow on CPU hosts: gloo process group, the CPU oracle as the worker, a small
VLP-16 window (tests/test_distributed_gloo.py spawns it with --gpus 2).
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "lio-mapping_amd"))

import numpy as np  # noqa: E402

EXTRA_FRAMES = 4


def make_dataset(kind, W, seed_shift=0.0):
    from lio_amd import synth

    frame_dt = 0.3 if kind == "outdoor" else 0.2  # odom_io = 3 (HDL-64) / 2 (VLP-16) x 0.1 s
    return synth.make_dataset(kind, W + 1 + EXTRA_FRAMES, frame_dt, t0=1.0 + seed_shift)


def feature_clouds(lib, ds):
    """PointProcessor on every sweep -> [(less_flat 'surf_last', less_sharp 'corner_last')], per-scan wall ms."""
    from lio_amd import capi

    pp = capi.PointProcessor(lib, ds.lidar.lower_deg, ds.lidar.upper_deg, ds.lidar.rings)
    clouds, ms = [], []
    for f in ds.frames:
        t = time.perf_counter()
        pp.process(f.scan)
        ms.append((time.perf_counter() - t) * 1e3)
        clouds.append((pp.cloud(4), pp.cloud(2)))
    return clouds, ms


def est_config(lib, ds, kind, W, Wo):
    from lio_amd import pipeline

    cfg = pipeline.config_outdoor64(lib, W, Wo) if kind == "outdoor" else pipeline.config_indoor(lib, W, Wo)
    if kind != "outdoor":
        cfg.cutoff_deskew, cfg.keep_features, cfg.prior_factor = 1, 0, 1
    pipeline.set_extrinsic(cfg, ds)
    return cfg


def make_estimator(lib, ds, clouds, kind, W, Wo):
    """Window initialised from ground truth + noise, EXTRA_FRAMES-1 frames fed through ProcessLaserOdom so a
    marginalization prior exists, then the last frame pushed (upload + VoxelGrid + window push) — the state right
    before the SolveOptimization under test — and snapshotted."""
    from lio_amd import capi, pipeline

    cfg = est_config(lib, ds, kind, W, Wo)
    est = capi.Estimator(lib, cfg)
    pipeline.init_window(est, lib, ds, [c[0] for c in clouds], pos_sigma=0.01, rot_sigma=0.001, vel_sigma=0.01)
    est.solve()
    est.slide()
    n_frames = len(ds.frames)
    for k in range(W + 1, n_frames - 1):
        pipeline.feed_frame(est, ds, k, clouds[k][0], clouds[k][1])
    k = n_frames - 1
    f = ds.frames[k]
    for j in range(f.imu_dt.shape[0]):
        est.process_imu(float(f.imu_dt[j]), f.imu_acc[j], f.imu_gyr[j], float(f.imu_t[j]))
    T = capi.TransformF.make([0, 0, 0, 1], [0, 0, 0])
    est.push_frame(T, clouds[k][0], clouds[k][1], f.t)
    est.snapshot()
    return est


def one_step(est):
    est.restore()
    return est.solve()


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: start one rank per GPU with the command the driver uses and pass the
    flags through.  Fails loudly when the node has fewer than N GPUs (unless --dry-launch, which needs none)."""
    import socket
    import subprocess

    if not args.dry_launch:
        import torch

        n = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {n} GPU(s) visible on this node")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver
    return subprocess.run(cmd, env=env).returncode


def _shared_window(lib, kind, W, Wo):
    """The SAME window on every rank (shift 0): the unit of the factor-sharded mode."""
    ds = make_dataset(kind, W, 0.0)
    clouds, _ = feature_clouds(lib, ds)
    return make_estimator(lib, ds, clouds, kind, W, Wo)


def sharded_solve_stats(lib, kind, W, Wo, rank, world, steps, sync, device, rccl=None, allreduce_numpy=None):
    """ONE window solved cooperatively by all ranks (MarginalizationFactor.cc:245-269 across ranks): every rank
    evaluates its share of the lidar factors, the per-shard moments (Wo x 260 doubles) are summed by one all-reduce per
    linearisation — in-library RCCL on the estimator's stream (GPU) or the gloo callback (CPU rehearsal).  Collective: every rank
    calls it.  Returns the dict rank 0 prints (other ranks: None)."""
    from lio_amd import dist_util

    est = _shared_window(lib, kind, W, Wo)
    rep0 = one_step(est)                       # unsharded reference solve of the same window (every rank, no collective)
    own = None
    if allreduce_numpy is None:
        own = rccl = rccl or dist_util.make_rccl(lib, rank, world)
        est.set_factor_sharding_rccl(rccl)
    else:
        est.set_factor_sharding(rank, world, allreduce_numpy)
    for _ in range(2):
        rep = one_step(est)
    dist_util.barrier(world)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        rep = one_step(est)
    est.sync()
    sync()
    dist_util.barrier(world)
    dt = dist_util.max_over_ranks(time.perf_counter() - t0, world, device=device)
    us = rccl.bench_all_reduce(Wo * 260, 200) if own is not None or rccl is not None else None
    out = {
        "mode": f"1 window, lidar factors sharded over {world} ranks, one SUM all-reduce of {Wo} x 260 doubles per linearisation",
        "value": round(steps / dt, 3), "unit": "solves/s", "scaling": "strong", "steps": steps, "ms_per_step": round(1e3 * dt / steps, 4),
        "rccl_world": rccl.world_seen_by_rccl() if rccl is not None else None,
        "allreduce": "in-library ncclAllReduce on the estimator's stream" if rccl is not None else "gloo callback (CPU rehearsal)",
        "us_per_allreduce": round(us, 2) if us is not None else None,
        "linearisations_per_solve": int(rep.iterations) + 1,
        "solver_iterations": int(rep.iterations), "solver_iterations_unsharded": int(rep0.iterations),
        "final_cost_rel_gap_to_unsharded": float(abs(rep.final_cost - rep0.final_cost) / max(abs(rep0.final_cost), 1e-300)),
    }
    est.set_factor_sharding_rccl(None) if allreduce_numpy is None else est.set_factor_sharding(0, 1, None)
    return out if rank == 0 else None


def keyframes_measure(lib, rank, world, n_kf, steps, sync, device, rccl=None, use_library_gather=True, kind="outdoor", warmup=1):
    """configs[4]: a step = one refinement of ALL n_kf keyframes (each against its own local map); rank r owns keyframes r, r + N, ...;
    the one exchange is an all-gather of the refined poses — ncclAllGather inside the library from the device pose buffer, or the
    torch collective (CPU rehearsal).  Collective.  Returns the measurement dict on rank 0 (None elsewhere)."""
    from lio_amd import capi, dist_util, synth

    n_kf = max(n_kf, world)
    ds = make_dataset(kind, 15 if kind == "outdoor" else 4)      # the same scans on every rank
    clouds, _ = feature_clouds(lib, ds)
    captured = []
    mapping_ms_per_scan(lib, ds, clouds, n_frames=8 if kind == "outdoor" else 4, capture=captured)
    rng = np.random.default_rng(5)
    n_src = len(captured)
    mine = list(range(rank, n_kf, world))
    alg, T0s, maps, kfs = [], [], [], []
    for k in range(n_kf):                                 # every rank draws all initial poses: identical lists everywhere
        c = captured[k % n_src]
        q, p = c["T"]
        R = synth.rot_from_quat(np.asarray(q, np.float64)) @ synth.small_rot(rng.uniform(-0.01, 0.01, 3))
        T0s.append((synth.quat_from_rot(R), np.asarray(p, np.float64) + rng.uniform(-0.15, 0.15, 3)))
    per_rank = -(-n_kf // world)
    if world > 1 and use_library_gather and rccl is None:
        rccl = dist_util.make_rccl(lib, rank, world)
    b = capi.KeyframeBatch(lib)
    for j, k in enumerate(mine):
        c = captured[k % n_src]
        b.add_map(c["corner_map"], c["surf_map"])
        b.add_keyframe(j, c["corner"], c["surf"], T0s[k])
        M, N = c["corner"].shape[0] + c["surf"].shape[0], c["corner_map"].shape[0] + c["surf_map"].shape[0]
        alg.append(16 * (M + N) + 72 * M)

    def step():
        if world > 1 and rccl is not None:   # refinement + ncclAllGather of the poses from the device pose buffer, both inside the library
            g = b.refine_gather(rccl, per_rank)
            mine_rows = g[rank, : len(mine)]
            return dict(q=mine_rows[:, 0:4], p=mine_rows[:, 4:7], iterations=mine_rows[:, 7].astype(np.int32), device_ms=b.last_device_ms)
        r = b.refine()
        if world > 1:                        # CPU rehearsal: the same exchange through the torch collective
            import torch
            import torch.distributed as dist

            buf = torch.zeros((per_rank, 9), dtype=torch.float32)
            if mine:
                buf[: len(mine), 0:4], buf[: len(mine), 4:7] = torch.from_numpy(r["q"]), torch.from_numpy(r["p"])
                buf[: len(mine), 7] = torch.from_numpy(r["iterations"].astype(np.float32))
            dist.all_gather([torch.empty_like(buf) for _ in range(world)], buf)
        return r

    for _ in range(max(warmup, 1)):
        r = step()
    dist_util.barrier(world)
    sync()
    t0 = time.perf_counter()
    dev_ms = []
    for _ in range(steps):
        r = step()
        dev_ms.append(r["device_ms"])
    sync()
    dist_util.barrier(world)
    dt_max = dist_util.max_over_ranks(time.perf_counter() - t0, world, device=device)
    if rank != 0:
        return None
    alg_bytes = float(np.dot(np.asarray(alg, np.float64), r["iterations"].astype(np.float64)))
    ms = float(np.median(dev_ms))
    return {
        "n_kf": n_kf, "dt_max": dt_max, "steps": steps, "alg_bytes": alg_bytes, "device_ms": ms, "captured": captured,
        "iterations_mean": round(float(r["iterations"].mean()), 2),
        "line": {
            "value": round(n_kf * steps / dt_max, 1), "unit": "keyframes/s", "scaling": "strong", "steps": steps, "ms_per_step": round(1e3 * dt_max / steps, 3),
            "keyframes": n_kf, "rccl_world": rccl.world_seen_by_rccl() if rccl is not None else None,
            "exchange": ("in-library ncclAllGather of 9 floats per keyframe" if rccl is not None else "torch all_gather (CPU rehearsal)") if world > 1 else "none (1 rank)",
            "parallelism": f"keyframes sharded over {world} ranks, all-gather of the poses" if world > 1 else "1 batch",
        },
    }


def throughput_measure(lib, cfg, est0, B, steps, warmup, rank, world, sync, device):
    """The throughput mode of the metric: every rank solves B windows per step through lio_est_batch (copies of ITS window at distinct
    addresses, one launch per stage over all of them; no data-path collective — the windows are independent), EXACTLY `steps` steps
    between barrier + synchronize on both sides, max over ranks -> world x B x steps / time.  Parity gate: the B windows of a rank are
    identical inputs, so every stage of every window must have left identical bits (lio_est_batch_stage_digest) and identical reports;
    a rank where they differ voids the number (returned as parity "broken", value None)."""
    from lio_amd import capi, dist_util

    clones = []
    for _ in range(B):
        e = capi.Estimator(lib, cfg)
        e.copy_snapshot_of(est0)
        e.restore()
        clones.append(e)
    batch = capi.EstimatorBatch(lib, clones)
    batch.solve_restored(max(1, warmup))
    dist_util.barrier(world)
    sync()
    t0 = time.perf_counter()
    reps = batch.solve_restored(steps)       # (ends with a wait for the last step's marginalizations)
    sync()
    dist_util.barrier(world)
    dt = dist_util.max_over_ranks(time.perf_counter() - t0, world, device=device)
    same = all((r.iterations, r.successful_steps, r.termination, r.n_lidar_residuals, r.final_cost) ==
               (reps[0].iterations, reps[0].successful_steps, reps[0].termination, reps[0].n_lidar_residuals, reps[0].final_cost) for r in reps)
    differing = []
    for s_idx, s_name in enumerate(capi.EstimatorBatch.STAGES):
        d = batch.stage_digest(s_idx)
        if bool((d != d[0]).any()):
            differing.append(s_name)
    broken = dist_util.max_over_ranks(0.0 if (same and not differing) else 1.0, world, device=device) > 0.0
    if B >= 96:   # (stage times of a step solved as one part: two parts' stages overlap, see batched_windows)
        batch.set_option("parts", 1)
        batch.solve_restored(3)
    clk = batch.clock()
    batch.close()
    out = {"windows_per_rank": B, "ranks": world, "parts": 2 if B >= 96 else 1, "steps": steps, "ms_per_step": round(1e3 * dt / steps, 4),
           "parity": "broken" if broken else "ok", "stages_that_differ_between_windows_on_rank_0": differing,
           "value": None if broken else round(world * B * steps / dt, 1), "unit": "solves/s",
           "solver_iterations": int(reps[0].iterations), "n_lidar_residuals": int(reps[0].n_lidar_residuals),
           "windows_on_device_loop_rank_0": int(clk["n_device"]),
           "rank_0_stage_device_ms": {k: round(clk[k], 4) for k in ("dev_filter", "dev_grid", "dev_features", "dev_rounds", "dev_loop", "dev_marg", "dev_marg_wait")},
           "note": "lio_est_batch: every rank solves `windows_per_rank` copies of its own window per step, one launch per stage over all of them, trust-region loop and "
                   "marginalization on the device; a step = restore every window + lio_est_batch_solve, looped inside the library; no collective on the data path"}
    return out


def dry_run(args, rank, world, torch, dist):
    """CPU rehearsal of the multi-rank run (no GPU, nothing measured): gloo process group, the CPU oracle behind the same C-ABI
    as the worker, a small VLP-16 window.  Exercises exactly the control flow of the real run: per-rank windows between
    barriers + max-over-ranks, the factor-sharded solve with an all-reduce per linearisation, the sharded keyframe batch with
    its all-gather, one JSON line from rank 0."""
    from lio_amd import dist_util

    if world > 1:
        dist.init_process_group("gloo")
    lib = _oracle_lib()
    kind, W, Wo = "indoor", 6, 3
    ds = make_dataset(kind, W, dist_util.window_shift_for_rank(rank))
    clouds, _ = feature_clouds(lib, ds)
    est = make_estimator(lib, ds, clouds, kind, W, Wo)
    steps = max(1, min(args.steps, 2))
    rep = one_step(est)
    dist_util.barrier(world)
    t0 = time.perf_counter()
    for _ in range(steps):
        rep = one_step(est)
    est.sync()
    dist_util.barrier(world)
    dt = dist_util.max_over_ranks(time.perf_counter() - t0, world, device="cpu")
    sharded = keyframes = None
    thr = throughput_measure(lib, est_config(lib, ds, kind, W, Wo), est, 2, steps, 1, rank, world, sync=lambda: None, device="cpu")   # the N > 1 headline's control flow
    if world > 1:
        sharded = sharded_solve_stats(lib, kind, W, Wo, rank, world, steps, sync=lambda: None, device="cpu", allreduce_numpy=dist_util.make_allreduce("cpu"))
        km = keyframes_measure(lib, rank, world, 2 * world, 1, sync=lambda: None, device="cpu", use_library_gather=False, kind="indoor")
        keyframes = km["line"] if km else None
    if rank == 0:
        print(json.dumps({
            "dry_launch": True, "backend": lib.backend, "process_group": "gloo" if world > 1 else None,
            "metric": "CPU rehearsal of the multi-rank control flow (NOT a measurement)",
            "value": thr["value"] if world > 1 else round(world * steps / dt, 3), "unit": "solves/s",
            "n_gpus": world, "steps": steps, "warmup": 1, "ms_per_step": thr["ms_per_step"] if world > 1 else round(1e3 * dt / steps, 3), "higher_is_better": True, "scaling": "weak",
            "throughput_mode": thr, "single_window": {"value": round(world * steps / dt, 3), "ms_per_step": round(1e3 * dt / steps, 3)},
            "vs_baseline": None, "dtype": "f64 (solve) / f32 (features)", "data": "synthetic",
            "config": {"workload": f"VLP-16 indoor, window_size={W} opt_window_size={Wo}, CPU oracle", "n_lidar_residuals": int(rep.n_lidar_residuals),
                       "parallelism": f"{world} independent windows" if world > 1 else "1 window"},
            "sharded": sharded, "keyframes": keyframes,
        }))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="hdl64", choices=["hdl64", "vlp16", "keyframes"],
                    help="hdl64 (default, the headline metric) / vlp16: sliding-window solves.  keyframes: BASELINE.json configs[4], "
                         "--keyframes HDL-64 keyframes refined per step, the keyframe list sharded over the ranks + all-gather of the poses")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--min-seconds", type=float, default=0.5,
                    help="the timed --steps block is repeated until the timed region covers at least this long; the line reports the median block")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two child rocprofv3 --pmc passes that measure roofline.traffic live")
    ap.add_argument("--pmc-child", default=None, help=argparse.SUPPRESS)   # internal: a few solves on the pickled workload, run under rocprofv3
    ap.add_argument("--cpu-steps", type=int, default=8)
    ap.add_argument("--windows", default="8,64,512", help="batch sizes of the `batched` extra (lio_est_batch: B windows per launch chain); 0 = skip")
    ap.add_argument("--rank-windows", type=int, default=512, help="N > 1: windows per rank and step of the throughput mode (lio_est_batch) the line's value is measured in; 0 = one window per rank")
    ap.add_argument("--keyframes", type=int, default=1000, help="keyframes of the batched-refinement extra (configs[4]); 0 = skip")
    ap.add_argument("--shard-factors", action="store_true",
                    help="strong-scaling mode: ONE window, its lidar factors sharded over the ranks, RCCL all-reduce of the normal-"
                         "equation moments per linearisation (SURVEY.md §8e).  Default: one independent window per rank, no collective.")
    ap.add_argument("--no-fed", action="store_true", help="skip the fed-GPU points (B keyframes / sweeps / clouds at once) of the stages besides the moments kernel")
    ap.add_argument("--dry-launch", action="store_true",
                    help="CPU rehearsal of the multi-rank run: gloo + the CPU oracle on a small VLP-16 window (no GPU needed, nothing measured)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))          # one rank per GPU under torch.distributed.run, same flags

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: launch one rank per GPU (or run `python bench.py --gpus N` "
                         "without a launcher and let it start the ranks itself)")
    if args.dry_launch:
        return dry_run(args, rank, world, torch, dist)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from lio_amd import capi, dist_util

    hip = capi.load_hip()
    if args.workload == "keyframes":
        keyframes_workload(args, hip, rank, world, torch, dist)
        if world > 1:
            dist.destroy_process_group()
        return
    kind = "outdoor" if args.workload == "hdl64" else "indoor"
    W, Wo = 15, 5
    if args.pmc_child:   # child of measure_pmc_traffic(): same window, a few solves, nothing printed
        import pickle

        with open(args.pmc_child, "rb") as fh:
            ds, clouds = pickle.load(fh)
        est = make_estimator(hip, ds, clouds, kind, W, Wo)
        for _ in range(args.steps):
            one_step(est)
        est.sync()
        if kind == "outdoor":   # the keyframe-batch kernels under the same counters (64 keyframes, each with its own local map)
            captured = []
            mapping_ms_per_scan(hip, ds, clouds, capture=captured)
            keyframe_batch_stats(hip, captured, 64, reps=2)
        return
    t_setup = time.time()
    ds = make_dataset(kind, W, 0.0 if args.shard_factors else dist_util.window_shift_for_rank(rank))
    clouds, pp_ms = feature_clouds(hip, ds)
    est = make_estimator(hip, ds, clouds, kind, W, Wo)
    setup_s = time.time() - t_setup
    if args.shard_factors and world > 1:
        # the all-reduce of the per-shard moments runs INSIDE the library: ncclAllReduce on the estimator's stream (RCCL over xGMI)
        rccl = dist_util.make_rccl(hip, rank, world)
        est.set_factor_sharding_rccl(rccl)
        assert rccl.world_seen_by_rccl() == world
    new_stack_n = est.get_surf_stack(W).shape[0]

    for _ in range(args.warmup):
        rep = one_step(est)

    def timed_block():
        """EXACTLY --steps steps between barrier + synchronize on both sides; max over ranks."""
        dist_util.barrier(world)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = est.solve_restored(args.steps)   # --steps x (restore + SolveOptimization), looped inside the library (no interpreter between two solves)
        est.sync()                 # the last solve's deferred marginalization belongs to the timed region
        torch.cuda.synchronize()
        dist_util.barrier(world)
        return dist_util.max_over_ranks(time.perf_counter() - t0, world, device="cuda"), r

    # A block of --steps steps is tens of milliseconds: it is repeated (same bracket every time) until the timed region covers
    # --min-seconds, and the line reports the MEDIAN block (box-to-box and block-to-block spread: `timing`).  No HIP events are
    # recorded inside the timed blocks; the per-kernel times come from a separate block afterwards.
    first, rep = timed_block()
    n_blocks = int(max(1, min(200, np.ceil(args.min_seconds / max(first, 1e-6)))))
    n_blocks = int(dist_util.max_over_ranks(float(n_blocks), world, device="cuda"))
    block_s = [first]
    for _ in range(n_blocks - 1):
        b, rep = timed_block()
        block_s.append(b)
    dt_max = float(np.median(block_s))
    value = world * args.steps / dt_max
    if args.shard_factors:
        value = args.steps / dt_max  # one window solved cooperatively: total work is fixed
    # N > 1 (one rank per GPU, independent windows): the line's value is the THROUGHPUT mode — every rank solves --rank-windows windows per
    # step through lio_est_batch — because that is what a node full of GPUs is for; N ranks x one latency-bound window each is carried as
    # `single_window`.  At N = 1 the headline stays the single window and the same mode is the `batched` extra.
    thr = None
    if world > 1 and not args.shard_factors and args.rank_windows > 0:
        thr = throughput_measure(hip, est_config(hip, ds, kind, W, Wo), est, args.rank_windows, args.steps, args.warmup, rank, world, sync=torch.cuda.synchronize, device="cuda")

    resident = est.kernel_timing("moments_resident")   # passes served by the resident moments kernel so far (the timed blocks above)
    est.enable_kernel_timing(-1)                       # untimed block: HIP events around the resident kernel's launches (dispatch -> exit)
    for _ in range(max(5, args.steps // 5)):
        one_step(est)
    est.sync()
    resident_launch = est.kernel_timing("moments_resident_launch")
    est.enable_kernel_timing(False)
    names = ["features", "odom_features", "odom_rows", "odom_update", "moments", "voxel", "knn_grid", "concat"]
    est.enable_kernel_timing(1)   # HIP events around EVERY launch of each kernel kind on the estimator's stream: untimed block
    for _ in range(max(5, args.steps // 5)):
        one_step(est)
    est.sync()
    kt = {n: est.kernel_timing(n) for n in names}
    est.enable_kernel_timing(False)

    if rank == 0:
        def per_launch(n):
            k = kt[n]
            L = max(k["launches"], 1)
            avg_ms = k["total_ms"] / L
            gbps = (k["algorithmic_bytes"] / L) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            return avg_ms, gbps, k["algorithmic_bytes"] / L

        dom = max(("features", "odom_features", "moments", "odom_rows"), key=lambda n: kt[n]["total_ms"])
        avg_ms, achieved, alg_bytes = per_launch(dom)
        kernel_of = {"features": "k_features", "odom_features": "k_odom_round", "moments": "k_lidar_moments", "odom_rows": "k_odom_rows"}
        roofline = {
            "kernel": kernel_of[dom] + (" + k_moment_reduce" if dom == "moments" else " + k_odom_update_wide" if dom == "odom_features" else ""),
            "stage": dom,
            "bound": "hbm",
            "achieved": round(achieved, 3),
            "peak": 8000.0,
            "unit": "GB/s",
            "frac": round(achieved / 8000.0, 6),
            "traffic": None,
            "avg_launch_us": round(avg_ms * 1e3, 3),
            "launches": kt[dom]["launches"],
            "launches_note": "every launch of this stage timed with HIP events on the estimator's stream in a separate, untimed block of steps (the events cost ~10 % of a step when they sit in the timed region)",
            "algorithmic_bytes_per_launch": round(alg_bytes, 1),
            "algorithmic_note": "SURVEY.md 8(d) per-unit figure x units of one launch: 60 B per lidar residual slot (moments) / 16(M+N) + 8*5*M + 32*M (+ 33 B per row slot) per search call",
        }
        if dom == "moments":
            # what the kernel pair really moves: float4 point + float4 coefficient + u8 flag per slot = 33 B, the per-block
            # partials written once by the moments kernel and read once by the fold (blocks x 260 doubles), the folded moments
            slots = alg_bytes / 60.0
            bpf = max(1, min(64, int(np.ceil(max(slots / Wo, 1) / 512.0))))
            actual = 33.0 * slots + 2.0 * Wo * bpf * 260 * 8 + Wo * 260 * 8
            roofline["actual_bytes_per_launch"] = round(actual, 1)
            roofline["frac_actual"] = round(actual / (avg_ms * 1e-3) / 8e12, 6)
            roofline["actual_note"] = "33 B read per slot + per-block partials (written, then read by the fold); the 60 B figure counts p and w as fp64 triples, the kernel reads them as fp32"
        pmc, pmc_note = ({}, "skipped (--no-pmc or N > 1)")
        if not args.no_pmc and world == 1:
            pmc, pmc_note = measure_pmc(ds, clouds, args.workload)
            traffic, tnote = pmc_traffic(pmc, kernel_of[dom])
            roofline["traffic"] = traffic
            roofline["traffic_source"] = pmc_note + "; " + tnote
        roofline["others"] = {
            n: {"avg_launch_us": round(per_launch(n)[0] * 1e3, 3), "achieved_GBps": round(per_launch(n)[1], 2)}
            for n in ("features", "odom_features", "moments", "voxel", "knn_grid") if n != dom
        }
        # the search kernels are bound by vector-instruction issue, not by HBM: the yardstick that fits them
        for stage, kname in (("features", "k_features"), ("odom_features", "k_odom_round")):
            vi = pmc_valu_issue(pmc, kname)
            tgt = roofline if stage == dom else roofline["others"].get(stage)
            if vi and tgt is not None:
                tgt["valu_issue"] = vi
        if resident["launches"] > 0:
            # In the timed region the lidar moments do NOT come from the k_lidar_moments + k_moment_reduce launches timed above
            # (the same arithmetic as separate launches, used when events bracket every kernel) but from ONE resident kernel per
            # solve that serves every linearisation behind a doorbell.  A pass of it is timed on the device's wall
            # clock (doorbell seen -> sums posted); between passes the kernel idles while the host factors the system.
            p_us = 1e3 * resident["total_ms"] / resident["launches"]
            b = resident["algorithmic_bytes"] / resident["launches"]
            n_solves = max(1, n_blocks * args.steps + args.warmup)
            ppl = resident["launches"] / n_solves
            l_us = 1e3 * resident_launch["total_ms"] / max(resident_launch["launches"], 1)
            rt, rnote = pmc_traffic(pmc, "k_lidar_moments_resident")
            # `achieved` / `frac` are over the WHOLE launch (dispatch -> exit, HIP events = what rocprofv3 --stats reports as this kernel's
            # duration): algorithmic bytes of all passes of a launch / the launch duration.  The per-pass figure (device clock inside
            # the kernel, excludes the time the kernel idles between passes while the host factors) is carried as `frac_per_pass`.
            whole_gbps = b * ppl / (l_us * 1e-6) / 1e9 if l_us > 0 else None
            flop_launch = b / 60.0 * 684.0 * ppl          # 684 MFMA-flop per residual
            res = {
                "kernel": "k_lidar_moments_resident (fp64-MFMA form; 1 launch per solve, 1 pass per linearisation)",
                "stage": "moments", "bound": "hbm",
                "achieved": round(whole_gbps, 3) if whole_gbps else None, "peak": 8000.0, "unit": "GB/s",
                "frac": round(whole_gbps / 8000.0, 6) if whole_gbps else None,
                "frac_basis": "algorithmic bytes of one LAUNCH (60 B x residual slots x passes per launch) / average launch duration (HIP events, dispatch -> exit); reproducible from profiles/*_kernel_stats.md",
                "traffic": round(rt, 1) if rt else None,
                "traffic_source": pmc_note + "; " + rnote + f"; per LAUNCH (the features are read once per launch and stay in registers; {ppl:.2f} passes per launch)",
                "mfma_f64": {"achieved_TFLOPs": round(flop_launch / (l_us * 1e-6) / 1e12, 3) if l_us > 0 else None, "peak_TFLOPs": 78.6,
                             "frac": round(flop_launch / (l_us * 1e-6) / 78.6e12, 6) if l_us > 0 else None,
                             "frac_per_pass": round(b / 60.0 * 684.0 / (p_us * 1e-6) / 78.6e12, 6) if p_us > 0 else None,
                             "note": "684 flop per residual (SURVEY.md 8d) over the same launch duration / pass time"},
                "unit_of_work": "one LAUNCH = one solve's dogleg loop; one PASS = one linearisation of the window's lidar factors (what one k_lidar_moments + k_moment_reduce launch pair did in round 2)",
                "avg_launch_us": round(l_us, 2), "launches_timed": resident_launch["launches"], "passes_per_launch": round(ppl, 2),
                "algorithmic_bytes_per_launch": round(b * ppl, 1),
                "launch_timing": "HIP events around the kernel's launches in a separate untimed block: dispatch -> exit = the whole dogleg loop of a solve incl. the host's factorisations between passes (what rocprofv3 --stats reports as this kernel's duration)",
                "passes": resident["launches"], "avg_pass_us": round(p_us, 3), "algorithmic_bytes_per_pass": round(b, 1),
                "achieved_per_pass_GBps": round(b / (p_us * 1e-6) / 1e9, 3) if p_us > 0 else None,
                "frac_per_pass": round(b / (p_us * 1e-6) / 8e12, 6) if p_us > 0 else None,
                "pass_timing": "device wall clock inside the kernel, doorbell seen -> sums posted, slowest frame (HIP events cannot bracket a pass of a resident kernel)",
            }
            if dom == "moments":   # the resident kernel IS the dominant kernel of the timed region: it leads, the launch form follows as a cross-check
                launch_form = {k: v for k, v in roofline.items() if k != "others"}
                others = roofline["others"]
                roofline = res
                roofline["launch_form_of_the_same_pass"] = launch_form
                roofline["others"] = others
            else:
                roofline["moments_resident"] = res

        # the dominant kernel given B windows of work in one launch
        batched_kernel = None
        if not args.shard_factors and world == 1:
            batched_kernel = []
            for B in (1, 8, 64, 512):
                ms_b, bytes_b = est.bench_batched_moments(B, 20)
                gbps = bytes_b / (ms_b * 1e-3) / 1e9
                batched_kernel.append({"windows": B, "avg_launch_us": round(ms_b * 1e3, 2), "algorithmic_MB": round(bytes_b / 1e6, 2),
                                       "achieved_GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / 8000.0, 4),
                                       "actual_GBps_at_33B_per_slot": round(gbps * 33.0 / 60.0, 1), "frac_actual": round(gbps * 33.0 / 60.0 / 8000.0, 4),
                                       "mfma_f64_GFLOPs": round(bytes_b / 60.0 * 684.0 / (ms_b * 1e-3) / 1e9, 1)})
        batched = None
        if args.windows and not args.shard_factors and world == 1:
            sizes = sorted({int(v) for v in str(args.windows).split(",") if int(v) > 0})
            try:
                batched = batched_windows(hip, ds, kind, W, Wo, est, sizes, rep)
            except Exception as e:  # noqa: BLE001 -- an extra must not take the bench line down
                batched = {"error": f"{type(e).__name__}: {e}"}

        # the rest of the estimator step a sweep triggers (Estimator::ProcessLaserOdom = push + solve + slide, Estimator.cc:430-774):
        # IMU samples of the interval, PushFrame (upload + VoxelGrid of the new surf stack + window push), SlideWindow
        k_last = len(ds.frames) - 1
        f_last = ds.frames[k_last]
        T_id = capi.TransformF.make([0, 0, 0, 1], [0, 0, 0])
        imu_ms, push_ms, slide_ms = [], [], []
        for _ in range(5):
            est.restore()
            est.solve()
            t = time.perf_counter()
            est.slide()
            slide_ms.append((time.perf_counter() - t) * 1e3)
            t = time.perf_counter()
            est.process_imu_batch(f_last.imu_dt, f_last.imu_acc, f_last.imu_gyr, f_last.imu_t + 1.0)
            imu_ms.append((time.perf_counter() - t) * 1e3)
            t = time.perf_counter()
            est.push_frame(T_id, clouds[k_last][0], clouds[k_last][1], f_last.t + 1.0)
            push_ms.append((time.perf_counter() - t) * 1e3)
        est.restore()
        step_extra_ms = float(np.median(imu_ms) + np.median(push_ms) + np.median(slide_ms))
        odom_ms, packer_ms = odometry_ms_per_scan(hip, ds) if kind == "outdoor" else (None, None)
        captured = []
        map_stats = mapping_ms_per_scan(hip, ds, clouds, capture=captured)
        kf_stats = None
        if args.keyframes > 0 and world == 1 and captured:
            kf_stats = keyframe_batch_stats(hip, captured, args.keyframes)
            if not args.no_cpu_baseline:
                orc_kf = keyframe_batch_stats(_oracle_lib(), captured, 4, reps=1, distinct_maps=False)
                kf_stats["cpu_oracle_keyframes_per_s"] = orc_kf["keyframes_per_s"]
        fed = None
        if world == 1 and not args.no_fed and kind == "outdoor":
            try:
                fed = fed_gpu_points(hip, ds, est, captured, W, Wo)
            except Exception as e:  # noqa: BLE001 -- an extra must not take the bench line down
                fed = {"error": f"{type(e).__name__}: {e}"}
        if fed and "search_and_fit" in fed and pmc:
            vi = pmc_valu_issue(pmc, "k_kf_round*")
            tr, tn = pmc_traffic(pmc, "k_kf_round*")
            c0 = captured[0]
            M, N = c0["corner"].shape[0] + c0["surf"].shape[0], c0["corner_map"].shape[0] + c0["surf_map"].shape[0]
            fed["search_and_fit"]["pmc_at_64_keyframes"] = {"valu_issue": vi, "hbm_bytes_per_round": tr, "algorithmic_bytes_per_round": 64 * (16 * (M + N) + 72 * M),
                                                            "traffic_ratio": round(tr / (64.0 * (16 * (M + N) + 72 * M)), 2) if tr else None, "source": tn}
        cpu = None if (args.no_cpu_baseline or world > 1) else cpu_baseline(kind, W, Wo, args.cpu_steps, ds)  # N = 1 only
        odom_io = 3 if kind == "outdoor" else 2
        pp_med = float(np.median(pp_ms[1:]))
        out = {
            "metric": "sliding-window solves/sec, 64-line 130k-pt scans, window=15 (opt_window=5)" if kind == "outdoor"
            else "sliding-window solves/sec, VLP-16 28.8k-pt scans, window=15 (opt_window=5) [NOT the headline workload]",
            "value": round(value, 3),
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt_max / args.steps, 4),
            "timing": {"blocks": n_blocks, "steps_per_block": args.steps, "timed_seconds": round(float(np.sum(block_s)), 4),
                       "ms_per_step_min": round(1e3 * float(np.min(block_s)) / args.steps, 4), "ms_per_step_max": round(1e3 * float(np.max(block_s)) / args.steps, 4),
                       "note": "value / ms_per_step = the MEDIAN block of exactly --steps steps (barrier + synchronize on both sides, max over ranks); a step = restore + SolveOptimization, the --steps of a block looped inside the library (lio_est_solve_restored: no Python between two solves)"},
            "higher_is_better": True,
            "scaling": "strong" if args.shard_factors else "weak",
            "vs_baseline": None,
            "dtype": "f64 (solve) / f32 (features)",
            "data": "synthetic",
            "config": {
                "workload": "HDL-64E outdoor_test_config_64, S_outdoor ray-cast scans, window_size=15 opt_window_size=5, one SolveOptimization per step, clouds resident in HBM"
                if kind == "outdoor" else "VLP-16 indoor, window_size=15 opt_window_size=5",
                "points_per_scan": int(ds.frames[0].scan.shape[0]),
                "n_lidar_residuals": int(rep.n_lidar_residuals),
                "local_map_points": int(rep.n_local_map),
                "surf_stack_points": int(new_stack_n),
                "solver_iterations": int(rep.iterations),
                "laser_odom_iterations": int(rep.laser_odom_iterations),
                "parallelism": (f"1 window, factors sharded over {world} ranks + all-reduce" if args.shard_factors else f"{world} independent windows") if world > 1 else "1 window",
            },
            "roofline": roofline,
            "cpu_baseline": cpu,
            "sharded": None,
            "keyframes": None,
            "batched": batched,
            "fed_gpu": fed,
            "keyframe_batch": kf_stats,
            "batched_kernel_roofline": {"kernel": "k_lidar_moments_batched + k_moment_reduce (fp64-MFMA form at every size; the structured fp64-VALU form is LIO_MOMENTS=valu)", "note": "B copies of this window's lidar factors at distinct addresses, one launch, HIP events over 20 launches; 60 B and 684 MFMA-flop per residual (SURVEY.md §8d)", "points": batched_kernel},
            "stages_ms": {
                "t_build_map": round(rep.ms_build_map, 4),
                "feature_cost": round(rep.ms_features, 4),
                "prepare_for_ceres": round(rep.ms_prepare, 4),
                "t_opt": round(rep.ms_opt, 4),
                "whole_marginalization": round(rep.ms_marg, 4),
                "tic_toc_opt": round(rep.ms_total, 4),
            },
            "kernels": {n: {"launches": kt[n]["launches"], "total_ms": round(kt[n]["total_ms"], 4)} for n in names},
            "ms_per_scan": {
                "point_processor_incl_h2d_d2h": round(pp_med, 4),
                "point_odometry_incl_h2d": odom_ms,
                "point_odometry_packer_mode": packer_ms,
                "point_mapping_incl_h2d": map_stats,
                "estimator_process_imu_per_interval": round(float(np.median(imu_ms)), 4),
                "estimator_push_frame_incl_h2d": round(float(np.median(push_ms)), 4),
                "estimator_slide_window": round(float(np.median(slide_ms)), 4),
                "estimator_step_amortised_over_odom_io": round((1e3 * dt_max / args.steps + step_extra_ms) / odom_io, 4),
                "total_before_imu_init": round(pp_med + (odom_ms or 0.0) + map_stats["ms_per_scan"], 4),
                "total": round(pp_med + (packer_ms or 0.0) + (1e3 * dt_max / args.steps + step_extra_ms) / odom_io, 4),
                "note": "total = per 10 Hz sweep after IMU init: PointProcessor + PointOdometry in packer mode (the estimator disables it after IMU init, SURVEY.md A.18) + 1/odom_io of an estimator step (ProcessImu of the interval's samples + PushFrame + SolveOptimization + SlideWindow).  total_before_imu_init = PointProcessor + scan-to-scan odometry + scan-to-map (no solves yet)",
            },
            "setup_s": round(setup_s, 2),
        }
    # N > 1: the two modes with a real exchange step, measured in the same run on every rank (collective), reported by rank 0 in
    # the same line.  They come LAST and under a watchdog: if a rank fails or a collective never completes, rank 0 still prints the
    # line it has (the weak-scaling measurement) with the failure noted, instead of hanging the whole run.
    if rank == 0 and thr is None and world == 1 and isinstance(out.get("batched"), dict) and out["batched"].get("points"):
        # N = 1: the same throughput mode as the N > 1 line's value, from the `batched` extra at --rank-windows windows — the figure an
        # N-GPU `value` is to be compared with (the headline here stays the single window: SURVEY.md 8(d)(i))
        pt = [p for p in out["batched"]["points"] if p.get("windows") == args.rank_windows] or [out["batched"]["points"][-1]]
        out["throughput_mode"] = {"windows_per_rank": pt[0].get("windows"), "ranks": 1, "value": pt[0].get("value"), "unit": "solves/s", "ms_per_step": pt[0].get("ms_per_batch_step"),
                                  "parity": pt[0].get("parity"), "note": "lio_est_batch, the mode `bench.py --gpus N` (N > 1) reports as its `value`: compare an N-GPU value with N x this; details in batched.points"}
        out["single_window"] = {"value": out["value"], "unit": "solves/s", "ms_per_step": out["ms_per_step"], "note": "the line's value at N = 1: one latency-bound window"}
    if rank == 0 and thr is not None:
        # N > 1: the throughput mode leads the line; the N one-window-per-rank figure measured above stays beside it
        out["single_window"] = {"value": out["value"], "unit": "solves/s", "ms_per_step": out["ms_per_step"], "timing": out["timing"],
                                "note": f"{world} ranks x ONE window each (latency-bound: SURVEY.md 8(d)(i)); its roofline is `roofline` below"}
        out["throughput_mode"] = thr
        if thr["value"] is not None:
            out["value"] = thr["value"]
            out["ms_per_step"] = thr["ms_per_step"]
            out["config"]["parallelism"] = f"{world} ranks x {thr['windows_per_rank']} independent windows per step (lio_est_batch), no collective on the data path"
            out["config"]["workload"] += f"; throughput mode: {thr['windows_per_rank']} windows per rank and step"
            out["timing"] = {"blocks": 1, "steps_per_block": args.steps, "note": "value = ranks x windows_per_rank x --steps / the time of exactly --steps batch steps (barrier + synchronize on both sides, max over ranks); "
                             "compare with the N = 1 line's batched.points[windows = windows_per_rank].value, not with its single-window value"}
        else:
            out["throughput_mode_note"] = "identical windows disagreed on a rank: the line falls back to the single-window figure"
    if world > 1 and not args.shard_factors:
        done = threading.Event()

        def watchdog():
            if done.wait(timeout=240.0):
                return
            if rank == 0:
                out["sharded"] = out["keyframes"] = {"error": "the multi-GPU extras did not finish within 240 s (a rank failed or a collective hung); the headline line above them is unaffected"}
                print(json.dumps(out), flush=True)
            os._exit(0 if rank == 0 else 3)

        threading.Thread(target=watchdog, daemon=True).start()
        extras = {}
        try:
            extras["sharded"] = sharded_solve_stats(hip, kind, W, Wo, rank, world, max(10, args.steps // 2), sync=torch.cuda.synchronize, device="cuda")
            if args.keyframes > 0:
                km = keyframes_measure(hip, rank, world, min(args.keyframes, 256), 3, sync=torch.cuda.synchronize, device="cuda")
                extras["keyframes"] = km["line"] if km else None
        except Exception as e:  # noqa: BLE001 -- this rank failed: say so; the peers' watchdogs end them
            extras = {"sharded": {"error": f"rank {rank}: {type(e).__name__}: {e}"}, "keyframes": None}
            if rank != 0:
                done.set()
                os._exit(3)
        done.set()
        if rank == 0:
            out.update(extras)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        try:
            dist.barrier()   # leave the group together
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass


def keyframes_workload(args, hip, rank, world, torch, dist):
    """configs[4] as a bench line: a step = one refinement of ALL --keyframes keyframes (each with its own local map in HBM).
    Rank r owns keyframes r, r+N, ...; the one exchange is an all-gather of the refined poses (RCCL).  Strong scaling."""
    m = keyframes_measure(hip, rank, world, args.keyframes, args.steps, sync=torch.cuda.synchronize, device="cuda", warmup=args.warmup)
    if rank != 0:
        return
    n_kf, dt_max, alg_bytes, ms, captured = m["n_kf"], m["dt_max"], m["alg_bytes"], m["device_ms"], m["captured"]
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        o = keyframe_batch_stats(_oracle_lib(), captured, 4, reps=1, distinct_maps=False)
        cpu = {"value": o["keyframes_per_s"], "unit": "keyframes/s", "cores": 1, "kind": "port",
               "sample": "4 keyframes of the same workload refined one after the other by the CPU oracle (oracle/mapping.h RefineKeyframe)"}
    print(json.dumps({
        "metric": "keyframe refinements/sec, 64-line local maps (map_builder batched refinement, BASELINE.json configs[4])",
        "value": round(n_kf * args.steps / dt_max, 1), "unit": "keyframes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt_max / args.steps, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{n_kf} HDL-64E keyframes (S_outdoor), each against its own local map resident in HBM, one scan-to-map Gauss-Newton loop per keyframe",
                   "stack_points_per_keyframe": int(captured[0]["corner"].shape[0] + captured[0]["surf"].shape[0]),
                   "local_map_points_per_keyframe": int(captured[0]["corner_map"].shape[0] + captured[0]["surf_map"].shape[0]),
                   "iterations_mean": m["iterations_mean"], "rccl_world": m["line"]["rccl_world"],
                   "parallelism": m["line"]["parallelism"]},
        "roofline": {"kernel": "k_kf_round (+ k_kf_rows, k_kf_update)", "bound": "hbm", "achieved": round(alg_bytes / (ms * 1e-3) / 1e9, 1), "peak": 8000.0,
                     "unit": "GB/s", "frac": round(alg_bytes / (ms * 1e-3) / 8e12, 4), "traffic": None,
                     "note": "rank 0's share; algorithmic bytes = sum over its keyframes of iterations x (16(M+N) + 72 M) (SURVEY.md 8(d)); device time of the round loop by HIP events; the kernel is gather-latency-bound (DESIGN.md 3.4)"},
        "cpu_baseline": cpu,
    }))


def fed_gpu_points(hip, ds, est, captured, W, Wo):
    """SURVEY.md 8(d) ii for the stages besides the moments kernel: the same kernels given B units of work at once, B in {1, 8, 64(, 512)},
    against the bound that limits them.  Search + plane fit: B keyframes (each with its own local map in HBM) through k_kf_round,
    the batched form of the k_features / k_odom_round device body.  PointProcessor: B sweeps in flight (one handle + stream each,
    lio_pp_process_async).  VoxelGrid: one filter over B tiled copies of the window's surf clouds (B x 150 k points)."""
    from lio_amd import capi

    out = {}
    # (i) search + fit
    pts = []
    for B in (1, 8, 64, 512):
        if not captured:
            break
        k = keyframe_batch_stats(hip, captured, B, reps=3)
        pts.append({"keyframes": B, "device_ms": k["refine_device_ms"], "iterations_mean": k["iterations_mean"], "achieved_GBps": k["achieved_GBps"],
                    "frac_of_8TBps": k["frac_of_8TBps"], "keyframes_per_s": k["keyframes_per_s"]})
    out["search_and_fit"] = {"kernel": "k_kf_round (+ k_kf_rows, k_kf_update): B scan-to-map Gauss-Newton loops in lock-step, the batched form of the search + plane-fit body",
                             "bound": "vector-instruction issue (DESIGN.md 3.9); HBM fraction shown for reference", "points": pts}
    # (ii) PointProcessor, B sweeps per call: lio_pp_process_batch over handles of one sensor = ONE launch chain (every kernel once over all
    # sweeps, the sweep in blockIdx.z).  Headline of the stage: the sweeps already resident in HBM (lio_pp_process_batch_device — the
    # contract's "inputs resident in HBM when the timed region starts"); beside it the same call fed from host memory (the 2.1 MB of
    # every sweep over PCIe inside the clock) and round 5's form (one handle + stream per sweep in flight, lio_pp_process_async).
    import torch

    lid = ds.lidar
    scans = [f.scan for f in ds.frames[:4]]
    dev_scans = [torch.from_numpy(np.ascontiguousarray(sc, np.float32)).cuda() for sc in scans]
    torch.cuda.synchronize()
    npts = float(np.mean([sc.shape[0] for sc in scans]))
    pts = []
    for B in (1, 8, 64, 256):
        hs = [capi.PointProcessor(hip, lid.lower_deg, lid.upper_deg, lid.rings) for _ in range(B)]
        ins = [scans[i % len(scans)] for i in range(B)]
        dptr = [dev_scans[i % len(scans)].data_ptr() for i in range(B)]
        dn = [dev_scans[i % len(scans)].shape[0] for i in range(B)]
        reps = max(3, 128 // B)
        capi.PointProcessor.process_batch_device(hs, dptr, dn)     # warm-up: buffers
        t = time.perf_counter()
        for _ in range(reps):
            capi.PointProcessor.process_batch_device(hs, dptr, dn)
        dt_dev = time.perf_counter() - t
        n_out = sum(int(h.lib.dll.lio_pp_count(h.h, 4)) for h in hs[:4])
        row = {"sweeps_per_call": B, "sweeps_per_s": round(B * reps / dt_dev, 1), "ms_per_sweep": round(1e3 * dt_dev / (B * reps), 4),
               "achieved_GBps": round(B * reps / dt_dev * npts * 40.0 / 1e9, 2), "frac_of_8TBps": round(B * reps / dt_dev * npts * 40.0 / 8e12, 5),
               "less_flat_points_of_the_first_sweeps": n_out}
        if B <= 64:
            capi.PointProcessor.process_batch(hs, ins)
            t = time.perf_counter()
            for _ in range(reps):
                capi.PointProcessor.process_batch(hs, ins)
            dt_host = time.perf_counter() - t
            row["from_host_memory_sweeps_per_s"] = round(B * reps / dt_host, 1)
            for i, h in enumerate(hs):
                h.process(ins[i])
            t = time.perf_counter()
            for _ in range(reps):
                for i, h in enumerate(hs):
                    h.process_async(ins[i])
                for h in hs:
                    h.wait()
            dt_async = time.perf_counter() - t
            row["one_handle_and_stream_per_sweep_from_host_memory_sweeps_per_s"] = round(B * reps / dt_async, 1)
        pts.append(row)
        del hs
    out["point_processor"] = {"bound": "hbm, 40 B per input point (SURVEY.md 8(d)); sweeps resident in HBM, wall clock of lio_pp_process_batch_device (device-to-device copy into the chain's "
                                       "segments, the chain, the copy of all counts back, the host's wait)", "points": pts,
                              "note": "sweeps_per_call handles of one sensor = one launch chain over all sweeps (B = 1: the handle's own chain); from_host_memory_* = lio_pp_process_batch with the "
                                      "2.1 MB upload of every sweep over PCIe inside the clock; one_handle_and_stream_per_sweep_* = round 5's form of the batch (host-enqueue-bound)"}
    # (iii) VoxelGrid over B tiled copies
    cloud = np.concatenate([est.get_surf_stack(i) for i in range(W - Wo, W)], axis=0)
    cloud = cloud[(np.abs(cloud[:, 0]) < 60.0) & (np.abs(cloud[:, 1]) < 60.0) & (np.abs(cloud[:, 2]) < 20.0)]   # the 120 m x 120 m core: tiles 132 m apart stay inside the key
    pitch = 132.0
    order = [0, -1, 1, -2, 2]
    pts = []
    for B in (1, 8, 64):
        tiles = []
        for b in range(B):
            ix, iy, iz = order[b % 5], order[(b // 5) % 5], order[(b // 25) % 5]
            c = cloud.copy()
            c[:, 0] += ix * pitch; c[:, 1] += iy * pitch; c[:, 2] += iz * 45.0
            tiles.append(c)
        big = np.concatenate(tiles, axis=0)
        ms, n_out = hip.bench_voxel_grid(big, 0.4, reps=5 if B < 64 else 2)
        pts.append({"copies": len(tiles), "points_in": int(big.shape[0]), "points_out": n_out, "device_ms": round(ms, 4),
                    "achieved_GBps": round(32.0 * big.shape[0] / (ms * 1e-3) / 1e9, 2), "frac_of_8TBps": round(32.0 * big.shape[0] / (ms * 1e-3) / 8e12, 5)})
    out["voxel_grid"] = {"bound": "hbm, 32 B per point (read 16 + written <= 16)", "points": pts,
                         "note": "one filter call over tiled copies of the window's surf clouds (the copies sit 2.2 extents apart, inside the 10 + 11 + 11-bit absolute-cell key)"}
    return out


def batched_windows(hip, ds, kind, W, Wo, est0, sizes, rep0, min_seconds=0.4):
    """B windows per launch chain through lio_est_batch (every stage ONE launch over all windows, trust-region
    loop and marginalization on the device).  The windows are copies of the headline window at distinct addresses
    (lio_est_copy_snapshot); a step = restore every window + lio_est_batch_solve, looped inside the library; the last step's
    marginalizations are inside the timed region (lio_est_batch_solve_restored ends with a wait for the batch's stream).
    Per stage: device time (HIP events on the batch's stream, last step), SURVEY.md 8(d) algorithmic bytes of B windows, fraction of
    the 8 TB/s HBM roofline; the trust-region loop also against the 78.6 TFLOP/s fp64 matrix pipe (684 flop per residual and pass)."""
    from lio_amd import capi

    pivot = W - Wo
    n_stack = [est0.get_surf_stack(i).shape[0] for i in range(W + 1)]
    n_local = sum(n_stack[pivot:W])
    m_static = sum(n_stack[pivot + 1:W])
    m_new = n_stack[W]
    n_map = int(rep0.n_local_map)
    n_res = int(rep0.n_lidar_residuals)
    n_slots = m_static + m_new
    cfg = est_config(hip, ds, kind, W, Wo)
    clones = []
    points = []
    for B in sizes:
        while len(clones) < B:
            e = capi.Estimator(hip, cfg)
            e.copy_snapshot_of(est0)
            e.restore()
            clones.append(e)
        batch = capi.EstimatorBatch(hip, clones[:B])
        batch.solve_restored(2)            # warm-up: buffers, the sort's and the scan's scratch, the priors' upload
        steps, dt = 2, 0.0
        while True:
            t0 = time.perf_counter()
            reps = batch.solve_restored(steps)
            dt = time.perf_counter() - t0
            if dt >= min_seconds or steps >= 256:
                break
            steps = min(256, max(steps * 2, int(steps * 1.3 * min_seconds / max(dt, 1e-6)) + 1))
        passes = reps[0].iterations + 1
        # parity gate of the figure: the B windows are identical inputs, so every stage of every window must have left identical bits
        # on the device (lio_est_batch_stage_digest) and identical reports; rep0 is the same window through the single-window handle
        it_same = all((r.iterations, r.successful_steps, r.termination, r.n_lidar_residuals, r.final_cost) ==
                      (reps[0].iterations, reps[0].successful_steps, reps[0].termination, reps[0].n_lidar_residuals, reps[0].final_cost) for r in reps)
        differing = []
        for s_idx, s_name in enumerate(capi.EstimatorBatch.STAGES):
            d = batch.stage_digest(s_idx)
            if bool((d != d[0]).any()):
                differing.append(s_name)
        # A batch of >= 96 windows is solved as two halves side by side (lio_est_batch_set_option "parts"): the halves' stages overlap on the
        # GPU, so their event-bracketed times cannot be attributed.  The VALUE and the digests above are the default's (two parts); the
        # per-stage device times below come from three more steps of the same batch solved as ONE part (stages one after the other).
        two_parts = B >= 96
        if two_parts:
            batch.set_option("parts", 1)
            batch.solve_restored(3)
        clk = batch.clock()
        same_as_single = (reps[0].iterations == rep0.iterations and reps[0].n_lidar_residuals == rep0.n_lidar_residuals)
        if not it_same or differing:
            points.append({"windows": B, "parity": "broken", "value": None, "all_windows_same_decisions": bool(it_same), "stages_that_differ_between_windows": differing,
                           "same_decisions_as_the_single_window_handle": bool(same_as_single),
                           "note": "identical windows disagreed: no throughput is reported for this size (tests/test_gpu_batch_scale.py pins this)"})
            batch.close()
            continue

        def stage(ms, nbytes, flops=None):
            d = {"device_ms": round(ms, 4), "algorithmic_MB": round(nbytes * B / 1e6, 2)}
            if ms > 0:
                gbps = nbytes * B / (ms * 1e-3) / 1e9
                d["achieved_GBps"] = round(gbps, 1)
                d["frac_of_8TBps"] = round(gbps / 8000.0, 4)
                if flops is not None:
                    d["mfma_f64_TFLOPs"] = round(flops * B / (ms * 1e-3) / 1e12, 3)
                    d["frac_of_78p6_TFLOPs"] = round(flops * B / (ms * 1e-3) / 78.6e12, 4)
            return d

        rounds = int(clk["rounds"])
        points.append({
            "windows": B, "steps": steps, "value": round(B * steps / dt, 1), "unit": "solves/s", "ms_per_batch_step": round(1e3 * dt / steps, 3),
            "parts": 2 if two_parts else 1,
            "windows_on_device_loop": int(clk["n_device"]), "parity": "ok", "all_windows_same_decisions": True, "all_stage_digests_equal": True, "same_decisions_as_the_single_window_handle": bool(same_as_single),
            "solver_iterations": int(reps[0].iterations), "n_lidar_residuals": int(reps[0].n_lidar_residuals), "newest_frame_rounds": rounds,
            "host_ms": {k: round(clk[k], 3) for k in ("describe", "filter", "grid_features_rounds", "pack", "solve", "finish", "total")},
            "stages": {
                "filter (concat + keys, segmented sort, heads, centroids)": stage(clk["dev_filter"], 32.0 * n_local),
                "knn_grid (cell keys, segmented sort, run-start table + cell-sorted points)": stage(clk["dev_grid"], 32.0 * n_map),
                "features (older frames)": stage(clk["dev_features"], 16.0 * (m_static + n_map) + 72.0 * m_static),
                "newest_frame_rounds": stage(clk["dev_rounds"], rounds * (16.0 * (m_new + n_map) + 72.0 * m_new + 33.0 * m_new)),
                "trust_region_loop (moments + aux row, step)": stage(clk["dev_loop"], 60.0 * n_slots * passes, 684.0 * n_res * passes),
                "marginalization (aux row, Schur + eigensolves)": {"device_ms": round(clk["dev_marg"], 4)},
                "wait_for_the_previous_marginalization (own stream; joined before the problems' upload, not part of the rounds above)": {"device_ms": round(clk["dev_marg_wait"], 4)},
            },
        })
        batch.close()
    return {"note": "lio_est_batch: B copies of the headline window at distinct addresses, one launch per stage over all windows; value = B x steps / wall time "
                    "of lio_est_batch_solve_restored (restore + solve per step, the last marginalizations inside; from 96 windows the batch solves its two halves "
                    "side by side from two host threads: `parts`); stage device times and host_ms = HIP events / host clock of a step of the same batch solved as ONE part",
            "per_window": {"local_map_points_before_filter": int(n_local), "local_map_points": n_map, "older_frames_queries": int(m_static), "newest_frame_queries": int(m_new),
                           "residual_slots": int(n_slots)},
            "points": points}


def odometry_ms_per_scan(hip, ds):
    """Scan-to-scan odometry step on consecutive sweeps of the same scene (its role before IMU initialisation)."""
    from lio_amd import capi, synth

    sweeps, _, lid = synth.make_sweeps("outdoor", 4)
    od = capi.PointOdometry(hip, 0.1, 3, 25, False)
    pp = capi.PointProcessor(hip, lid.lower_deg, lid.upper_deg, lid.rings)
    ms = []
    for sw in sweeps:
        pp.process(sw)
        cl = [pp.cloud(w) for w in (1, 2, 3, 4)]
        t = time.perf_counter()
        od.process(*cl)
        ms.append((time.perf_counter() - t) * 1e3)
    od.enable(False)   # packer mode after IMU initialisation (A.18)
    ms_packer = []
    for sw in sweeps[:3]:
        pp.process(sw)
        cl = [pp.cloud(w) for w in (1, 2, 3, 4)]
        t = time.perf_counter()
        od.process(*cl)
        ms_packer.append((time.perf_counter() - t) * 1e3)
    return round(float(np.median(ms[1:])), 4), round(float(np.median(ms_packer)), 4)


def mapping_ms_per_scan(lib, ds, clouds, n_frames=8, capture=None):
    """Scan-to-map step (PointMapping::Process: from-map extraction, stack VoxelGrid, <=10 Gauss-Newton rounds against the
    cube map, map update) on consecutive sweeps; transform_sum = ground truth + a growing drift.  Median wall ms per sweep
    and the map sizes of the last one."""
    from lio_amd import capi, synth

    mp = capi.PointMapping(lib)
    f0 = ds.frames[0]
    R0 = f0.R_wb @ ds.R_lb.T
    p0 = f0.p_wb - R0 @ ds.t_lb
    ms = []
    r = None
    for k, f in enumerate(ds.frames[:n_frames]):
        R = f.R_wb @ ds.R_lb.T
        p = f.p_wb - R @ ds.t_lb
        q = synth.quat_from_rot(R0.T @ R @ synth.small_rot(np.array([0.002, -0.001, 0.003]) * k))
        T = (q, R0.T @ (p - p0) + np.array([0.05, -0.03, 0.02]) * k)
        surf, corner = clouds[k]
        t = time.perf_counter()
        r = mp.process(corner, surf, T)
        ms.append((time.perf_counter() - t) * 1e3)
        if capture is not None and k >= 2:   # local map + down-sampled stacks + settled pose of this sweep = one keyframe
            capture.append(dict(corner_map=mp.cloud(2), surf_map=mp.cloud(3), corner=mp.cloud(0), surf=mp.cloud(1), T=r["T_aft"]))
    return {
        "ms_per_scan": round(float(np.median(ms[2:])), 4),
        "from_map_points": int(mp.cloud(2).shape[0] + mp.cloud(3).shape[0]),
        "stack_points": int(mp.cloud(0).shape[0] + mp.cloud(1).shape[0]),
        "iterations_last": int(r["iterations"]),
        "rows_last": int(r["num_selected"]),
    }


def keyframe_batch_stats(lib, captured, n_keyframes, reps=3, distinct_maps=True):
    """BASELINE.json configs[4]: n_keyframes 64-line keyframes, each with its own local map (a separate copy in HBM when
    distinct_maps), initial poses = the settled pose of the source sweep perturbed by up to 0.15 m / 0.01 rad; one
    OptimizeTransformTobeMapped loop per keyframe, every stage of a round one launch over the whole batch."""
    from lio_amd import capi, synth

    rng = np.random.default_rng(5)
    b = capi.KeyframeBatch(lib)
    t0 = time.perf_counter()
    n_src = len(captured)
    n_maps = n_keyframes if distinct_maps else n_src
    for m in range(n_maps):
        c = captured[m % n_src]
        b.add_map(c["corner_map"], c["surf_map"])
    alg_bytes_round = []
    for k in range(n_keyframes):
        c = captured[k % n_src]
        q, p = c["T"]
        R = synth.rot_from_quat(np.asarray(q, np.float64)) @ synth.small_rot(rng.uniform(-0.01, 0.01, 3))
        T0 = (synth.quat_from_rot(R), np.asarray(p, np.float64) + rng.uniform(-0.15, 0.15, 3))
        b.add_keyframe(k if distinct_maps else k % n_src, c["corner"], c["surf"], T0)
        M = c["corner"].shape[0] + c["surf"].shape[0]
        N = c["corner_map"].shape[0] + c["surf_map"].shape[0]
        alg_bytes_round.append(16 * (M + N) + 8 * 5 * M + 32 * M)   # kNN + fit, per call
    setup_s = time.perf_counter() - t0
    r = b.refine()   # warm-up: uploads the stacks
    wall, dev = [], []
    for _ in range(reps):
        t = time.perf_counter()
        r = b.refine()
        wall.append((time.perf_counter() - t) * 1e3)
        dev.append(r["device_ms"])
    wall_ms, dev_ms = float(np.median(wall)), float(np.median(dev))
    alg = float(np.dot(np.asarray(alg_bytes_round, np.float64), r["iterations"].astype(np.float64)))
    spread = float(np.max(np.linalg.norm(r["p"][::n_src][:, :2] - r["p"][::n_src][:, :2].mean(axis=0), axis=1))) if n_keyframes >= 2 * n_src else 0.0
    return {
        "keyframes": n_keyframes, "distinct_local_maps": n_maps,
        "stack_points_per_keyframe": int(captured[0]["corner"].shape[0] + captured[0]["surf"].shape[0]),
        "local_map_points_per_keyframe": int(captured[0]["corner_map"].shape[0] + captured[0]["surf_map"].shape[0]),
        "refine_wall_ms": round(wall_ms, 3), "refine_device_ms": round(dev_ms, 3),
        "keyframes_per_s": round(n_keyframes / (wall_ms * 1e-3), 1),
        "iterations_mean": round(float(r["iterations"].mean()), 2), "iterations_max": int(r["iterations"].max()),
        "algorithmic_GB": round(alg / 1e9, 3), "achieved_GBps": round(alg / (dev_ms * 1e-3) / 1e9, 1) if dev_ms > 0 else None,
        "frac_of_8TBps": round(alg / (dev_ms * 1e-3) / 8e12, 4) if dev_ms > 0 else None,
        "xy_spread_of_refined_copies_m": round(spread, 4),
        "setup_s": round(setup_s, 2),
        "note": "algorithmic bytes = sum over keyframes of iterations x (16(M+N) + 8*5*M + 32*M), SURVEY.md 8(d) kNN+fit row",
    }


def _oracle_lib():
    """The CPU oracle library: used ONLY for the reported cpu baselines (never on the measured path)."""
    import subprocess

    from lio_amd import capi

    so = os.path.join(ROOT, "oracle", "liblio_oracle.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    return capi.LioLib(so)


def measure_pmc(ds, clouds, workload, counters=("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU")):
    """Per-kernel PMC counters measured NOW on this box: the workload is pickled and a child `bench.py --pmc-child` replays a few
    solves on it under `rocprofv3 --pmc <counter> --kernel-trace`, ONE counter per pass (FETCH_SIZE and WRITE_SIZE do not fit one
    pass, MI355X_MICROARCH.md; SQ counters in a pass of their own).  Returns ({counter: {kernel: (grid, launches, avg value, avg
    duration ns)}} for the most frequent grid size of every kernel, note) — or ({}, reason)."""
    import pickle
    import shutil
    import sqlite3
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {}, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="lio_pmc_")
    out = {}
    try:
        pk = os.path.join(tmp, "workload.pkl")
        with open(pk, "wb") as fh:
            pickle.dump((ds, clouds), fh)
        for counter in counters:
            d = os.path.join(tmp, counter)
            env = dict(os.environ, TMPDIR=tmp)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__), "--pmc-child", pk,
                   "--steps", "3", "--workload", workload]
            r = subprocess.run(cmd, cwd=tmp, env=env, capture_output=True, text=True, timeout=300)
            dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
            if r.returncode != 0 or not dbs:
                return out, f"rocprofv3 --pmc {counter} failed (rc {r.returncode})"
            cur = sqlite3.connect(dbs[0]).cursor()
            q = ("select kernel_name, grid_size, count(*), avg(value), avg(duration) from counters_collection where counter_name=? "
                 "group by kernel_name, grid_size")
            best = {}
            for k, g, n, v, dur in cur.execute(q, (counter,)):
                short = k.split("(")[0].split("<")[0].split("::")[-1].strip()
                if short not in best or n > best[short][1]:
                    best[short] = (g, n, v, dur)
            out[counter] = best
        return out, "live: child runs of this bench under rocprofv3 --pmc <one counter> --kernel-trace (separate passes)"
    except Exception as e:  # noqa: BLE001 -- a profiler problem must not take the bench line down
        return out, f"pmc measurement failed: {type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _pmc_row(pmc, counter, kernel):
    """the counter's row of `kernel`; a trailing '*' matches the most-launched kernel whose name starts with the prefix"""
    rows = pmc.get(counter, {})
    if not kernel.endswith("*"):
        return rows.get(kernel)
    cand = [(v[1], v) for k, v in rows.items() if k.startswith(kernel[:-1])]
    return max(cand, key=lambda t: t[0])[1] if cand else None


def pmc_traffic(pmc, kernel):
    """HBM bytes per launch of `kernel`: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 — both counters are in KB and FETCH_SIZE reports half
    the bytes of a wide coalesced read on gfx950 (MI355X_MICROARCH.md).  (bytes or None, note)"""
    f, w = _pmc_row(pmc, "FETCH_SIZE", kernel), _pmc_row(pmc, "WRITE_SIZE", kernel)
    if not f or not w:
        return None, f"no FETCH_SIZE / WRITE_SIZE rows for {kernel}"
    return round((2.0 * f[2] + w[2]) * 1024.0, 1), (f"kernel {kernel}, grid {f[0]}, {f[1]} launches; (2 x FETCH_SIZE {f[2]:.1f} KB + WRITE_SIZE {w[2]:.1f} KB) x 1024; "
                                                   "this kernel only")


def pmc_valu_issue(pmc, kernel, n_simd=1024, ghz=2.4):
    """Fraction of the chip's vector-issue slots the kernel fills: SQ_INSTS_VALU wave-instructions x 4 clocks each, spread over
    1024 SIMDs, against the kernel's duration in the same (profiled) run."""
    v = _pmc_row(pmc, "SQ_INSTS_VALU", kernel)
    if not v or not v[3]:
        return None
    issue_us = v[2] / n_simd * 4.0 / (ghz * 1e3)
    return {"SQ_INSTS_VALU": round(v[2]), "issue_us_at_2p4GHz": round(issue_us, 2), "duration_us_profiled": round(v[3] / 1e3, 2),
            "valu_issue_frac": round(issue_us / (v[3] / 1e3), 3)}


def cpu_baseline(kind, W, Wo, steps, ds):
    """The CPU oracle on the same workload, on this box's host cores (solve single-threaded like Ceres with
    num_threads=1, marginalization on 4 threads like the reference).  Bounded sample: `steps` solves."""
    import subprocess

    from lio_amd import capi

    so = os.path.join(ROOT, "oracle", "liblio_oracle.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    orc = capi.LioLib(so)
    t = time.perf_counter()
    clouds, pp_ms = feature_clouds(orc, ds)
    est = make_estimator(orc, ds, clouds, kind, W, Wo)
    ts = []
    rep = None
    for _ in range(steps):
        est.restore()
        t = time.perf_counter()
        rep = est.solve()
        ts.append(time.perf_counter() - t)
    med = float(np.median(ts))
    # the same window with the reference's 0.10 s solver cap (Estimator.cc:1921) left ON, as shipped: Ceres stops at the first
    # iteration boundary past the cap, so the CPU reference does fewer iterations than the uncapped (parity) configuration
    capped = None
    try:
        from lio_amd import pipeline

        cfg = pipeline.config_outdoor64(orc, W, Wo, parity=False) if kind == "outdoor" else pipeline.config_indoor(orc, W, Wo, parity=False)
        if kind != "outdoor":
            cfg.cutoff_deskew, cfg.keep_features, cfg.prior_factor = 1, 0, 1
        pipeline.set_extrinsic(cfg, ds)
        est_c = capi.Estimator(orc, cfg)
        pipeline.init_window(est_c, orc, ds, [c[0] for c in clouds], pos_sigma=0.01, rot_sigma=0.001, vel_sigma=0.01)
        est_c.solve(); est_c.slide()
        for k in range(W + 1, len(ds.frames) - 1):
            pipeline.feed_frame(est_c, ds, k, clouds[k][0], clouds[k][1])
        f = ds.frames[-1]
        for j in range(f.imu_dt.shape[0]):
            est_c.process_imu(float(f.imu_dt[j]), f.imu_acc[j], f.imu_gyr[j], float(f.imu_t[j]))
        est_c.push_frame(capi.TransformF.make([0, 0, 0, 1], [0, 0, 0]), clouds[-1][0], clouds[-1][1], f.t)
        est_c.snapshot()
        tc, rc = [], None
        for _ in range(max(3, steps // 2)):
            est_c.restore()
            t = time.perf_counter()
            rc = est_c.solve()
            tc.append(time.perf_counter() - t)
        mc = float(np.median(tc))
        capped = {"value": round(1.0 / mc, 4), "unit": "solves/s", "median_ms": round(mc * 1e3, 1), "solver_iterations": int(rc.iterations), "termination": int(rc.termination),
                  "note": "max_solver_time = 0.10 s as shipped (Estimator.cc:1921); the product under the same cap runs all 10 iterations (its solve takes < 1 ms), i.e. the headline value applies to it unchanged"}
    except Exception as e:  # noqa: BLE001
        capped = {"error": f"{type(e).__name__}: {e}"}
    model = ""
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {
        "value": round(1.0 / med, 4),
        "unit": "solves/s",
        "cores": 4,
        "kind": "port",
        "sample": f"{steps} SolveOptimization calls of the CPU oracle on the same window (median {med * 1e3:.1f} ms; solve 1 thread, marginalization 4 threads); host has {os.cpu_count()} logical cores, {model}",
        "n_lidar_residuals": int(rep.n_lidar_residuals),
        "stages_ms": {"t_build_map": round(rep.ms_build_map, 3), "feature_cost": round(rep.ms_features, 3), "t_opt": round(rep.ms_opt, 3), "whole_marginalization": round(rep.ms_marg, 3)},
        "point_processor_ms_per_scan": round(float(np.median(pp_ms[1:])), 3),
        "point_mapping": mapping_ms_per_scan(orc, ds, clouds, n_frames=5),
        "point_odometry_ms_per_scan": odometry_ms_per_scan(orc, ds)[0] if kind == "outdoor" else None,   # PointOdometry::Process on the same four sweeps (oracle/odometry.h)
        "as_shipped_with_0p1s_solver_cap": capped,
        "note": "the oracle has none of the reference's ROS/PCL/Ceres/heap overheads: a faster-than-reference, conservative baseline; the reference itself cannot be built here (Eigen/PCL/Ceres/ROS absent)",
    }


if __name__ == "__main__":
    main()
