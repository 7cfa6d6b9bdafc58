#!/usr/bin/env python
"""bench.py — sliding-window solves/sec on the BASELINE.json workload.

One "step" = one Estimator::SolveOptimization (BuildLocalMap + kNN/plane features + newest-frame GN +
<= 10 dogleg iterations + marginalization; Estimator.cc:1648-2438) on a steady-state window snapshot of
synthetic HDL-64E data (64 rings, ~133 k points/scan), window_size 15 / opt_window_size 5, with the
clouds already resident in HBM.  N > 1: every rank owns an independent window (weak scaling, no
data-path collective — SURVEY.md §8e: the all-reduce of normal equations does not pay at this factor
count); value = N * K / max-over-ranks time.

Prints ONE JSON line (rank 0).  Extra objects: `roofline` for the dominant kernel (HIP events on the
estimator's stream), `cpu_baseline` = the CPU oracle timed on this box's host cores on the same window.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "lio-mapping_amd"))

import numpy as np  # noqa: E402


def build_window(lib, kind, W, Wo, extra_frames, seed_shift=0.0, lidar=None, ds=None):
    from lio_amd import capi, pipeline, synth

    n_frames = W + 1 + extra_frames
    frame_dt = 0.3 if kind == "outdoor" else 0.2  # odom_io = 3 (HDL-64) / 2 (VLP-16) x 0.1 s
    if ds is None:
        ds = synth.make_dataset(kind, n_frames, frame_dt, t0=1.0 + seed_shift, lidar=lidar)
    t0 = time.time()
    pp = capi.PointProcessor(lib, ds.lidar.lower_deg, ds.lidar.upper_deg, ds.lidar.rings)
    clouds = []
    pp_ms = []
    for f in ds.frames:
        t = time.perf_counter()
        pp.process(f.scan)
        pp_ms.append((time.perf_counter() - t) * 1e3)
        clouds.append((pp.cloud(4), pp.cloud(2)))
    cfg = pipeline.config_outdoor64(lib, W, Wo) if kind == "outdoor" else pipeline.config_indoor(lib, W, Wo)
    if kind != "outdoor":
        cfg.cutoff_deskew, cfg.keep_features, cfg.prior_factor = 1, 0, 1
    pipeline.set_extrinsic(cfg, ds)
    est = capi.Estimator(lib, cfg)
    pipeline.init_window(est, lib, ds, [c[0] for c in clouds], pos_sigma=0.01, rot_sigma=0.001, vel_sigma=0.01)
    est.solve()
    est.slide()
    last = None
    for k in range(W + 1, n_frames - 1):
        last = pipeline.feed_frame(est, ds, k, clouds[k][0], clouds[k][1])
    # stop right before the last frame's solve: push it, then snapshot the full window
    k = n_frames - 1
    f = ds.frames[k]
    for j in range(f.imu_dt.shape[0]):
        est.process_imu(float(f.imu_dt[j]), f.imu_acc[j], f.imu_gyr[j], float(f.imu_t[j]))
    return ds, clouds, est, k, pp_ms, time.time() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="hdl64", choices=["hdl64", "vlp16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shard-factors", action="store_true",
                    help="strong-scaling mode: ONE window, its lidar factors sharded over the ranks, RCCL all-reduce of the normal-equation "
                         "moments per linearisation (SURVEY.md §8e).  Default is weak scaling: one independent window per rank, no collective.")
    ap.add_argument("--cpu-steps", type=int, default=8)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from lio_amd import capi, dist_util, pipeline, synth

    hip = capi.load_hip()
    kind = "outdoor" if args.workload == "hdl64" else "indoor"
    W, Wo = 15, 5
    ds, clouds, est, k_last, pp_ms, setup_s = build_window(hip, kind, W, Wo, extra_frames=4, seed_shift=0.0 if args.shard_factors else dist_util.window_shift_for_rank(rank))
    if args.shard_factors and world > 1:
        est.set_factor_sharding(rank, world, dist_util.make_allreduce("cuda"))

    # The step under test is the SolveOptimization that ProcessLaserOdom runs for the last frame: push that
    # frame (upload + VoxelGrid + window push, untimed), snapshot, then time restore + SolveOptimization with
    # every cloud already resident in HBM.
    T = capi.TransformF.make([0, 0, 0, 1], [0, 0, 0])
    est.push_frame(T, clouds[k_last][0], clouds[k_last][1], ds.frames[k_last].t)
    new_stack_n = est.get_surf_stack(W).shape[0]
    est.snapshot()

    def one_step():
        est.restore()
        return est.solve()

    for _ in range(args.warmup):
        rep = one_step()
    est.enable_kernel_timing(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rep = one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    value_all, dt_max = dist_util.aggregate_throughput(args.steps, dt, world, device="cuda")
    if args.shard_factors:
        value_all = args.steps / dt_max  # one window solved cooperatively: total work is fixed

    names = ["features", "odom_features", "odom_rows", "odom_update", "moments", "voxel", "knn_grid", "concat"]
    kt = {n: est.kernel_timing(n) for n in names}
    est.enable_kernel_timing(False)

    out = None
    if rank == 0:
        value = value_all
        dom = max(("features", "odom_features", "moments", "odom_rows"), key=lambda n: kt[n]["total_ms"])
        d = kt[dom]
        avg_ms = d["total_ms"] / max(d["launches"], 1)
        achieved = (d["algorithmic_bytes"] / max(d["launches"], 1)) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        roofline = {
            "kernel": {"features": "k_features", "odom_features": "k_features", "moments": "k_lidar_moments", "odom_rows": "k_odom_rows"}[dom],
            "stage": dom,
            "bound": "hbm",
            "achieved": round(achieved, 3),
            "peak": 8000.0,
            "unit": "GB/s",
            "frac": round(achieved / 8000.0, 6),
            "traffic": None,
            "avg_launch_us": round(avg_ms * 1e3, 3),
            "launches": d["launches"],
            "algorithmic_bytes_per_launch": round(d["algorithmic_bytes"] / max(d["launches"], 1), 1),
        }
        # HBM traffic of that kernel from the committed rocprofv3 PMC passes (profiles/pmc_summary.py: FETCH_SIZE and
        # WRITE_SIZE collected in separate runs, (2*FETCH + WRITE) * 1024 — the gfx950 half-count correction for wide
        # coalesced reads).  null when no PMC summary has been committed for this kernel.
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc.json")))
            ent = max(pmc.get("lio::" + roofline["kernel"], []), key=lambda e: e["launches"], default=None)
            if ent:
                roofline["traffic"] = round(ent["hbm_bytes_corrected"], 1)
                roofline["traffic_source"] = "profiles/r1_c_pmc_hbm_traffic.md (per launch, most frequent grid size)"
        except (OSError, ValueError):
            pass
        roofline["others"] = {
            n: {"avg_launch_us": round(1e3 * kt[n]["total_ms"] / max(kt[n]["launches"], 1), 3),
                "achieved_GBps": round((kt[n]["algorithmic_bytes"] / max(kt[n]["launches"], 1)) / max(kt[n]["total_ms"] / max(kt[n]["launches"], 1) * 1e-3, 1e-12) / 1e9, 2)}
            for n in ("features", "odom_features", "moments", "voxel", "knn_grid") if n != dom
        }
        cpu = None
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(kind, W, Wo, args.cpu_steps, ds)
        out = {
            "metric": "sliding-window solves/sec, 64-line 130k-pt scans, window=15 (opt_window=5)",
            "value": round(value, 3),
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt_max / args.steps, 4),
            "higher_is_better": True,
            "scaling": "strong" if args.shard_factors else "weak",
            "vs_baseline": None,
            "dtype": "f64 (solve) / f32 (features)",
            "data": "synthetic",
            "config": {
                "workload": "HDL-64E outdoor_test_config_64, S_outdoor ray-cast scans, window_size=15 opt_window_size=5, one SolveOptimization per step, clouds resident in HBM"
                if kind == "outdoor"
                else "VLP-16 indoor, window_size=15 opt_window_size=5",
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
                "point_processor_incl_h2d_d2h": round(float(np.median(pp_ms[1:])), 4),
                "estimator_step_amortised_over_odom_io": round(1e3 * dt_max / args.steps / (3 if kind == "outdoor" else 2), 4),
                "note": "PointOdometry (scan-to-scan) is not part of this round; after IMU init the reference disables it (SURVEY.md A.18)",
            },
            "setup_s": round(setup_s, 2),
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(kind, W, Wo, steps, ds=None):
    """The CPU oracle on the same workload, on this box's host cores (solve single-threaded like Ceres with
    num_threads=1, marginalization on 4 threads like the reference).  Bounded sample: `steps` solves."""
    import subprocess

    from lio_amd import capi

    so = os.path.join(ROOT, "oracle", "liblio_oracle.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    orc = capi.LioLib(so)
    ds, clouds, est, k_last, _, _ = build_window(orc, kind, W, Wo, extra_frames=4, ds=ds)
    T = capi.TransformF.make([0, 0, 0, 1], [0, 0, 0])
    est.push_frame(T, clouds[k_last][0], clouds[k_last][1], ds.frames[k_last].t)
    est.snapshot()
    ts = []
    rep = None
    for _ in range(steps):
        est.restore()
        t = time.perf_counter()
        rep = est.solve()
        ts.append(time.perf_counter() - t)
    med = float(np.median(ts))
    ncpu = os.cpu_count()
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
        "sample": f"{steps} SolveOptimization calls of the CPU oracle on the same window (median {med * 1e3:.1f} ms; solve 1 thread, marginalization 4 threads); host has {ncpu} logical cores, {model}",
        "n_lidar_residuals": int(rep.n_lidar_residuals),
        "stages_ms": {"t_build_map": round(rep.ms_build_map, 3), "feature_cost": round(rep.ms_features, 3), "t_opt": round(rep.ms_opt, 3), "whole_marginalization": round(rep.ms_marg, 3)},
        "note": "the oracle has none of the reference's ROS/PCL/Ceres/heap overheads: a faster-than-reference, conservative baseline; the reference itself cannot be built here (Eigen/PCL/Ceres/ROS absent)",
    }


if __name__ == "__main__":
    main()
