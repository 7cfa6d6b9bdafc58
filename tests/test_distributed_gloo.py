"""N>1 path of bench.py on CPU: world_size 2, gloo.  Each rank owns an independent window (weak scaling,
no data-path collective); the collective part is only the timing contract (barrier, MAX over ranks,
whole-job aggregate).  The per-rank work here is the CPU oracle on a tiny window — the HIP product cannot
run without a GPU, and tests are allowed to use the oracle."""
import os
import socket
import sys
import time

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, os.path.join(ROOT, "lio-mapping_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from lio_amd import capi, dist_util, pipeline, synth

    dist.init_process_group("gloo", rank=rank, world_size=world)
    r, lr, w = dist_util.rank_info()
    assert (r, w) == (rank, world)
    lib = capi.LioLib(os.path.join(ROOT, "oracle", "liblio_oracle.so"))
    lidar = synth.Lidar(16, -15, 15, 450)
    ds = synth.make_dataset("indoor", 5, 0.2, t0=1.0 + dist_util.window_shift_for_rank(rank), lidar=lidar)
    clouds = [pipeline.feature_clouds(lib, ds.lidar, f.scan)[0] for f in ds.frames]
    cfg = pipeline.config_indoor(lib, 4, 2)
    cfg.cutoff_deskew, cfg.keep_features, cfg.prior_factor = 1, 0, 1
    pipeline.set_extrinsic(cfg, ds)
    est = capi.Estimator(lib, cfg)
    pipeline.init_window(est, lib, ds, clouds, pos_sigma=0.005, rot_sigma=0.0005, vel_sigma=0.005)
    est.snapshot()
    K = 2
    dist_util.barrier(world)
    t0 = time.perf_counter()
    for _ in range(K):
        est.restore()
        rep = est.solve()
    dt = time.perf_counter() - t0 + 0.05 * rank  # make the ranks measurably different
    dist_util.barrier(world)
    value, tmax = dist_util.aggregate_throughput(K, dt, world)
    out[rank] = (value, tmax, dt, float(est.get_window()["Ps"][0, 0]), rep.n_lidar_residuals)
    dist.destroy_process_group()


def test_two_rank_weak_scaling_contract(oracle):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    (v0, t0, d0, p0, n0), (v1, t1, d1, p1, n1) = out[0], out[1]
    assert t0 == t1 == max(d0, d1)                 # MAX over ranks reached every rank
    assert np.isclose(v0, world * 2 / max(d0, d1))  # whole-job aggregate: N*K / max time
    assert v0 == v1
    assert p0 != p1 and n0 > 100 and n1 > 100       # the two ranks really worked on different windows


def _shard_worker(rank, world, port, out):
    sys.path.insert(0, os.path.join(ROOT, "lio-mapping_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from lio_amd import capi, dist_util, pipeline, synth

    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = capi.LioLib(os.path.join(ROOT, "oracle", "liblio_oracle.so"))
    ds = synth.make_dataset("indoor", 6, 0.2, lidar=synth.Lidar(16, -15, 15, 450))  # the SAME window on every rank
    clouds = [pipeline.feature_clouds(lib, ds.lidar, f.scan) for f in ds.frames]
    cfg = pipeline.config_indoor(lib, 4, 2)
    cfg.cutoff_deskew, cfg.keep_features, cfg.prior_factor = 1, 0, 1
    pipeline.set_extrinsic(cfg, ds)
    est = capi.Estimator(lib, cfg)
    pipeline.init_window(est, lib, ds, [c[0] for c in clouds], pos_sigma=0.005, rot_sigma=0.0005, vel_sigma=0.005)
    est.set_factor_sharding(rank, world, dist_util.make_allreduce("cpu"))
    r1 = est.solve()
    est.slide()
    r2 = pipeline.feed_frame(est, ds, 5, clouds[5][0], clouds[5][1])
    w = est.get_window()
    out[rank] = (w["Ps"].copy(), r1.final_cost, r2.final_cost, r2.marginalized, est.prior()["JtJ"].copy())
    dist.destroy_process_group()


def test_factor_sharding_allreduce_matches_single_rank(oracle):
    """residuals partitioned over ranks, [H|g] summed by all-reduce, every rank takes the same step.
    Two gloo ranks must (a) agree bit for bit with each other and (b) reproduce the unsharded solve."""
    from lio_amd import capi, pipeline, synth

    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_shard_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    (p0, c0, d0, m0, j0), (p1, c1, d1, m1, j1) = out[0], out[1]
    np.testing.assert_array_equal(p0, p1)       # lockstep replicas
    assert c0 == c1 and d0 == d1 and m0 == m1 == 1
    # unsharded reference run
    ds = synth.make_dataset("indoor", 6, 0.2, lidar=synth.Lidar(16, -15, 15, 450))
    clouds = [pipeline.feature_clouds(oracle, ds.lidar, f.scan) for f in ds.frames]
    cfg = pipeline.config_indoor(oracle, 4, 2)
    cfg.cutoff_deskew, cfg.keep_features, cfg.prior_factor = 1, 0, 1
    pipeline.set_extrinsic(cfg, ds)
    est = capi.Estimator(oracle, cfg)
    pipeline.init_window(est, oracle, ds, [c[0] for c in clouds], pos_sigma=0.005, rot_sigma=0.0005, vel_sigma=0.005)
    r1 = est.solve()
    est.slide()
    r2 = pipeline.feed_frame(est, ds, 5, clouds[5][0], clouds[5][1])
    np.testing.assert_allclose(p0, est.get_window()["Ps"], atol=1e-7)   # summation order differs across shards only
    np.testing.assert_allclose([c0, d0], [r1.final_cost, r2.final_cost], rtol=1e-7)
    np.testing.assert_allclose(j0, est.prior()["JtJ"], rtol=1e-5, atol=1e-6 * np.abs(j0).max())


def _kf_worker(rank, world, port, out):
    sys.path.insert(0, os.path.join(ROOT, "lio-mapping_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from kf_util import keyframe_inputs
    from lio_amd import capi, dist_util

    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = capi.LioLib(os.path.join(ROOT, "oracle", "liblio_oracle.so"))
    maps, kfs = keyframe_inputs(lib, "indoor", 3, 2)     # deterministic: the same 4 + 1 keyframes on every rank
    kfs = kfs + [kfs[0]]                                  # odd count: the ranks' shares differ in size
    r = dist_util.refine_keyframes_sharded(lib, maps, kfs, world, rank)
    out[rank] = (r["q"], r["p"], r["iterations"], r["rows"])
    if rank == 0:
        full = dist_util.refine_keyframes_sharded(lib, maps, kfs, 1, 0)
        out["full"] = (full["q"], full["p"], full["iterations"], full["rows"])
    dist.destroy_process_group()


def test_keyframe_batch_shards_over_ranks(oracle):
    """configs[4] at N = 2: keyframes split round-robin, no collective on the refinement itself, one all-gather of the
    poses; every rank ends with all poses, identical to the single-rank batch."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_kf_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for a, b, c in zip(out[0], out[1], out["full"]):
        np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(a, c)
    assert out[0][0].shape == (5, 4) and np.all(out[0][2] > 0)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher must start two ranks itself (torch.distributed.run on 127.0.0.1) and print ONE
    line with n_gpus = 2 plus the `sharded` and `keyframes` extras; --dry-launch runs that control flow on gloo with the CPU
    oracle as the worker.  Without --dry-launch and without GPUs the launcher refuses loudly instead of running one rank."""
    import json
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch", "--steps", "1"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                      # rank 0 only
    out = json.loads(lines[0])
    assert out["dry_launch"] and out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["parallelism"] == "2 independent windows"
    # N > 1: the line's value is the throughput mode (every rank solves a batch of windows per step through lio_est_batch, parity-gated)
    thr = out["throughput_mode"]
    assert thr["parity"] == "ok" and thr["ranks"] == 2 and thr["windows_per_rank"] == 2 and thr["value"] > 0
    assert out["value"] == thr["value"] and out["ms_per_step"] == thr["ms_per_step"] and out["single_window"]["value"] > 0
    sh, kf = out["sharded"], out["keyframes"]
    assert sh["scaling"] == "strong" and sh["value"] > 0 and sh["linearisations_per_solve"] == sh["solver_iterations"] + 1
    assert sh["solver_iterations"] == sh["solver_iterations_unsharded"] and sh["final_cost_rel_gap_to_unsharded"] < 1e-9
    assert kf["scaling"] == "strong" and kf["keyframes"] == 4 and kf["value"] > 0
    import torch

    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
