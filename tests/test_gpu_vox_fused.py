"""pcl::VoxelGrid as ONE launch (csrc/cloud_kernels.hip: k_vox_fused — a counting sort over the cloud's own box of cells, four grid
barriers inside one kernel) against the oracle, bit for bit, and the hand-over to the sorted path where the one-launch form cannot run.

Same voxel order (PCL's index i0 + i1 d0 + i2 d0 d1 ascending) and the same within-voxel summation order (ascending original index)
as the sorted path, hence `assert_array_equal` against the oracle throughout.  lio_vox_fused_stats counts the filters that took the
one-launch form and those that handed the cloud back, so every case also asserts WHICH path ran (the product must not pass on a silent
fallback): regular clouds stay on the one-launch form (with the box of cells handed to the launch once the handle has seen a cloud, and a
second launch with the bounds taken inside the kernel when a point falls outside it); a crowded voxel (> 32 points), a box of more cells than the counter table, a
cloud without a finite point and a cloud above the grid's register capacity go to the sorted path; and a regular cloud right after each
of those is filtered by the one-launch form again with a clean counter table."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _one_launch_form_on(hip):
    """the one-launch form is opt-in (LIO_VOX_FUSED=1): these tests switch it on through the C-ABI hook"""
    before = hip.vox_fused_set(1)
    yield
    hip.vox_fused_set(before)


def _cloud(rng, n, extent=40.0, height=6.0):
    pts = np.zeros((n, 4), np.float32)
    pts[:, 0] = rng.uniform(-extent, extent, n)
    pts[:, 1] = rng.uniform(-extent, extent, n)
    pts[:, 2] = np.where(rng.random(n) < 0.6, rng.normal(0, 0.03, n), rng.uniform(0, height, n))
    pts[:, 3] = rng.uniform(0, 64, n)
    return pts


def _hands_over(pts, leaf):
    """what k_vox_fused decides from the cloud: more than 32 points in a voxel, a box of more cells than the counter table
    (64 M, counted in whole 2048-cell stretches), or no finite point"""
    ok = np.isfinite(pts[:, :3]).all(axis=1)
    if not ok.any():
        return True
    inv = np.float32(1.0) / np.float32(leaf)
    c = np.floor(pts[ok, :3] * inv).astype(np.int64)
    lo, hi = np.floor(pts[ok, :3].min(axis=0) * inv).astype(np.int64), np.floor(pts[ok, :3].max(axis=0) * inv).astype(np.int64)
    d = hi - lo + 1
    if (int(d[0]) * int(d[1]) * int(d[2]) + 2047) // 2048 * 512 > (16 << 20):
        return True
    key = (c[:, 0] - lo[0]) + d[0] * ((c[:, 1] - lo[1]) + d[1] * (c[:, 2] - lo[2]))
    return bool(np.unique(key, return_counts=True)[1].max() > 32)


def _run(hip, oracle, pts, leaf, expect_fused, expect_fallback):
    a0, b0 = hip.vox_fused_stats()
    out, ref = hip.voxel_grid(pts, leaf), oracle.voxel_grid(pts, leaf)
    a1, b1 = hip.vox_fused_stats()
    np.testing.assert_array_equal(out, ref)
    # launches: one, or two when a point lay outside the box that came with the first (the union of the boxes the handle has seen so far,
    # a few cells wider: no bounds phase then) and the filter ran again with the bounds taken inside the kernel
    launches = (1, 2) if expect_fused else (0,)
    assert a1 - a0 in launches and b1 - b0 == int(expect_fallback), ("one-launch filters / hand-overs", a1 - a0, b1 - b0)
    return len(ref)


@pytest.mark.parametrize("n,leaf,extent", [(1, 0.4, 40.0), (63, 0.4, 40.0), (5000, 0.4, 40.0), (150000, 0.4, 100.0), (260000, 0.4, 100.0),
                                           (40000, 0.2, 30.0), (90000, 0.8, 100.0)])
def test_one_launch_filter_matches_oracle(hip, oracle, n, leaf, extent):
    rng = np.random.default_rng(1000 + n)
    pts = _cloud(rng, n, extent)
    if n > 100:
        pts[rng.integers(0, n, 25), rng.integers(0, 3, 25)] = np.nan          # non-finite points are skipped (B.1)
    m = _run(hip, oracle, pts, leaf, True, False)
    print("n", n, "leaf", leaf, "->", m, "voxels (one-launch form)")


def test_same_handle_many_clouds_in_a_row(hip, oracle):
    """the counter table is cleaned by the points themselves: forty different clouds through the same handle"""
    rng = np.random.default_rng(7)
    kept = 0
    for k in range(40):
        n = int(rng.integers(1, 120000))
        pts, leaf = _cloud(rng, n, float(rng.uniform(5, 120)), float(rng.uniform(1, 30))), float(rng.choice([0.2, 0.4, 0.8]))
        over = _hands_over(pts, leaf)          # small, dense members put more than 32 points into a voxel
        kept += 0 if over else 1
        _run(hip, oracle, pts, leaf, True, over)
    assert kept >= 20, kept


def test_box_from_the_previous_clouds(hip, oracle):
    """the second filter of a handle gets its box of cells with the launch (no bounds phase); a cloud that leaves it is filtered again"""
    rng = np.random.default_rng(23)
    base = _cloud(rng, 60000, 50.0)
    _run(hip, oracle, base, 0.4, True, False)
    a0, _ = hip.vox_fused_stats()
    _run(hip, oracle, base + np.array([0.3, -0.2, 0.1, 0.0], np.float32), 0.4, True, False)     # inside the margin: one launch
    a1, _ = hip.vox_fused_stats()
    assert a1 - a0 == 1
    moved = base + np.array([40.0, 0.0, 0.0, 0.0], np.float32)                                   # leaves the box: two launches, same result
    _run(hip, oracle, moved, 0.4, True, False)
    a2, _ = hip.vox_fused_stats()
    assert a2 - a1 == 2
    _run(hip, oracle, base, 0.4, True, False)                                                    # the union holds both now
    _run(hip, oracle, moved, 0.4, True, False)
    a3, _ = hip.vox_fused_stats()
    assert a3 - a2 == 2
    _run(hip, oracle, base, 0.2, True, False)                                                    # another leaf: the box starts over


def test_hand_over_to_the_sorted_path(hip, oracle):
    rng = np.random.default_rng(11)
    regular = _cloud(rng, 30000)
    _run(hip, oracle, regular, 0.4, True, False)
    # a crowded voxel: 200 points inside one 0.4 m cell
    crowded = _cloud(rng, 20000)
    crowded[:200, :3] = (np.array([3.25, 3.25, 1.25]) + rng.uniform(0, 0.1, (200, 3))).astype(np.float32)
    _run(hip, oracle, crowded, 0.4, True, True)
    _run(hip, oracle, regular, 0.4, True, False)
    # exactly at the limit: 32 points in one cell stay on the one-launch form, 33 do not
    edge = _cloud(rng, 20000)
    edge[:32, :3] = (np.array([-7.3, 2.1, 2.9]) + rng.uniform(0, 0.05, (32, 3))).astype(np.float32)
    edge[32:, :3] += np.where(np.all(np.abs(edge[32:, :3] - np.array([-7.3, 2.1, 2.9])) < 0.5, axis=1, keepdims=True), 5.0, 0.0).astype(np.float32)
    _run(hip, oracle, edge, 0.4, True, False)
    edge[32, :3] = edge[0, :3]
    _run(hip, oracle, edge, 0.4, True, True)
    # a box of more cells than the counter table holds (64 M): 900 x 900 x 600 m at 0.4 m
    wide = _cloud(rng, 20000, 450.0)
    wide[:100, 2] = rng.uniform(-300, 300, 100).astype(np.float32)
    _run(hip, oracle, wide, 0.4, True, True)
    _run(hip, oracle, regular, 0.4, True, False)
    # no finite point at all
    nan = np.full((500, 4), np.nan, np.float32)
    _run(hip, oracle, nan, 0.4, True, True)
    # more points than the grid holds in registers (256 blocks x 1024 threads): the sorted path from the start
    big = _cloud(rng, 300000, 150.0)
    _run(hip, oracle, big, 0.4, False, False)
    _run(hip, oracle, regular, 0.4, True, False)


def test_one_launch_filter_timing(hip):
    """not a bound, a record: the filter on the 150 k-point local-map shape, as lio_bench_voxel_grid times it"""
    rng = np.random.default_rng(3)
    pts = _cloud(rng, 150000, 100.0)
    for n in (44000, 150000):
        sub = pts[:n]
        ms, m = hip.bench_voxel_grid(sub, 0.4, reps=50)
        hip.vox_fused_set(0)
        ms0, m0 = hip.bench_voxel_grid(sub, 0.4, reps=50)
        hip.vox_fused_set(1)
        assert m == m0
        print("VoxelGrid of %d points -> %d voxels: one launch %.1f us per filter, sorted path %.1f us" % (n, m, 1e3 * ms, 1e3 * ms0))
