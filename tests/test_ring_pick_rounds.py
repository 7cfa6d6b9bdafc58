"""Design check for the next form of k_ring_pick: the reference's sequential pick loops of a ring equal a
parallel-rounds construction of the lexicographically-first maximal independent set per (subregion, phase) group, truncated to
the quota.  Pure numpy models (tests/ring_pick_model.py); no GPU, no oracle."""
import numpy as np
import pytest

from ring_pick_model import greedy, rounds


@pytest.mark.parametrize("seed", range(12))
def test_parallel_rounds_equal_the_sequential_picks(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(200, 2100))                       # VLP-16 rings have ~1800 points, HDL-64E ~2080
    curv = rng.gamma(0.6, 0.2, n).astype(np.float32)
    curv[rng.random(n) < 0.1] = np.float32(0.05)           # ties in the curvature: the index decides
    p_gap = [0.98, 0.9, 0.6][seed % 3]
    gap_ok = np.concatenate([rng.random(n - 1) < p_gap, [False]])
    mask0 = (rng.random(n) < [0.02, 0.2, 0.5][(seed // 3) % 3]).astype(np.int32)
    th = 0.1
    pa, ma = greedy(curv, gap_ok, mask0, th)
    pb, mb, nr = rounds(curv, gap_ok, mask0, th)
    assert pa == pb
    np.testing.assert_array_equal(ma, mb)
    assert sum(len(g[2]) for g in pa) > 0
    # what a kernel pays: rounds per group instead of up to 20 + 4 dependent picks per subregion
    assert max(nr) <= 12, nr
    print(f"seed {seed}: n {n}, picks {sum(len(g[2]) for g in pa)}, rounds per group max {max(nr)} mean {np.mean(nr):.2f}")


def test_quota_binds_and_later_groups_see_only_the_kept_masks():
    """Dense corners: every subregion has far more than 20 corner candidates, so the truncation matters — a member of the
    independent set beyond the quota must NOT mask the flats or the next subregion."""
    rng = np.random.default_rng(99)
    n = 1200
    curv = np.where(rng.random(n) < 0.7, rng.uniform(0.2, 5.0, n), rng.uniform(0.0, 0.09, n)).astype(np.float32)
    gap_ok = np.concatenate([rng.random(n - 1) < 0.95, [False]])
    mask0 = np.zeros(n, np.int32)
    pa, ma = greedy(curv, gap_ok, mask0, 0.1)
    pb, mb, nr = rounds(curv, gap_ok, mask0, 0.1)
    assert pa == pb and np.array_equal(ma, mb)
    assert all(len(g[2]) == 20 for g in pa if g[0] == "corner")


def test_worst_case_is_a_monotone_ramp():
    """A curvature ramp over connected points is the adversary: every candidate is out-ranked by its left neighbour, so a round
    admits one member per connected stretch and a group needs ~ region / (nc + 1) rounds.  The kernel therefore needs a bound on
    the rounds with the sequential walk behind it; on scans (next test) a group takes a handful."""
    n = 700
    curv = np.linspace(5.0, 0.2, n).astype(np.float32)
    gap_ok = np.concatenate([np.ones(n - 1, bool), [False]])
    pa, ma = greedy(curv, gap_ok, np.zeros(n, np.int32), 0.1)
    pb, mb, nr = rounds(curv, gap_ok, np.zeros(n, np.int32), 0.1)
    assert pa == pb and np.array_equal(ma, mb)
    assert max(nr) >= 15


def test_rounds_on_ray_cast_rings(oracle):
    """Rings of the synthetic VLP-16 / HDL-64E sweeps (curvature from the reference's formula, gaps from the 0.05 m^2 rule, no
    PrepareRing masks — they only remove candidates): same picks from both models, and the round counts a kernel would see."""
    from lio_amd import capi, synth

    worst, total, groups = 0, 0, 0
    for kind in ("indoor", "outdoor"):
        ds = synth.make_dataset(kind, 1, 0.1)
        pp = capi.PointProcessor(oracle, ds.lidar.lower_deg, ds.lidar.upper_deg, ds.lidar.rings)
        pp.process(ds.frames[0].scan)
        rings = pp.cloud(capi.PointProcessor.RINGS) if hasattr(capi.PointProcessor, "RINGS") else pp.cloud(0)
        offs = pp.ring_offsets()
        for r in list(range(0, ds.lidar.rings, max(1, ds.lidar.rings // 6)))[:6]:
            P = rings[offs[r]:offs[r + 1], :3].astype(np.float32)
            n = len(P)
            if n < 100:
                continue
            c = np.zeros(n, np.float32)
            acc = -10.0 * P[5:n - 5]
            for q in range(1, 6):
                acc = acc + P[5 + q:n - 5 + q] + P[5 - q:n - 5 - q]
            c[5:n - 5] = (acc * acc).sum(axis=1)
            d = P[1:] - P[:-1]
            gap_ok = np.concatenate([(d * d).sum(axis=1) <= 0.05, [False]])
            pa, ma = greedy(c, gap_ok, np.zeros(n, np.int32), 0.1)
            pb, mb, nr = rounds(c, gap_ok, np.zeros(n, np.int32), 0.1)
            assert pa == pb and np.array_equal(ma, mb)
            worst, total, groups = max(worst, max(nr)), total + sum(nr), groups + len(nr)
    print(f"ray-cast rings: {groups} groups, rounds per group mean {total / max(groups, 1):.2f}, worst {worst}")
    assert groups >= 50 and worst <= 40
