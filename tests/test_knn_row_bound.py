"""The one-lane-per-query K-NN walk (csrc/cloud_kernels.hip: knn_scan_group, LPQ <= 2) skips a row of cells when a lower bound
of the distance from the query to that row exceeds the current fifth-best distance.  The search stays exact only if the bound
really is a lower bound of the fp32 squared distance the kernel computes for EVERY point binned into that row — including the
rounding of the cell arithmetic (the 1e-3-cell margin) and the points the grid clamps into its border rows.  This replays
that arithmetic in numpy float32 (CPU, no GPU needed)."""
import numpy as np

f32 = np.float32


def _bound_terms(q, inv_cell):
    u = (q * inv_cell).astype(f32)
    f = (u - np.floor(u)).astype(f32)
    cell = f32(1.0) / inv_cell
    lo = (np.maximum(f - f32(1e-3), f32(0)) * cell).astype(f32)        # distance to the rows below
    hi = (np.maximum(f32(1.0) - f - f32(1e-3), f32(0)) * cell).astype(f32)
    return lo, hi


def _check(extent, cell, n=400000, seed=0, clamp_rows=None):
    rng = np.random.default_rng(seed)
    cell = f32(cell)
    inv_cell = f32(1.0) / cell
    q = rng.uniform(-extent, extent, (n, 3)).astype(f32)
    # a point in one of the 8 neighbouring rows (dy, dz in {-1, 0, 1}, not both 0), anywhere along x within the 3-cell run
    d = rng.integers(-1, 2, (n, 2))
    d[(d == 0).all(1), 0] = 1
    cq = np.floor((q * inv_cell).astype(f32))
    p = np.empty_like(q)
    p[:, 0] = ((cq[:, 0] + rng.uniform(-1, 2, n)) / inv_cell).astype(f32)
    for k, col in enumerate((1, 2)):
        target = cq[:, col] + d[:, k]
        frac = rng.uniform(0, 1, n)
        frac[rng.random(n) < 0.2] = 0.0            # points right on the cell face
        frac[rng.random(n) < 0.1] = 1.0 - 1e-7
        v = ((target + frac) / inv_cell).astype(f32)
        # only keep what the grid would really bin into that row: floor(v * inv_cell) == target (or beyond it when clamped)
        c = np.floor((v * inv_cell).astype(f32))
        if clamp_rows is None:
            bad = c != target
        else:                                      # border row: everything at or beyond it lands there
            v = np.where(d[:, k] != 0, ((target + d[:, k] * rng.uniform(0, clamp_rows, n) + frac) / inv_cell).astype(f32), v)
            c = np.floor((v * inv_cell).astype(f32))
            bad = np.where(d[:, k] > 0, c < target, np.where(d[:, k] < 0, c > target, c != target))
        v = np.where(bad, ((target + 0.5) / inv_cell).astype(f32), v)
        p[:, col] = v
    lo_y, hi_y = _bound_terms(q[:, 1], inv_cell)
    lo_z, hi_z = _bound_terms(q[:, 2], inv_cell)
    ey = np.where(d[:, 0] < 0, lo_y, np.where(d[:, 0] > 0, hi_y, f32(0))).astype(f32)
    ez = np.where(d[:, 1] < 0, lo_z, np.where(d[:, 1] > 0, hi_z, f32(0))).astype(f32)
    bound = (ey * ey + ez * ez).astype(f32)
    dd = (p - q).astype(f32)
    dist = (dd[:, 0] * dd[:, 0]).astype(f32)
    dist = (dist + (dd[:, 1] * dd[:, 1]).astype(f32)).astype(f32)
    dist = (dist + (dd[:, 2] * dd[:, 2]).astype(f32)).astype(f32)
    assert np.all(bound <= dist), float((bound - dist).max())
    return float(np.mean(bound > f32(0.3) * cell * cell))   # how often a row would be skipped at a fifth-best of 0.3 cell^2


def test_row_bound_is_a_lower_bound_of_the_fp32_distance():
    for extent, cell in ((60.0, 1.0001), (400.0, 1.0001), (2000.0, 1.0001), (100.0, 5.0005), (30.0, 0.5)):
        frac = _check(extent, cell, seed=int(extent))
        assert frac > 0.2                          # the bound is not vacuous


def test_row_bound_holds_for_points_clamped_into_border_rows():
    _check(100.0, 1.0001, seed=5, clamp_rows=40.0)
