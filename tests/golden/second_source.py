"""Independent restatements ("second sources") of the third-party semantics the oracle restates, written
against the libraries' published behaviour with numpy / scipy — NOT against oracle/*.h.  tests/test_second_source.py compares
the oracle with them on the vectors make_second_source_vectors.py commits.  Eigen, PCL, FLANN and Ceres themselves are not in
this image, so these are second opinions by the same author, not the libraries: they catch slips of the restatement, not
misreadings shared by both.  What each one pins and what stays unpinned is listed per function.
"""
import numpy as np
import scipy.linalg


# ------------------------------------------------------------------------------------------------ B.4 Eigen colPivHouseholderQr().solve
def colpiv_qr_solve(A, b):
    """Eigen::ColPivHouseholderQR<Matrix<float,..>>(A).solve(b) through LAPACK's pivoted QR (scipy.linalg.qr(pivoting=True)).
    Eigen 3.3 counts the meaningful pivots k with (largest remaining squared column norm) >= (max column norm * eps)^2 / rows
    * (rows - k) and back-substitutes over those only (the remaining unknowns are 0, then the column permutation is undone).
    Pinned: the full-rank solution; the basic solution of exactly rank-deficient systems.  Unpinned: which side of the pivot
    threshold a column with a relative norm of ~1e-7 falls on (LAPACK and Eigen down-date column norms differently)."""
    A = np.asarray(A, np.float32)
    b = np.asarray(b, np.float32)
    m, n = A.shape
    Q, R, piv = scipy.linalg.qr(A.astype(np.float64), pivoting=True, mode="economic")
    eps = np.finfo(np.float32).eps
    max_norm = np.sqrt((A.astype(np.float64) ** 2).sum(0)).max()
    helper = (max_norm * eps) ** 2 / m
    k_used = 0
    for k in range(min(m, n)):
        remaining = (R[k:, k:] ** 2).sum(0).max() if R[k:, k:].size else 0.0   # trailing columns are what is left after k reflections
        if remaining < helper * (m - k):
            break
        k_used += 1
    y = np.zeros(n)
    if k_used:
        c = Q.T @ b.astype(np.float64)
        y[:k_used] = scipy.linalg.solve_triangular(R[:k_used, :k_used], c[:k_used])
    x = np.zeros(n)
    x[piv] = y
    return x.astype(np.float32), k_used


# ------------------------------------------------------------------------------------------------ MarginalizationFactor.cc:271-302 via numpy.linalg.eigh
def marginalize_schur(A, b, m, eps=1e-8):
    """Eigen::SelfAdjointEigenSolver steps of MarginalizationInfo::Marginalize with numpy.linalg.eigh: Amm^+ with eigenvalues
    <= eps zeroed, Schur complement, S = V diag(s) V^T -> linearized_jacobians = sqrt(s) V^T, linearized_residuals =
    sqrt(1/s) V^T b (eigenvalues <= eps -> 0).  Eigenvectors are sign / basis ambiguous, so the comparison is on the invariants
    J^T J, J^T r and |r|.  Pinned: the cut convention (strictly greater than 1e-8, absolute), the pseudo-inverse, the square-root
    factors.  Unpinned: eigenvalues within rounding of the cut."""
    A = np.asarray(A, np.float64)
    b = np.asarray(b, np.float64)
    Amm = 0.5 * (A[:m, :m] + A[:m, :m].T)
    w, V = np.linalg.eigh(Amm)
    winv = np.where(w > eps, 1.0 / np.where(w > eps, w, 1.0), 0.0)
    Amm_inv = (V * winv) @ V.T
    Arm, Amr, Arr = A[m:, :m], A[:m, m:], A[m:, m:]
    S = Arr - Arm @ Amm_inv @ Amr
    bs = b[m:] - Arm @ Amm_inv @ b[:m]
    s, V2 = np.linalg.eigh(S)
    keep = s > eps
    sq = np.where(keep, np.sqrt(np.where(keep, s, 0.0)), 0.0)
    isq = np.where(keep, 1.0 / np.sqrt(np.where(keep, s, 1.0)), 0.0)
    J = sq[:, None] * V2.T
    r = isq * (V2.T @ bs)
    return J, r, s


# ------------------------------------------------------------------------------------------------ B.1 pcl::VoxelGrid<PointXYZI>
def voxel_grid_pcl(pts, leaf):
    """pcl::VoxelGrid (PCL 1.8, downsample_all_data, no field filter) from its published algorithm, in float32 like PCL:
    min/max of the finite points, min_b = floor(min * inverse_leaf), div_b = max_b - min_b + 1, idx = ijk0 + ijk1 * div_b0 +
    ijk2 * div_b0 * div_b1 with ijk = floor(p * inverse_leaf) - min_b, one centroid (all four fields) per occupied voxel in
    ascending idx.  The within-voxel summation order is unspecified in PCL (std::sort is unstable); ascending input index is
    used here, which is also what the oracle and the product fix.  Returns (centroids float32, idx per output voxel)."""
    pts = np.asarray(pts, np.float32)
    fin = np.isfinite(pts[:, :3]).all(1)
    p = pts[fin]
    inv = np.float32(1.0) / np.float32(leaf)
    mn, mx = p[:, :3].min(0), p[:, :3].max(0)
    min_b = np.floor(mn * inv).astype(np.int64)
    max_b = np.floor(mx * inv).astype(np.int64)
    div = max_b - min_b + 1
    ijk = (np.floor(p[:, :3] * inv) - min_b.astype(np.float32)).astype(np.int64)
    idx = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    order = np.argsort(idx, kind="stable")
    idx_s, p_s = idx[order], p[order]
    starts = np.flatnonzero(np.r_[True, idx_s[1:] != idx_s[:-1]])
    ends = np.r_[starts[1:], len(idx_s)]
    out = np.zeros((len(starts), 4), np.float32)
    for k, (a, e) in enumerate(zip(starts, ends)):
        acc = np.zeros(4, np.float32)
        for row in p_s[a:e]:          # float32 accumulation in input order, then one division
            acc = acc + row
        out[k] = acc / np.float32(e - a)
    return out, idx_s[starts]


# ------------------------------------------------------------------------------------------------ B.3 Ceres 1.14 trust region, traditional dogleg
class CeresDogleg:
    """Transliteration of Ceres 1.14's TrustRegionMinimizer + DoglegStrategy (TRADITIONAL_DOGLEG) as documented in the Ceres
    solver docs and source comments, with the reference's options (Estimator.cc:1909-1921) and Ceres' defaults: Jacobi scaling
    from the first Jacobian (1 / (1 + sqrt(diag J^T J))), initial_trust_region_radius 1e4, min/max mu 1e-8 / 1, mu x10 on an
    invalid or failed step, min_relative_decrease 1e-3, radius halved on rejection or step quality < 0.25, radius = max(radius,
    3 |step|) above 0.75, mu = max(min_mu, 2 mu / 10) after an accepted step, function_tolerance 1e-6, parameter_tolerance 1e-8,
    gradient_tolerance 1e-10, max_num_consecutive_invalid_steps 5.  It works from the normal equations (H = J^T J, g = J^T r) of
    each linearisation and the cost of each candidate, which is what make_second_source_vectors.py dumps from the oracle."""

    def __init__(self, H0, g0, cost0):
        self.n = len(g0)
        self.scale = 1.0 / (1.0 + np.sqrt(np.diag(H0)))
        self.radius, self.mu = 1e4, 1e-8
        self.reuse = False
        self.invalid = 0
        self.set_linearisation(H0, g0, cost0)

    def set_linearisation(self, H, g, cost):
        s = self.scale
        self.H = H * np.outer(s, s)
        self.g = g * s
        self.cost = cost

    def compute_step(self):
        """-> (delta in the unscaled tangent space or None when the linear solve / model is invalid, model_cost_change)"""
        H, g = self.H, self.g
        if not self.reuse:
            self.reuse = True
            self.diag = np.sqrt(np.clip(np.diag(H), 1e-6, 1e32))
            self.gradient = g / self.diag
            sg = self.gradient / self.diag
            self.alpha = (self.gradient @ self.gradient) / (sg @ (H @ sg))
            ok = False
            while self.mu < 1.0:
                A = H + np.diag(self.diag ** 2 * self.mu)
                try:
                    c = scipy.linalg.cho_factor(A, lower=True, check_finite=False)
                    gn = scipy.linalg.cho_solve(c, g, check_finite=False)
                    ok = bool(np.all(np.isfinite(gn)))
                except scipy.linalg.LinAlgError:
                    ok = False
                if ok:
                    break
                self.mu *= 10.0
            if not ok:
                return None, 0.0
            self.gn = -gn * self.diag
        gnorm, gnn = np.linalg.norm(self.gradient), np.linalg.norm(self.gn)
        if gnn <= self.radius:
            step, self.step_norm = self.gn.copy(), gnn
        elif gnorm * self.alpha >= self.radius:
            step, self.step_norm = -(self.radius / gnorm) * self.gradient, self.radius
        else:
            b_dot_a = -self.alpha * (self.gradient @ self.gn)
            a_sq = (self.alpha * gnorm) ** 2
            bma = a_sq - 2 * b_dot_a + gnn ** 2
            c = b_dot_a - a_sq
            d = np.sqrt(c * c + bma * (self.radius ** 2 - a_sq))
            beta = (d - c) / bma if c <= 0 else (self.radius ** 2 - a_sq) / (d + c)
            step = (-self.alpha * (1 - beta)) * self.gradient + beta * self.gn
            self.step_norm = np.linalg.norm(step)
        step = step / self.diag
        model = -(step @ g + 0.5 * step @ (H @ step))
        if not model > 0:
            return None, model
        return step * self.scale, model

    def step_invalid(self):
        self.invalid += 1
        self.mu *= 10.0
        self.reuse = False
        return self.invalid >= 5

    def decide(self, cand_cost, model, step_norm_ambient, x_norm):
        """-> 'param_tol' | 'func_tol' | 'accept' | 'reject' for a valid step (Ceres' order of the tests)"""
        self.invalid = 0
        if step_norm_ambient <= 1e-8 * (x_norm + 1e-8):
            return "param_tol"
        change = self.cost - cand_cost
        if abs(change) <= 1e-6 * self.cost:
            return "func_tol"
        rho = change / model
        if rho > 1e-3:
            if rho < 0.25:
                self.radius *= 0.5
            if rho > 0.75:
                self.radius = max(self.radius, 3.0 * self.step_norm)
            self.mu = max(1e-8, 2.0 * self.mu / 10.0)
            self.reuse = False
            return "accept"
        self.radius *= 0.5
        self.reuse = True
        return "reject"
