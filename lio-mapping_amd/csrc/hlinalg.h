// hlinalg.h — host-side dense linear algebra for the small systems of the solve (D <= 246, n <= 120):
// Cholesky, Gauss-Jordan inverse, symmetric eigendecomposition (Householder tridiagonalisation +
// implicit-shift QL).  Row-major double.  These replace the Eigen calls at ImuFactor.h:74-75 and
// MarginalizationFactor.cc:276-302 and Ceres' dense solve (Estimator.cc:1911).
#pragma once
#include <immintrin.h>

#include <algorithm>
#include <cmath>
#include <vector>

namespace lio {

struct DMat {
  int r = 0, c = 0;
  std::vector<double> a;
  DMat() = default;
  DMat(int r_, int c_) : r(r_), c(c_), a(size_t(r_) * size_t(c_), 0.0) {}
  double &operator()(int i, int j) { return a[size_t(i) * c + j]; }
  double operator()(int i, int j) const { return a[size_t(i) * c + j]; }
  void zero() { std::fill(a.begin(), a.end(), 0.0); }
};

// In-place lower Cholesky of the leading n x n of A (row-major, stride lda).  false on a non-positive pivot.
inline bool chol_factor(double *A, int n, int lda) {
  for (int j = 0; j < n; ++j) {
    double d = A[j * lda + j];
    for (int k = 0; k < j; ++k) d -= A[j * lda + k] * A[j * lda + k];
    if (!(d > 0.0)) return false;
    double l = std::sqrt(d);
    A[j * lda + j] = l;
    for (int i = j + 1; i < n; ++i) {
      double s = A[i * lda + j];
      const double *ri = A + i * lda, *rj = A + j * lda;
      for (int k = 0; k < j; ++k) s -= ri[k] * rj[k];
      A[i * lda + j] = s / l;
    }
  }
  return true;
}
inline void chol_solve_inplace(const double *L, int n, int lda, double *b) {
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[i * lda + k] * b[k];
    b[i] = s / L[i * lda + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int k = i + 1; k < n; ++k) s -= L[k * lda + i] * b[k];
    b[i] = s / L[i * lda + i];
  }
}

// Right-looking upper Cholesky A = U^T U, in place in the upper triangle (row-major).  Every inner loop is
// an axpy over contiguous memory, which vectorises under strict IEEE semantics (the dot-product form of
// chol_factor needs reassociation to vectorise).  Used for the per-iteration D x D dogleg solve.
inline bool chol_upper_portable(double *A, int n, int lda) {
  // Panels of 4 pivot rows: the panel is factored with the plain recurrence, then every trailing row takes the
  // four rank-1 updates in one pass (5 loads + 1 store per 4 multiply-subtracts instead of 2 + 1 per one).
  // The subtractions stay in pivot order, so the result is bit-identical to the unblocked recurrence.
  for (int j0 = 0; j0 < n; j0 += 4) {
    const int jb = (n - j0 < 4) ? n - j0 : 4;
    for (int j = j0; j < j0 + jb; ++j) {
      double *rj = A + size_t(j) * lda;
      const double d = rj[j];
      if (!(d > 0.0)) return false;
      const double u = std::sqrt(d);
      rj[j] = u;
      const double iu = 1.0 / u;
      for (int i = j + 1; i < n; ++i) rj[i] *= iu;
      for (int k = j + 1; k < j0 + jb; ++k) {
        const double f = rj[k];
        double *rk = A + size_t(k) * lda;
        for (int i = k; i < n; ++i) rk[i] -= f * rj[i];
      }
    }
    if (jb < 4) {
      for (int j = j0; j < j0 + jb; ++j) {
        const double *rj = A + size_t(j) * lda;
        for (int k = j0 + jb; k < n; ++k) { const double f = rj[k]; double *rk = A + size_t(k) * lda; for (int i = k; i < n; ++i) rk[i] -= f * rj[i]; }
      }
      continue;
    }
    const double *__restrict__ r0 = A + size_t(j0) * lda;
    const double *__restrict__ r1 = r0 + lda;
    const double *__restrict__ r2 = r1 + lda;
    const double *__restrict__ r3 = r2 + lda;
    for (int k = j0 + 4; k < n; ++k) {
      const double f0 = r0[k], f1 = r1[k], f2 = r2[k], f3 = r3[k];
      double *__restrict__ rk = A + size_t(k) * lda;
      for (int i = k; i < n; ++i) {
        double v = rk[i];
        v -= f0 * r0[i];
        v -= f1 * r1[i];
        v -= f2 * r2[i];
        v -= f3 * r3[i];
        rk[i] = v;
      }
    }
  }
  return true;
}
// The same factorisation for hosts with AVX-512 FMA (EPYC Genoa / Turin, Xeon SPR: every MI355X host this was run on), selected at
// run time.  Blocked and UP-looking: per panel of 8 rows (1) the contribution of all rows above is subtracted by a register-tiled
// 8 x 24 kernel — 24 accumulators, per pivot row 3 vector loads + 8 broadcasts feed 24 FMAs, every panel element is loaded and
// stored once, every loop has a trip count that is a multiple of 8 —, (2) the 8 x 8 diagonal block is factored in scalars,
// (3) the rest of the panel is solved against it with the 8 row vectors of a column chunk in registers.  The right-looking form
// above re-reads and re-writes the trailing matrix once per 4 pivots with row loops of every length (D = 96 on the EPYC 9575F:
// 6.3 us against ~3 us for this form).  Out of place: A = chol(S + diag(shift)) with S untouched (S == A allowed), which
// also saves the dogleg its copy of H.  Subtractions are fused multiply-subtracts in ascending pivot order, so the factor differs
// from the portable one in the last bits only.  The strictly lower triangle of A's diagonal blocks is scratch.
// acc <- S[j0 + r][i ..] (zero if S is null);  acc -= sum_{k in [k0, k1)} U[k][j0 + r] * U[k][i ..];  A[j0 + r][i ..] <- acc
// (r = 0 .. 7, NV vectors of 8 columns from i, the last one masked).  Rows k whose eight multipliers are all zero are skipped:
// the speed-bias block of the window's normal matrix is block tridiagonal, its factor block bidiagonal.
template <int NV>
__attribute__((target("avx512f,fma"), always_inline)) inline void chol_blk_update(const double *S, const double *U, double *A, int lda, int j0, int i,
                                                                                  __mmask8 mlast, int k0, int k1) {
  __m512d acc[8][NV];
  for (int r = 0; r < 8; ++r)
    for (int c = 0; c < NV; ++c) {
      if (!S) { acc[r][c] = _mm512_setzero_pd(); continue; }
      const double *src = S + size_t(j0 + r) * lda + i + 8 * c;
      acc[r][c] = (c == NV - 1) ? _mm512_maskz_loadu_pd(mlast, src) : _mm512_loadu_pd(src);
    }
  const __m512d zero = _mm512_setzero_pd();
  for (int k = k0; k < k1; ++k) {
    const double *uk = U + size_t(k) * lda;
    if (_mm512_cmp_pd_mask(_mm512_loadu_pd(uk + j0), zero, _CMP_NEQ_UQ) == 0) continue;
    __m512d u[NV];
    for (int c = 0; c < NV; ++c) u[c] = (c == NV - 1) ? _mm512_maskz_loadu_pd(mlast, uk + i + 8 * c) : _mm512_loadu_pd(uk + i + 8 * c);
#pragma GCC unroll 8
    for (int r = 0; r < 8; ++r) {
      const __m512d b = _mm512_set1_pd(uk[j0 + r]);
      for (int c = 0; c < NV; ++c) acc[r][c] = _mm512_fnmadd_pd(b, u[c], acc[r][c]);
    }
  }
  for (int r = 0; r < 8; ++r)
    for (int c = 0; c < NV; ++c) {
      double *dst = A + size_t(j0 + r) * lda + i + 8 * c;
      if (c == NV - 1) _mm512_mask_storeu_pd(dst, mlast, acc[r][c]); else _mm512_storeu_pd(dst, acc[r][c]);
    }
}
// the 8-row band j0 of chol_blk_update over the columns [j0, ncols)
__attribute__((target("avx512f,fma"))) inline void chol_band_update(const double *S, const double *U, double *A, int lda, int j0, int ncols, int k0, int k1) {
  int i = j0;
  for (; i + 24 <= ncols; i += 24) chol_blk_update<3>(S, U, A, lda, j0, i, __mmask8(0xFF), k0, k1);
  const int rem = ncols - i;   // 0 .. 23
  if (rem > 16) chol_blk_update<3>(S, U, A, lda, j0, i, __mmask8((1u << (rem - 16)) - 1u), k0, k1);
  else if (rem > 8) chol_blk_update<2>(S, U, A, lda, j0, i, __mmask8((1u << (rem - 8)) - 1u), k0, k1);
  else if (rem > 0) chol_blk_update<1>(S, U, A, lda, j0, i, __mmask8((1u << rem) - 1u), k0, k1);
}
// Rows [j_begin, j_end) of the factorisation A = chol(S + diag(shift)) of an n x n matrix carried with ncols >= n columns (the
// columns from n on are right-hand sides: they come out as U^-T b).  j_begin a multiple of 8; j_end a multiple of 8 or n.  The
// rows above j_begin must already hold their part of the factor in A, and the rows from j_begin on of S must already carry the
// contribution of the rows [0, k_begin) (chol_gram_avx512) — k_begin = 0 for a factorisation in one go.  lda >= ncols rounded
// up to 8 is NOT required (all accesses are masked), lda >= ncols is.
__attribute__((target("avx512f,fma"))) inline bool chol_upper_panels_avx512(const double *S, double *A, int n, int ncols, int lda, const double *shift,
                                                                            int j_begin, int j_end, int k_begin) {
  int j0 = j_begin;
  for (; j0 + 8 <= j_end; j0 += 8) {
    chol_band_update(S, A, A, lda, j0, ncols, k_begin, j0);
    double d[8][8], inv[8];
    for (int r = 0; r < 8; ++r) for (int c = r; c < 8; ++c) d[r][c] = A[size_t(j0 + r) * lda + j0 + c];
    if (shift) for (int r = 0; r < 8; ++r) d[r][r] += shift[j0 + r];
    for (int j = 0; j < 8; ++j) {
      if (!(d[j][j] > 0.0)) return false;
      const double u = std::sqrt(d[j][j]);
      d[j][j] = u; inv[j] = 1.0 / u;
      for (int c = j + 1; c < 8; ++c) d[j][c] *= inv[j];
      for (int r = j + 1; r < 8; ++r) for (int c = r; c < 8; ++c) d[r][c] = std::fma(-d[j][r], d[j][c], d[r][c]);
    }
    for (int r = 0; r < 8; ++r) for (int c = r; c < 8; ++c) A[size_t(j0 + r) * lda + j0 + c] = d[r][c];
    for (int i = j0 + 8; i < ncols; i += 8) {
      const __mmask8 m = (ncols - i >= 8) ? __mmask8(0xFF) : __mmask8((1u << (ncols - i)) - 1u);
      __m512d a[8];
      for (int r = 0; r < 8; ++r) a[r] = _mm512_maskz_loadu_pd(m, A + size_t(j0 + r) * lda + i);
#pragma GCC unroll 8
      for (int j = 0; j < 8; ++j) {
        a[j] = _mm512_mul_pd(a[j], _mm512_set1_pd(inv[j]));
#pragma GCC unroll 8
        for (int r = j + 1; r < 8; ++r) a[r] = _mm512_fnmadd_pd(_mm512_set1_pd(d[j][r]), a[j], a[r]);
      }
      for (int r = 0; r < 8; ++r) _mm512_mask_storeu_pd(A + size_t(j0 + r) * lda + i, m, a[r]);
    }
  }
  if (j_end < n) return true;
  for (int j = j0; j < n; ++j) {   // the last n mod 8 rows: plain recurrence
    double *rj = A + size_t(j) * lda;
    const double *sj = S + size_t(j) * lda;
    for (int i = j; i < ncols; ++i) rj[i] = sj[i];
    if (shift) rj[j] += shift[j];
    for (int k = k_begin; k < j; ++k) { const double *rk = A + size_t(k) * lda; const double f = rk[j]; for (int i = j; i < ncols; ++i) rj[i] = std::fma(-f, rk[i], rj[i]); }
    if (!(rj[j] > 0.0)) return false;
    const double u = std::sqrt(rj[j]);
    rj[j] = u;
    const double iu = 1.0 / u;
    for (int i = j + 1; i < ncols; ++i) rj[i] *= iu;
  }
  return true;
}
// T[j][i] = - sum_{k in [k0, k1)} U[k][j] U[k][i] for the rows j in [j_begin, n) and the columns i from the start of j's 8-row band
// (the row itself for the last n mod 8 rows) to ncols: what the factor rows [k0, k1) take from the rows below them, formed ahead
// of time (SplitFactor: the speed-bias rows' contribution to the pose block while the lidar moments are still on their way).
__attribute__((target("avx512f,fma"))) inline void chol_gram_avx512(const double *U, double *T, int n, int ncols, int lda, int j_begin, int k0, int k1) {
  int j0 = j_begin;
  for (; j0 + 8 <= n; j0 += 8) chol_band_update(nullptr, U, T, lda, j0, ncols, k0, k1);
  for (int j = j0; j < n; ++j) {
    double *tj = T + size_t(j) * lda;
    for (int i = j; i < ncols; ++i) tj[i] = 0.0;
    for (int k = k0; k < k1; ++k) { const double *rk = U + size_t(k) * lda; const double f = rk[j]; if (f == 0.0) continue; for (int i = j; i < ncols; ++i) tj[i] = std::fma(-f, rk[i], tj[i]); }
  }
}
__attribute__((target("avx512f,fma"))) inline bool chol_upper_from_avx512(const double *S, double *A, int n, int lda, const double *shift) {
  return chol_upper_panels_avx512(S, A, n, n, lda, shift, 0, n, 0);
}
inline bool host_has_avx512() {
  static const bool has512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("fma");
  return has512;
}
// A = chol(S + diag(shift)): upper factor in A's upper triangle, S untouched unless S == A; shift may be null
inline bool chol_upper_from(const double *S, double *A, int n, int lda, const double *shift) {
  if (host_has_avx512()) return chol_upper_from_avx512(S, A, n, lda, shift);
  if (S != A) for (int i = 0; i < n; ++i) for (int j = i; j < n; ++j) A[size_t(i) * lda + j] = S[size_t(i) * lda + j];
  if (shift) for (int i = 0; i < n; ++i) A[size_t(i) * lda + i] += shift[i];
  return chol_upper_portable(A, n, lda);
}
inline bool chol_upper(double *A, int n, int lda) { return chol_upper_from(A, A, n, lda, nullptr); }

// x^T H x for a symmetric row-major H (the dogleg's |J g|^2 and s^T H s).  The plain double loop is a chain of n dependent adds
// per row under strict IEEE semantics (2.8 us at n = 96 on the build host); eight-lane partial sums, two accumulators per row.
// y (optional) receives H x.
inline double sym_quad_portable(const double *H, const double *x, int n, int lda, double *y) {
  double q = 0;
  for (int i = 0; i < n; ++i) {
    const double *row = H + size_t(i) * lda;
    double s = 0;
    for (int j = 0; j < n; ++j) s += row[j] * x[j];
    if (y) y[i] = s;
    q += x[i] * s;
  }
  return q;
}
__attribute__((target("avx512f,fma"))) inline double sym_quad_avx512(const double *H, const double *x, int n, int lda, double *y) {
  double q = 0;
  for (int i = 0; i < n; ++i) {
    const double *row = H + size_t(i) * lda;
    __m512d a0 = _mm512_setzero_pd(), a1 = _mm512_setzero_pd();
    int j = 0;
    for (; j + 16 <= n; j += 16) {
      a0 = _mm512_fmadd_pd(_mm512_loadu_pd(row + j), _mm512_loadu_pd(x + j), a0);
      a1 = _mm512_fmadd_pd(_mm512_loadu_pd(row + j + 8), _mm512_loadu_pd(x + j + 8), a1);
    }
    if (j + 8 <= n) { a0 = _mm512_fmadd_pd(_mm512_loadu_pd(row + j), _mm512_loadu_pd(x + j), a0); j += 8; }
    if (j < n) {
      const __mmask8 m = __mmask8((1u << (n - j)) - 1u);
      a1 = _mm512_fmadd_pd(_mm512_maskz_loadu_pd(m, row + j), _mm512_maskz_loadu_pd(m, x + j), a1);
    }
    const double s = _mm512_reduce_add_pd(_mm512_add_pd(a0, a1));
    if (y) y[i] = s;
    q += x[i] * s;
  }
  return q;
}
inline double sym_quad(const double *H, const double *x, int n, int lda, double *y = nullptr) {
  return host_has_avx512() ? sym_quad_avx512(H, x, n, lda, y) : sym_quad_portable(H, x, n, lda, y);
}
// y = y0 + A x for a row-major n x n A (the marginalization prior's r0 + J0 dx and J^T r0 + J^T J dx, twice per linearisation): a
// row's dot product is a chain of n dependent adds under strict IEEE semantics; eight-lane partial sums on hosts with AVX-512.
inline void affine_matvec_portable(const double *A, int n, int lda, const double *x, const double *y0, double *y) {
  for (int i = 0; i < n; ++i) { double s = y0[i]; const double *row = A + size_t(i) * lda; for (int j = 0; j < n; ++j) s += row[j] * x[j]; y[i] = s; }
}
__attribute__((target("avx512f,fma"))) inline void affine_matvec_avx512(const double *A, int n, int lda, const double *x, const double *y0, double *y) {
  const int nb = n & ~7;
  const __mmask8 m = __mmask8((1u << (n - nb)) - 1u);
  for (int i = 0; i < n; ++i) {
    const double *row = A + size_t(i) * lda;
    __m512d a0 = _mm512_setzero_pd(), a1 = _mm512_setzero_pd();
    int j = 0;
    for (; j + 16 <= nb; j += 16) {
      a0 = _mm512_fmadd_pd(_mm512_loadu_pd(row + j), _mm512_loadu_pd(x + j), a0);
      a1 = _mm512_fmadd_pd(_mm512_loadu_pd(row + j + 8), _mm512_loadu_pd(x + j + 8), a1);
    }
    if (j + 8 <= nb) { a0 = _mm512_fmadd_pd(_mm512_loadu_pd(row + j), _mm512_loadu_pd(x + j), a0); j += 8; }
    if (m) a1 = _mm512_fmadd_pd(_mm512_maskz_loadu_pd(m, row + j), _mm512_maskz_loadu_pd(m, x + j), a1);
    y[i] = y0[i] + _mm512_reduce_add_pd(_mm512_add_pd(a0, a1));
  }
}
inline void affine_matvec(const double *A, int n, int lda, const double *x, const double *y0, double *y) {
  if (host_has_avx512()) affine_matvec_avx512(A, n, lda, x, y0, y); else affine_matvec_portable(A, n, lda, x, y0, y);
}

// solve U^T U x = b in place
inline void chol_upper_solve_portable(const double *U, int n, int lda, double *b) {
  for (int i = 0; i < n; ++i) {  // forward: U^T y = b, column-oriented (axpy)
    const double y = b[i] / U[size_t(i) * lda + i];
    b[i] = y;
    const double *ri = U + size_t(i) * lda;
    for (int k = i + 1; k < n; ++k) b[k] -= ri[k] * y;
  }
  for (int i = n - 1; i >= 0; --i) {  // backward: U x = y
    const double *ri = U + size_t(i) * lda;
    double s = b[i];
    for (int k = i + 1; k < n; ++k) s -= ri[k] * b[k];
    b[i] = s / ri[i];
  }
}

// Both sweeps are latency chains when done element by element: forward, b[i + 1] waits for the axpy of step i (divide + FMA +
// store-to-load forwarding per step); backward, a dot product and a divide per row.  Blocked by 8: the 8 x 8 triangle of a block
// is solved in scalars with reciprocals of the diagonal (computed once, eight at a time), everything outside the triangle is
// eight independent vector FMAs per column chunk (forward) or eight independent dot products (backward).
__attribute__((target("avx512f,fma"))) inline void chol_upper_solve_avx512(const double *U, int n, int lda, double *b) {
  double inv[256];
  std::vector<double> inv_big;
  double *iv = inv;
  if (n > 256) { inv_big.resize(n); iv = inv_big.data(); }
  for (int i = 0; i < n; ++i) iv[i] = U[size_t(i) * lda + i];
  {
    const __m512d one = _mm512_set1_pd(1.0);
    int i = 0;
    for (; i + 8 <= n; i += 8) _mm512_storeu_pd(iv + i, _mm512_div_pd(one, _mm512_loadu_pd(iv + i)));
    for (; i < n; ++i) iv[i] = 1.0 / iv[i];
  }
  const int nb = n & ~7;
  // forward: U^T y = b
  for (int i0 = 0; i0 < nb; i0 += 8) {
    double y[8];
    for (int j = 0; j < 8; ++j) y[j] = b[i0 + j];
    for (int j = 0; j < 8; ++j) {
      y[j] *= iv[i0 + j];
      const double *rj = U + size_t(i0 + j) * lda + i0;
      for (int r = j + 1; r < 8; ++r) y[r] = std::fma(-rj[r], y[j], y[r]);
    }
    for (int j = 0; j < 8; ++j) b[i0 + j] = y[j];
    __m512d Y[8];
    for (int j = 0; j < 8; ++j) Y[j] = _mm512_set1_pd(y[j]);
    for (int k = i0 + 8; k < n; k += 8) {
      const __mmask8 m = (n - k >= 8) ? __mmask8(0xFF) : __mmask8((1u << (n - k)) - 1u);
      __m512d v = _mm512_maskz_loadu_pd(m, b + k);
#pragma GCC unroll 8
      for (int j = 0; j < 8; ++j) v = _mm512_fnmadd_pd(_mm512_maskz_loadu_pd(m, U + size_t(i0 + j) * lda + k), Y[j], v);
      _mm512_mask_storeu_pd(b + k, m, v);
    }
  }
  for (int i = nb; i < n; ++i) {   // the last n mod 8 rows
    const double yv = b[i] * iv[i];
    b[i] = yv;
    const double *ri = U + size_t(i) * lda;
    for (int k = i + 1; k < n; ++k) b[k] = std::fma(-ri[k], yv, b[k]);
  }
  // backward: U x = y
  for (int i = n - 1; i >= nb; --i) {
    const double *ri = U + size_t(i) * lda;
    double sres = b[i];
    for (int k = i + 1; k < n; ++k) sres = std::fma(-ri[k], b[k], sres);
    b[i] = sres * iv[i];
  }
  for (int i0 = nb - 8; i0 >= 0; i0 -= 8) {
    __m512d acc[8];
    for (int r = 0; r < 8; ++r) acc[r] = _mm512_setzero_pd();
    for (int k = i0 + 8; k < n; k += 8) {
      const __mmask8 m = (n - k >= 8) ? __mmask8(0xFF) : __mmask8((1u << (n - k)) - 1u);
      const __m512d xv = _mm512_maskz_loadu_pd(m, b + k);
#pragma GCC unroll 8
      for (int r = 0; r < 8; ++r) acc[r] = _mm512_fmadd_pd(_mm512_maskz_loadu_pd(m, U + size_t(i0 + r) * lda + k), xv, acc[r]);
    }
    double x[8];
    for (int r = 0; r < 8; ++r) x[r] = b[i0 + r] - _mm512_reduce_add_pd(acc[r]);
    for (int j = 7; j >= 0; --j) {
      const double *rj = U + size_t(i0 + j) * lda + i0;
      double sres = x[j];
      for (int r = j + 1; r < 8; ++r) sres = std::fma(-rj[r], x[r], sres);
      x[j] = sres * iv[i0 + j];
    }
    for (int r = 0; r < 8; ++r) b[i0 + r] = x[r];
  }
}
// x <- U^-1 x (the backward sweep of chol_upper_solve_avx512 on its own: SplitFactor gets U^-T b from the factorisation itself)
__attribute__((target("avx512f,fma"))) inline void upper_backsolve_avx512(const double *U, int n, int lda, double *b) {
  double inv[256];
  std::vector<double> inv_big;
  double *iv = inv;
  if (n > 256) { inv_big.resize(n); iv = inv_big.data(); }
  for (int i = 0; i < n; ++i) iv[i] = 1.0 / U[size_t(i) * lda + i];
  const int nb = n & ~7;
  for (int i = n - 1; i >= nb; --i) {
    const double *ri = U + size_t(i) * lda;
    double sres = b[i];
    for (int k = i + 1; k < n; ++k) sres = std::fma(-ri[k], b[k], sres);
    b[i] = sres * iv[i];
  }
  for (int i0 = nb - 8; i0 >= 0; i0 -= 8) {
    __m512d acc[8];
    for (int r = 0; r < 8; ++r) acc[r] = _mm512_setzero_pd();
    for (int k = i0 + 8; k < n; k += 8) {
      const __mmask8 m = (n - k >= 8) ? __mmask8(0xFF) : __mmask8((1u << (n - k)) - 1u);
      const __m512d xv = _mm512_maskz_loadu_pd(m, b + k);
#pragma GCC unroll 8
      for (int r = 0; r < 8; ++r) acc[r] = _mm512_fmadd_pd(_mm512_maskz_loadu_pd(m, U + size_t(i0 + r) * lda + k), xv, acc[r]);
    }
    double x[8];
    for (int r = 0; r < 8; ++r) x[r] = b[i0 + r] - _mm512_reduce_add_pd(acc[r]);
    for (int j = 7; j >= 0; --j) {
      const double *rj = U + size_t(i0 + j) * lda + i0;
      double sres = x[j];
      for (int r = j + 1; r < 8; ++r) sres = std::fma(-rj[r], x[r], sres);
      x[j] = sres * iv[i0 + j];
    }
    for (int r = 0; r < 8; ++r) b[i0 + r] = x[r];
  }
}
inline void chol_upper_solve(const double *U, int n, int lda, double *b) {
  if (host_has_avx512()) chol_upper_solve_avx512(U, n, lda, b); else chol_upper_solve_portable(U, n, lda, b);
}

// Gauss-Jordan inverse with partial pivoting, n x n.
inline bool gj_inverse(const double *Ain, int n, double *Ainv) {
  std::vector<double> M(Ain, Ain + size_t(n) * n);
  for (int i = 0; i < n * n; ++i) Ainv[i] = 0;
  for (int i = 0; i < n; ++i) Ainv[i * n + i] = 1;
  for (int col = 0; col < n; ++col) {
    int piv = col;
    double best = std::fabs(M[col * n + col]);
    for (int i = col + 1; i < n; ++i)
      if (std::fabs(M[i * n + col]) > best) { best = std::fabs(M[i * n + col]); piv = i; }
    if (best == 0.0) return false;
    if (piv != col)
      for (int j = 0; j < n; ++j) { std::swap(M[piv * n + j], M[col * n + j]); std::swap(Ainv[piv * n + j], Ainv[col * n + j]); }
    double d = M[col * n + col];
    for (int j = 0; j < n; ++j) { M[col * n + j] /= d; Ainv[col * n + j] /= d; }
    for (int i = 0; i < n; ++i) {
      if (i == col) continue;
      double f = M[i * n + col];
      if (f == 0.0) continue;
      for (int j = 0; j < n; ++j) { M[i * n + j] -= f * M[col * n + j]; Ainv[i * n + j] -= f * Ainv[col * n + j]; }
    }
  }
  return true;
}

// Symmetric eigendecomposition: A (n x n, symmetric) -> eigenvalues ascending in w, eigenvectors as the
// COLUMNS of V (row-major n x n).  Householder reduction to tridiagonal form followed by the implicit
// QL algorithm (the classical tred2/tql2 pair).
inline bool sym_eig(const double *A, int n, double *w, double *V) {
  // The classical formulation walks COLUMNS of a row-major V in every inner loop (stride n: no vector loads).  All of it
  // runs here on T = V^T instead — same operations in the same order, contiguous inner loops — and V is T^T at the end.
  std::vector<double> e(n, 0.0), Tbuf(size_t(n) * n);
  double *T = Tbuf.data();
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) T[j * n + i] = A[i * n + j];
  double *d = w;
  // --- tred2
  for (int j = 0; j < n; ++j) d[j] = T[(j) * n + (n - 1)];
  for (int i = n - 1; i > 0; --i) {
    double scale = 0.0, h = 0.0;
    for (int k = 0; k < i; ++k) scale += std::fabs(d[k]);
    if (scale == 0.0) {
      e[i] = d[i - 1];
      for (int j = 0; j < i; ++j) { d[j] = T[(j) * n + (i - 1)]; T[(j) * n + i] = 0.0; T[(i) * n + j] = 0.0; }
    } else {
      for (int k = 0; k < i; ++k) { d[k] /= scale; h += d[k] * d[k]; }
      double f = d[i - 1];
      double g = std::sqrt(h);
      if (f > 0) g = -g;
      e[i] = scale * g;
      h -= f * g;
      d[i - 1] = f - g;
      for (int j = 0; j < i; ++j) e[j] = 0.0;
      for (int j = 0; j < i; ++j) {
        f = d[j];
        T[(i) * n + j] = f;
        g = e[j] + T[(j) * n + j] * f;
        for (int k = j + 1; k <= i - 1; ++k) { g += T[(j) * n + k] * d[k]; e[k] += T[(j) * n + k] * f; }
        e[j] = g;
      }
      f = 0.0;
      for (int j = 0; j < i; ++j) { e[j] /= h; f += e[j] * d[j]; }
      double hh = f / (h + h);
      for (int j = 0; j < i; ++j) e[j] -= hh * d[j];
      for (int j = 0; j < i; ++j) {
        f = d[j]; g = e[j];
        { double *__restrict tj = T + size_t(j) * n; for (int k = j; k <= i - 1; ++k) tj[k] -= (f * e[k] + g * d[k]); }
        d[j] = T[(j) * n + (i - 1)];
        T[(j) * n + i] = 0.0;
      }
    }
    d[i] = h;
  }
  for (int i = 0; i < n - 1; ++i) {
    T[(i) * n + (n - 1)] = T[(i) * n + i];
    T[(i) * n + i] = 1.0;
    double h = d[i + 1];
    if (h != 0.0) {
      for (int k = 0; k <= i; ++k) d[k] = T[(i + 1) * n + k] / h;
      for (int j = 0; j <= i; ++j) {
        double g = 0.0;
        for (int k = 0; k <= i; ++k) g += T[(i + 1) * n + k] * T[(j) * n + k];
        { double *__restrict tj = T + size_t(j) * n; for (int k = 0; k <= i; ++k) tj[k] -= g * d[k]; }
      }
    }
    for (int k = 0; k <= i; ++k) T[(i + 1) * n + k] = 0.0;
  }
  for (int j = 0; j < n; ++j) { d[j] = T[(j) * n + (n - 1)]; T[(j) * n + (n - 1)] = 0.0; }
  T[(n - 1) * n + (n - 1)] = 1.0;
  e[0] = 0.0;
  // --- tql2
  for (int i = 1; i < n; ++i) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  double f = 0.0, tst1 = 0.0;
  const double eps = std::pow(2.0, -52.0);
  for (int l = 0; l < n; ++l) {
    tst1 = std::max(tst1, std::fabs(d[l]) + std::fabs(e[l]));
    int m = l;
    while (m < n) {
      if (std::fabs(e[m]) <= eps * tst1) break;
      ++m;
    }
    if (m > l) {
      int iter = 0;
      do {
        if (++iter > 200) return false;
        double g = d[l];
        double p = (d[l + 1] - g) / (2.0 * e[l]);
        double r = std::hypot(p, 1.0);
        if (p < 0) r = -r;
        d[l] = e[l] / (p + r);
        d[l + 1] = e[l] * (p + r);
        double dl1 = d[l + 1];
        double h = g - d[l];
        for (int i = l + 2; i < n; ++i) d[i] -= h;
        f += h;
        p = d[m];
        double c = 1.0, c2 = c, c3 = c, el1 = e[l + 1], s = 0.0, s2 = 0.0;
        for (int i = m - 1; i >= l; --i) {
          c3 = c2; c2 = c; s2 = s;
          g = c * e[i];
          h = c * p;
          r = std::hypot(p, e[i]);
          e[i + 1] = s * r;
          s = e[i] / r;
          c = p / r;
          p = c * d[i] - s * g;
          d[i + 1] = h + s * (c * g + s * d[i]);
          double *__restrict t1 = T + size_t(i + 1) * n, *__restrict t0 = T + size_t(i) * n;  // two distinct rows: vectorises
          for (int k = 0; k < n; ++k) {
            const double hk = t1[k], vk = t0[k];
            t1[k] = s * vk + c * hk;
            t0[k] = c * vk - s * hk;
          }
        }
        p = -s * s2 * c3 * el1 * e[l] / dl1;
        e[l] = s * p;
        d[l] = c * p;
      } while (std::fabs(e[l]) > eps * tst1);
    }
    d[l] = d[l] + f;
    e[l] = 0.0;
  }
  // sort ascending
  for (int i = 0; i < n - 1; ++i) {
    int k = i;
    double p = d[i];
    for (int j = i + 1; j < n; ++j)
      if (d[j] < p) { k = j; p = d[j]; }
    if (k != i) {
      d[k] = d[i]; d[i] = p;
      for (int j = 0; j < n; ++j) std::swap(T[(i) * n + j], T[(k) * n + j]);
    }
  }
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[i * n + j] = T[j * n + i];
  return true;
}

}  // namespace lio
