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
// Same factorisation with AVX-512 FMA for the trailing update (two target rows per pass so every pivot-row load feeds two
// FMAs).  Selected at run time on hosts that have it (EPYC Genoa/Turin, Xeon SPR: every MI355X host this was run on);
// the fused multiply-subtract rounds once, so the factor differs from the portable one in the last bits only.
__attribute__((target("avx512f,fma"))) inline bool chol_upper_avx512(double *A, int n, int lda) {
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
    const double *r0 = A + size_t(j0) * lda, *r1 = r0 + lda, *r2 = r1 + lda, *r3 = r2 + lda;
    int k = j0 + 4;
    for (; k + 1 < n; k += 2) {
      double *ra = A + size_t(k) * lda, *rb = ra + lda;
      const double a0 = r0[k], a1 = r1[k], a2 = r2[k], a3 = r3[k];
      const double b0 = r0[k + 1], b1 = r1[k + 1], b2 = r2[k + 1], b3 = r3[k + 1];
      ra[k] = ra[k] - a0 * r0[k] - a1 * r1[k] - a2 * r2[k] - a3 * r3[k];  // the element row b does not have
      const __m512d A0 = _mm512_set1_pd(a0), A1 = _mm512_set1_pd(a1), A2 = _mm512_set1_pd(a2), A3 = _mm512_set1_pd(a3);
      const __m512d B0 = _mm512_set1_pd(b0), B1 = _mm512_set1_pd(b1), B2 = _mm512_set1_pd(b2), B3 = _mm512_set1_pd(b3);
      int i = k + 1;
      for (; i + 8 <= n; i += 8) {
        const __m512d p0 = _mm512_loadu_pd(r0 + i), p1 = _mm512_loadu_pd(r1 + i), p2 = _mm512_loadu_pd(r2 + i), p3 = _mm512_loadu_pd(r3 + i);
        __m512d va = _mm512_loadu_pd(ra + i), vb = _mm512_loadu_pd(rb + i);
        va = _mm512_fnmadd_pd(A0, p0, va); vb = _mm512_fnmadd_pd(B0, p0, vb);
        va = _mm512_fnmadd_pd(A1, p1, va); vb = _mm512_fnmadd_pd(B1, p1, vb);
        va = _mm512_fnmadd_pd(A2, p2, va); vb = _mm512_fnmadd_pd(B2, p2, vb);
        va = _mm512_fnmadd_pd(A3, p3, va); vb = _mm512_fnmadd_pd(B3, p3, vb);
        _mm512_storeu_pd(ra + i, va); _mm512_storeu_pd(rb + i, vb);
      }
      if (i < n) {
        const __mmask8 m = __mmask8((1u << (n - i)) - 1u);
        const __m512d p0 = _mm512_maskz_loadu_pd(m, r0 + i), p1 = _mm512_maskz_loadu_pd(m, r1 + i), p2 = _mm512_maskz_loadu_pd(m, r2 + i),
                      p3 = _mm512_maskz_loadu_pd(m, r3 + i);
        __m512d va = _mm512_maskz_loadu_pd(m, ra + i), vb = _mm512_maskz_loadu_pd(m, rb + i);
        va = _mm512_fnmadd_pd(A0, p0, va); vb = _mm512_fnmadd_pd(B0, p0, vb);
        va = _mm512_fnmadd_pd(A1, p1, va); vb = _mm512_fnmadd_pd(B1, p1, vb);
        va = _mm512_fnmadd_pd(A2, p2, va); vb = _mm512_fnmadd_pd(B2, p2, vb);
        va = _mm512_fnmadd_pd(A3, p3, va); vb = _mm512_fnmadd_pd(B3, p3, vb);
        _mm512_mask_storeu_pd(ra + i, m, va); _mm512_mask_storeu_pd(rb + i, m, vb);
      }
    }
    for (; k < n; ++k) {
      const double f0 = r0[k], f1 = r1[k], f2 = r2[k], f3 = r3[k];
      double *rk = A + size_t(k) * lda;
      for (int i = k; i < n; ++i) rk[i] = rk[i] - f0 * r0[i] - f1 * r1[i] - f2 * r2[i] - f3 * r3[i];
    }
  }
  return true;
}
inline bool chol_upper(double *A, int n, int lda) {
  static const bool has512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("fma");
  return has512 ? chol_upper_avx512(A, n, lda) : chol_upper_portable(A, n, lda);
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

// The backward sweep is a chain of dot products: one scalar accumulator makes it latency-bound (n^2/2 dependent
// multiply-adds, 6 us at n = 96).  Eight-lane partial sums cut the chain eight-fold.
__attribute__((target("avx512f,fma"))) inline void chol_upper_solve_avx512(const double *U, int n, int lda, double *b) {
  for (int i = 0; i < n; ++i) {  // forward: U^T y = b, column-oriented (axpy)
    const double y = b[i] / U[size_t(i) * lda + i];
    b[i] = y;
    const double *ri = U + size_t(i) * lda;
    const __m512d Y = _mm512_set1_pd(y);
    int k = i + 1;
    for (; k + 8 <= n; k += 8) _mm512_storeu_pd(b + k, _mm512_fnmadd_pd(_mm512_loadu_pd(ri + k), Y, _mm512_loadu_pd(b + k)));
    if (k < n) {
      const __mmask8 m = __mmask8((1u << (n - k)) - 1u);
      _mm512_mask_storeu_pd(b + k, m, _mm512_fnmadd_pd(_mm512_maskz_loadu_pd(m, ri + k), Y, _mm512_maskz_loadu_pd(m, b + k)));
    }
  }
  for (int i = n - 1; i >= 0; --i) {  // backward: U x = y
    const double *ri = U + size_t(i) * lda;
    __m512d acc = _mm512_setzero_pd();
    int k = i + 1;
    for (; k + 8 <= n; k += 8) acc = _mm512_fmadd_pd(_mm512_loadu_pd(ri + k), _mm512_loadu_pd(b + k), acc);
    if (k < n) {
      const __mmask8 m = __mmask8((1u << (n - k)) - 1u);
      acc = _mm512_fmadd_pd(_mm512_maskz_loadu_pd(m, ri + k), _mm512_maskz_loadu_pd(m, b + k), acc);
    }
    b[i] = (b[i] - _mm512_reduce_add_pd(acc)) / ri[i];
  }
}
inline void chol_upper_solve(const double *U, int n, int lda, double *b) {
  static const bool has512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("fma");
  if (has512) chol_upper_solve_avx512(U, n, lda, b); else chol_upper_solve_portable(U, n, lda, b);
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
