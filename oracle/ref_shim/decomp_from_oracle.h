// The dense decompositions behind oracle/ref_shim/Eigen/Eigen's ColPivHouseholderQR / SelfAdjointEigenSolver facades, forwarded to
// the ORACLE's restatements (oracle/liomath.h).  Whatever goes through them is NOT an independent check of those two algorithms —
// the reference's own logic around them (what is solved, with which rows, what happens to the solution) is what gets exercised.
#pragma once
#include <vector>

#include "../liomath.h"
#include "Eigen/Eigen"
namespace Eigen {
template <typename T> void shim_colpiv_qr_solve(int m, int n, const T *A, const T *b, T *x) {
  std::vector<T> a(A, A + size_t(m) * n), bb(b, b + m);
  orc::colpiv_qr_solve<T>(m, n, a.data(), bb.data(), x);
}
template <typename T> void shim_sym_eigen(int n, const T *A, T *w, T *V) { orc::sym_eigen<T>(n, A, w, V); }
// (A.ldlt().solve(b) for the small symmetric systems of the IMU initialiser: a pivot-free L D L^T, the textbook recurrence)
template <typename T> void shim_ldlt_solve(int n, const T *A, const T *b, T *x) {
  std::vector<T> L(size_t(n) * n, T(0)), D(n), y(n);
  for (int j = 0; j < n; ++j) {
    T d = A[size_t(j) * n + j];
    for (int k = 0; k < j; ++k) d -= L[size_t(j) * n + k] * L[size_t(j) * n + k] * D[k];
    D[j] = d; L[size_t(j) * n + j] = T(1);
    for (int i = j + 1; i < n; ++i) { T s = A[size_t(i) * n + j]; for (int k = 0; k < j; ++k) s -= L[size_t(i) * n + k] * L[size_t(j) * n + k] * D[k]; L[size_t(i) * n + j] = s / d; }
  }
  for (int i = 0; i < n; ++i) { T s = b[i]; for (int k = 0; k < i; ++k) s -= L[size_t(i) * n + k] * y[k]; y[i] = s; }
  for (int i = 0; i < n; ++i) y[i] /= D[i];
  for (int i = n - 1; i >= 0; --i) { T s = y[i]; for (int k = i + 1; k < n; ++k) s -= L[size_t(k) * n + i] * x[k]; x[i] = s; }
}
}  // namespace Eigen
