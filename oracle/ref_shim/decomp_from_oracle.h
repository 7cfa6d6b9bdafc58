// The dense decompositions behind oracle/ref_shim/Eigen/Eigen's ColPivHouseholderQR / SelfAdjointEigenSolver facades, forwarded to
// the ORACLE's restatements (oracle/liomath.h).  Whatever goes through them is NOT an independent check of those two algorithms —
// the reference's own logic around them (what is solved, with which rows, what happens to the solution) is what gets exercised.
#pragma once
#include <vector>

#include "../imu_init.h"
#include "../liomath.h"
#include "Eigen/Eigen"
namespace Eigen {
template <typename T> void shim_colpiv_qr_solve(int m, int n, const T *A, const T *b, T *x) {
  std::vector<T> a(A, A + size_t(m) * n), bb(b, b + m);
  orc::colpiv_qr_solve<T>(m, n, a.data(), bb.data(), x);
}
template <typename T> void shim_sym_eigen(int n, const T *A, T *w, T *V) { orc::sym_eigen<T>(n, A, w, V); }
// A.ldlt().solve(b): the oracle restates it as Gaussian elimination with partial pivoting (oracle/imu_init.h: DenseSolve)
template <typename T> void shim_ldlt_solve(int n, const T *A, const T *b, T *x) {
  orc::Mat M(n, n);
  std::vector<double> bb(n);
  for (int i = 0; i < n; ++i) { bb[i] = double(b[i]); for (int k = 0; k < n; ++k) M(i, k) = double(A[size_t(i) * n + k]); }
  const std::vector<double> r = orc::DenseSolve(M, bb);
  for (int i = 0; i < n; ++i) x[i] = T(r[i]);
}
}  // namespace Eigen
