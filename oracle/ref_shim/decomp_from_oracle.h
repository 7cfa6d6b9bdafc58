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
}  // namespace Eigen
