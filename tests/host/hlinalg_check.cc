// Stand-alone check of the product's host linear algebra (lio-mapping_amd/csrc/hlinalg.h): reads problems from stdin,
// prints solutions; tests/test_host_linalg.py compares them with numpy.  Built with g++ by the test (no GPU, no HIP).
#include <cstdio>
#include <vector>

#include "hlinalg.h"

int main() {
  int n;
  while (std::scanf("%d", &n) == 1) {
    std::vector<double> H(size_t(n) * n), g(n), A, x, w(n), V(size_t(n) * n), Ap;
    for (double &v : H) if (std::scanf("%lf", &v) != 1) return 1;
    for (double &v : g) if (std::scanf("%lf", &v) != 1) return 1;
    A = H; Ap = H;
    const bool ok = lio::chol_upper(A.data(), n, n), okp = lio::chol_upper_portable(Ap.data(), n, n);
    x = g;
    if (ok) lio::chol_upper_solve(A.data(), n, n, x.data());
    std::vector<double> xp = g;
    if (okp) lio::chol_upper_solve_portable(Ap.data(), n, n, xp.data());
    const bool eok = lio::sym_eig(H.data(), n, w.data(), V.data());
    // out-of-place factorisation of H + diag(shift) (the dogleg's mu-regularised system; H must stay untouched), and x^T H x
    std::vector<double> shift(n), As(size_t(n) * n, 0.0), Hkeep(H), xs(g), Hg(n), Hgp(n);
    for (int i = 0; i < n; ++i) shift[i] = 0.25 * H[size_t(i) * n + i] + 1e-3;
    const bool oks = lio::chol_upper_from(H.data(), As.data(), n, n, shift.data()) && H == Hkeep;
    if (oks) lio::chol_upper_solve(As.data(), n, n, xs.data());
    const double q = lio::sym_quad(H.data(), g.data(), n, n, Hg.data()), qp = lio::sym_quad_portable(H.data(), g.data(), n, n, Hgp.data());
    std::printf("%d %d %d %d\n", int(ok), int(okp), int(eok), int(oks));
    for (int i = 0; i < n; ++i) std::printf("%.17g ", x[i]);
    std::printf("\n");
    for (int i = 0; i < n; ++i) std::printf("%.17g ", xp[i]);
    std::printf("\n");
    for (int i = 0; i < n; ++i) std::printf("%.17g ", w[i]);
    std::printf("\n");
    for (size_t i = 0; i < V.size(); ++i) std::printf("%.17g ", V[i]);
    std::printf("\n");
    for (int i = 0; i < n; ++i) std::printf("%.17g ", xs[i]);
    std::printf("\n");
    std::printf("%.17g %.17g ", q, qp);
    for (int i = 0; i < n; ++i) std::printf("%.17g %.17g ", Hg[i], Hgp[i]);
    std::printf("\n");
  }
  return 0;
}
