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
    std::printf("%d %d %d\n", int(ok), int(okp), int(eok));
    for (int i = 0; i < n; ++i) std::printf("%.17g ", x[i]);
    std::printf("\n");
    for (int i = 0; i < n; ++i) std::printf("%.17g ", xp[i]);
    std::printf("\n");
    for (int i = 0; i < n; ++i) std::printf("%.17g ", w[i]);
    std::printf("\n");
    for (size_t i = 0; i < V.size(); ++i) std::printf("%.17g ", V[i]);
    std::printf("\n");
  }
  return 0;
}
