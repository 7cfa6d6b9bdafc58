// Stand-alone check of the product's degeneracy count (lio-mapping_amd/csrc/hmath.h: count_eigs_below<6>, the number of
// eigenvalues of the 6x6 AtA below a threshold by Sylvester's inertia).  Reads 6x6 float matrices + a threshold from
// stdin, prints the count; tests/test_host_linalg.py compares with numpy's eigenvalues.  Built with g++ by the test (no GPU).
#include <cstdio>

#include "hmath.h"

int main() {
  float A[36];
  double tau;
  for (;;) {
    for (int i = 0; i < 36; ++i) if (std::scanf("%f", &A[i]) != 1) return 0;
    if (std::scanf("%lf", &tau) != 1) return 0;
    std::printf("%d\n", lio::count_eigs_below<6>(A, tau));
  }
}
