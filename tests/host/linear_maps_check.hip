// Stand-alone check (host code only, built with hipcc by tests/test_host_linalg.py): the 18x13 linear map L and the 13-vector l
// that host_solver.h builds per (pivot, frame i, extrinsic) triple must reproduce PivotPointPlaneFactor's residual and Jacobians
// (ppp_factor, host_factors.h) for arbitrary points and plane coefficients: j = L z, r = l^T z, z = [w (x) (p, 1); d].
#include <cstdio>
#include <random>
#include "host_solver.h"
using namespace lio;
int main() {
  std::mt19937 rng(3); std::normal_distribution<double> N(0, 1);
  double worst = 0, worst_res = 0;
  for (int trial = 0; trial < 200; ++trial) {
    double pose[3][7];
    for (auto &p : pose) { for (int k = 0; k < 3; ++k) p[k] = 3 * N(rng); double q[4], s = 0; for (auto &v : q) { v = N(rng); s += v * v; } s = std::sqrt(s); for (int k = 0; k < 4; ++k) p[3 + k] = q[k] / s; }
    double L[18 * 13], l[13];
    lidar_linear_maps(pose[0], pose[1], pose[2], L, l);
    // random residual: j = L z must equal the factor's Jacobian
    V3d p(10 * N(rng), 10 * N(rng), 3 * N(rng));
    double coeff[4] = {N(rng), N(rng), N(rng), N(rng)};
    double res, Jp[7], Ji[7], Jx[7];
    ppp_factor(p, coeff, pose[0], pose[1], pose[2], &res, Jp, Ji, Jx);
    double z[13];
    for (int a = 0; a < 3; ++a) { z[4 * a] = coeff[a] * p.x; z[4 * a + 1] = coeff[a] * p.y; z[4 * a + 2] = coeff[a] * p.z; z[4 * a + 3] = coeff[a]; }
    z[12] = coeff[3];
    for (int k = 0; k < 18; ++k) { double s = 0; for (int c = 0; c < 13; ++c) s += L[k * 13 + c] * z[c]; double ref = k < 6 ? Jp[k] : (k < 12 ? Ji[k - 6] : Jx[k - 12]); worst = std::max(worst, std::fabs(s - ref) / (1 + std::fabs(ref))); }
    double s = 0; for (int c = 0; c < 13; ++c) s += l[c] * z[c];
    worst_res = std::max(worst_res, std::fabs(s - res) / (1 + std::fabs(res)));
  }
  std::printf("%.3e %.3e\n", worst, worst_res);
  return 0;
}
