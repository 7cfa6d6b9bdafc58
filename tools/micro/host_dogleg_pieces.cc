// Host pieces of one dogleg iteration at D = 96 (lio-mapping_amd/csrc/hlinalg.h): out-of-place blocked Cholesky, the portable
// right-looking form, the triangular solves, the quadratic form — best of 400 x 50 repetitions on the host it runs on.
// g++ -O3 -std=c++17 -ffp-contract=off -mavx2 -I lio-mapping_amd/csrc tools/micro/host_dogleg_pieces.cc -o /tmp/hdp && /tmp/hdp
#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "hlinalg.h"
using namespace lio;
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const int n = 96;
  std::mt19937 rng(1); std::normal_distribution<double> N(0, 1);
  std::vector<double> J(size_t(200) * n), H(size_t(n) * n), A(size_t(n) * n), g(n), x(n), shift(n, 1e-6);
  for (auto &v : J) v = N(rng);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < 200; ++k) s += J[k * n + i] * J[k * n + j]; H[i * n + j] = s; }
  for (int i = 0; i < n; ++i) g[i] = N(rng);
  double sink = 0, bf = 1e9, bp = 1e9, bs = 1e9, bq = 1e9, bqp = 1e9;
  for (int rep = 0; rep < 400; ++rep) {
    double t0 = now_us();
    for (int r = 0; r < 50; ++r) { chol_upper_from(H.data(), A.data(), n, n, shift.data()); sink += A[7]; }
    double t1 = now_us();
    for (int r = 0; r < 50; ++r) { std::memcpy(A.data(), H.data(), sizeof(double) * n * n); chol_upper_portable(A.data(), n, n); sink += A[7]; }
    double t2 = now_us();
    for (int r = 0; r < 50; ++r) { x = g; chol_upper_solve(A.data(), n, n, x.data()); sink += x[3]; }
    double t3 = now_us();
    for (int r = 0; r < 50; ++r) { g[0] += 1e-12; sink += sym_quad(H.data(), g.data(), n, n); }
    double t4 = now_us();
    for (int r = 0; r < 50; ++r) { g[0] += 1e-12; sink += sym_quad_portable(H.data(), g.data(), n, n, nullptr); }
    double t5 = now_us();
    bf = std::min(bf, (t1 - t0) / 50); bp = std::min(bp, (t2 - t1) / 50); bs = std::min(bs, (t3 - t2) / 50); bq = std::min(bq, (t4 - t3) / 50); bqp = std::min(bqp, (t5 - t4) / 50);
  }
  std::printf("n 96 best of 400x50: chol_upper_from %.3f | copy + portable %.3f | solve %.3f | sym_quad %.3f | portable quad %.3f us  (%g)\n", bf, bp, bs, bq, bqp, sink);
}
