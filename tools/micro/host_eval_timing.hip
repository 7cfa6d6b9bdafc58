// Host share of the dogleg loop on a sliding-window problem of tests/golden/ref_solve_problems.npz (host code only; the problem file
// is what tests/test_ref_solve_problem.py::pack writes).  The lidar moments of every linearisation are recorded once (CPU sums) and
// then REPLAYED through the split callbacks (lidar_launch = nothing, lidar_wait_frame = a copy), so that a timed solve_dogleg contains
// exactly the host's own work of a solve: prior + IMU factors + linear maps (hidden under the device pass in the product), frame-block
// assembly, factorisation, step.  Prints per-pass figures.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mavx2 -I lio-mapping_amd/csrc tools/micro/host_eval_timing.hip -o /tmp/het && /tmp/het problem.f64
#include <chrono>
#include <cstdio>
#include <vector>

#define LIO_DOGLEG_CLOCK 1
#include "solve_step.h"
using namespace lio;

namespace {
struct FrameData { std::vector<double> pts, coef; };
void cpu_moments(const FrameData &fd, const double R[9], const double t[3], FrameMoments &m) {
  for (double &v : m.S) v = 0;
  double lg = 0;
  const size_t n = fd.pts.size() / 3;
  for (size_t s = 0; s < n; ++s) {
    const double *p = &fd.pts[3 * s], *c = &fd.coef[4 * s];
    const double qx = R[0] * p[0] + R[1] * p[1] + R[2] * p[2] + t[0], qy = R[3] * p[0] + R[4] * p[1] + R[5] * p[2] + t[1],
                 qz = R[6] * p[0] + R[7] * p[1] + R[8] * p[2] + t[2];
    const double r = c[0] * qx + c[1] * qy + c[2] * qz + c[3];
    const double sw = 1.0 / std::sqrt(1.0 + r * r);
    double z[16] = {0};
    for (int a = 0; a < 3; ++a) { z[4 * a] = sw * c[a] * p[0]; z[4 * a + 1] = sw * c[a] * p[1]; z[4 * a + 2] = sw * c[a] * p[2]; z[4 * a + 3] = sw * c[a]; }
    z[12] = sw * c[3];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) m.S[i * 16 + j] += z[i] * z[j];
    lg += std::log(1.0 + r * r);
  }
  m.cost = 0.5 * lg; m.count = double(n);
}
double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  std::vector<double> d;
  {
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    double buf[4096]; size_t k;
    while ((k = std::fread(buf, sizeof(double), 4096, f)) > 0) d.insert(d.end(), buf, buf + k);
    std::fclose(f);
  }
  size_t at = 0;
  auto next = [&]() { return d.at(at++); };
  const int Wo = int(next());
  const bool ex_constant = next() != 0, has_prior = next() != 0, use_prior_factor = next() != 0;
  const int max_it = int(next());
  PimNoise noise;
  noise.acc_n = next(); noise.gyr_n = next(); noise.acc_w = next(); noise.gyr_w = next(); noise.g_norm = next();
  double pr7[7];
  for (double &v : pr7) v = next();
  auto read_params = [&](WindowParams &P) {
    P.Wo = Wo; P.pose.resize(Wo + 1); P.sb.resize(Wo + 1);
    for (int i = 0; i <= Wo; ++i) { for (double &v : P.pose[i]) v = next(); for (double &v : P.sb[i]) v = next(); }
    for (double &v : P.ex) v = next();
    P.ex_constant = ex_constant;
  };
  WindowParams P0, Pm;
  read_params(P0); read_params(Pm);
  WindowSystem sys;
  sys.Wo = Wo; sys.use_lidar = true;
  sys.pim.assign(Wo, nullptr);
  for (int i = 0; i < Wo; ++i) {
    const int n = int(next());
    if (n < 0) continue;
    double h[12];
    for (double &v : h) v = next();
    auto pm = std::make_shared<Preintegration>(V3d(h[0], h[1], h[2]), V3d(h[3], h[4], h[5]), V3d(h[6], h[7], h[8]), V3d(h[9], h[10], h[11]), noise);
    for (int s = 0; s < n; ++s) { double v[7]; for (double &x : v) x = next(); pm->push_back(v[0], V3d(v[1], v[2], v[3]), V3d(v[4], v[5], v[6])); }
    sys.pim[i] = pm;
  }
  std::vector<FrameData> fr(Wo + 1);
  for (int i = 1; i <= Wo; ++i) {
    const size_t n = size_t(next());
    fr[i].pts.resize(3 * n); fr[i].coef.resize(4 * n);
    for (double &v : fr[i].pts) v = next();
    for (double &v : fr[i].coef) v = next();
  }
  if (has_prior) {
    auto pr = std::make_shared<MargPrior>();
    pr->n = int(next());
    const int nb = int(next());
    for (int k = 0; k < nb; ++k) { KeepBlock kb; kb.kind = int(next()); kb.index = int(next()); kb.size = int(next()); kb.idx = int(next()); pr->keep.push_back(kb); }
    for (const KeepBlock &kb : pr->keep) { std::vector<double> x(kb.size); for (double &v : x) v = next(); pr->x0.push_back(x); }
    pr->lin_jac = DMat(pr->n, pr->n);
    for (double &v : pr->lin_jac.a) v = next();
    pr->lin_res.resize(pr->n);
    for (double &v : pr->lin_res) v = next();
    pr->finalize();
    sys.prior = pr;
  }
  sys.use_prior_factor = use_prior_factor;
  sys.prior_pos = V3d(pr7[0], pr7[1], pr7[2]); sys.prior_rot = Qd(pr7[6], pr7[3], pr7[4], pr7[5]);
  // ---- pass 1: record the moments of every linearisation
  std::vector<std::vector<FrameMoments>> tape;
  sys.lidar_eval = [&](const WindowParams &P, std::vector<FrameMoments> &m) {
    for (int i = 1; i <= Wo; ++i) {
      double R[9], t[3];
      relative_lidar_pose(P.pose[0].data(), P.pose[i].data(), P.ex.data(), R, t);
      cpu_moments(fr[i], R, t, m[i]);
    }
    tape.push_back(m);
  };
  WindowParams P = P0;
  SolveSummary s0 = solve_dogleg(sys, P, max_it, -1.0, nullptr);
  std::printf("recorded: %d iterations, %zu passes, final cost %.12g, dim %d, prior %d\n", s0.iterations, tape.size(), s0.final_cost,
              WindowSystem::solve_layout(P0).dim, sys.prior ? sys.prior->n : 0);
  // ---- pass 2: replay
  size_t call = 0;
  sys.lidar_eval = nullptr;
  sys.lidar_launch = [&](const WindowParams &) {};
  sys.lidar_wait = [&](std::vector<FrameMoments> &m) { m = tape.at(call); ++call; };
  sys.lidar_wait_frame = [&](int i, FrameMoments &m) { m = tape.at(call)[i]; if (i == Wo) ++call; return true; };
  const int reps = argc > 2 ? std::atoi(argv[2]) : 300;
  double best = 1e30, sum = 0;
  SolveSummary s;
  double piece[6] = {1e30, 1e30, 1e30, 1e30, 1e30, 1e30}, loopc[5] = {1e30, 1e30, 1e30, 1e30, 1e30};   // per-piece minima over the repetitions (a shared host is noisy)
  for (int r = 0; r < reps; ++r) {
    call = 0; P = P0;
    sys.eclk = WindowSystem::EvalClock();
    dogleg_clock() = DoglegClock();
    const double t0 = now_us();
    s = solve_dogleg(sys, P, max_it, -1.0, nullptr);
    const double t1 = now_us();
    best = std::min(best, t1 - t0);
    sum += t1 - t0;
    const double v[6] = {1e3 * sys.eclk.launch, 1e3 * sys.eclk.prior, 1e3 * sys.eclk.imu, 1e3 * sys.eclk.wait, 1e3 * sys.eclk.assemble, 1e3 * s.ms_chol};
    for (int q = 0; q < 6; ++q) piece[q] = std::min(piece[q], v[q]);
    for (int q = 0; q < 5; ++q) loopc[q] = std::min(loopc[q], 1e3 * dogleg_clock().t[q]);
  }
  const double np = double(tape.size());
  std::printf("replayed: %d iterations, final cost %.12g (same: %s)\n", s.iterations, s.final_cost, s.final_cost == s0.final_cost ? "yes" : "NO");
  std::printf("host time per solve: best %.1f us, mean %.1f us; per pass (best): %.2f us\n", best, sum / reps, best / np);
  std::printf("  per pass (minimum of each piece): launch %.2f prior %.2f imu+maps(+prefactor) %.2f wait(copy) %.2f assemble %.2f chol+solve %.2f rest %.2f us\n",
              piece[0] / np, piece[1] / np, piece[2] / np, piece[3] / np, piece[4] / np, piece[5] / np,
              (best - (piece[0] + piece[1] + piece[2] + piece[3] + piece[4] + piece[5])) / np);
  std::printf("  loop, per iteration (minimum of each piece): diag + Cauchy %.2f | factor + GN step %.2f | dogleg + model change %.2f | candidate %.2f | acceptance %.2f us\n",
              loopc[0] / s.iterations, loopc[1] / s.iterations, loopc[2] / s.iterations, loopc[3] / s.iterations, loopc[4] / s.iterations);
  return 0;
}
