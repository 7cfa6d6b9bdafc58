// Stand-alone check (host code only, built with hipcc by tests/test_ref_solve_problem.py): the PRODUCT's host solver
// (csrc/host_solver.h: solve_dogleg, marginalize; csrc/host_factors.h: the IMU / prior factors) on a sliding-window problem exactly as
// THE REFERENCE's Estimator::SolveOptimization handed it to ceres::Solve — parameter blocks, raw IMU samples of every interval, every
// plane factor, the marginalization prior, the extrinsic prior (tests/golden/ref_solve_problems.npz, dumped from the reference's own
// Estimator.cc by tests/golden/make_ref_solve_problems.py).  The lidar moments the GPU kernels would return are formed here on the CPU
// by their defining sums (as in solve_step_check.hip; the kernels are held to those sums on the GPU by tests/test_gpu_parity.py).
//   argv[1]: the problem, a flat float64 file (layout: see read below);  stdout: the solve's result, then the new prior.
#include <cstdio>
#include <vector>

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
  // header: Wo, extrinsic constant, a prior is present, the extrinsic PriorFactor is present, max iterations, noise (5), prior pos + rot (7)
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
  read_params(P0);   // where the solve starts
  read_params(Pm);   // where the reference linearised its marginalization (after DoubleToVector / VectorToDouble)
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
  if (at != d.size()) { std::fprintf(stderr, "layout mismatch: read %zu of %zu\n", at, d.size()); return 2; }
  sys.lidar_eval = [&](const WindowParams &P, std::vector<FrameMoments> &m) {
    for (int i = 1; i <= Wo; ++i) {
      double R[9], t[3];
      relative_lidar_pose(P.pose[0].data(), P.pose[i].data(), P.ex.data(), R, t);
      cpu_moments(fr[i], R, t, m[i]);
    }
  };
  if (std::getenv("LIO_CHECK_DUMP_HG")) {   // debugging aid: the first linearisation
    Layout lay = WindowSystem::solve_layout(P0);
    DMat H; std::vector<double> g;
    WindowSystem::Costs c = sys.evaluate(P0, lay, 15, false, &H, &g);
    std::printf("costs %.17g %.17g %.17g %.17g\nH", c.marg, c.pim, c.ppp, c.prior);
    for (double v : H.a) std::printf(" %.17g", v);
    std::printf("\ng");
    for (double v : g) std::printf(" %.17g", v);
    std::printf("\n");
  }
  WindowParams P = P0;
  SolveSummary s = solve_dogleg(sys, P, max_it, -1.0, nullptr);
  std::printf("summary %d %d %d %.17g %.17g\n", s.iterations, s.successful, s.termination, s.initial_cost, s.final_cost);
  std::printf("trace");
  for (double v : s.trace) std::printf(" %.17g", v);
  std::printf("\nparams");
  for (int i = 0; i <= Wo; ++i) { for (double v : P.pose[i]) std::printf(" %.17g", v); for (double v : P.sb[i]) std::printf(" %.17g", v); }
  for (double v : P.ex) std::printf(" %.17g", v);
  std::printf("\n");
  // the marginalization at the reference's linearisation point: the old prior, ImuFactor(0 -> 1), every plane factor
  WindowSystem ms = sys;
  ms.lidar_eval = sys.lidar_eval;
  for (int i = 1; i < Wo; ++i) ms.pim[i] = nullptr;
  auto np = marginalize(ms, Pm);
  std::printf("prior %d\nJtJ", np->n);
  for (double v : np->JtJ.a) std::printf(" %.17g", v);
  std::printf("\nJtr");
  for (double v : np->Jtr0) std::printf(" %.17g", v);
  std::printf("\nx0");
  for (const auto &b : np->x0) for (double v : b) std::printf(" %.17g", v);
  std::printf("\n");
  return 0;
}
