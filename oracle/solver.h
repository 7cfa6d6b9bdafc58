// oracle/solver.h — TEST INFRASTRUCTURE ONLY (CPU oracle).  Not part of the product.
//
// (1) The sliding-window problem the reference hands to Ceres (Estimator.cc:1747-1904) evaluated into
//     dense normal equations block by block, with the robust-loss corrector of
//     MarginalizationFactor.cc:69-95 (identical to Ceres' Corrector).
// (2) A restatement of Ceres-solver 1.14.0's TrustRegionMinimizer + DoglegStrategy(TRADITIONAL)
//     with an exact dense solve in place of DENSE_SCHUR (algebraically equal).
//     Ceres is a third-party dependency pinned only in docker/Dockerfile:43-46 and absent from
//     /root/reference; the reference's tests hold no golden vector for it => PARITY UNPINNED for
//     the minimizer's step sequence.  The published algorithm restated here: Jacobi scaling fixed at
//     iteration 0 (1/(1+sqrt(colnorm^2))); dogleg diagonal = sqrt(clamp(diag(J^T J),1e-6,1e32));
//     mu in [1e-8,1] x10 on linear-solver failure, /5 on accepted step; radius0 1e4, x0.5 when
//     rho<0.25 or rejected, max(radius,3|step|) when rho>0.75; accept when rho>1e-3; function_tolerance
//     1e-6, parameter_tolerance 1e-8, gradient_tolerance 1e-10, max_consecutive_invalid_steps 5.
//     Round 3: the PROBLEM of (1) is pinned against the reference's own SolveOptimization (oracle/ref_estimator.cc; the stand-in
//     ceres::Solve there is this same restatement over generic blocks, so the minimizer stays unpinned) — equal factor counts and
//     costs within 1e-8 on 63 estimator steps, tests/test_ref_estimator_run.py.
// (3) MarginalizationInfo::Marginalize (MarginalizationFactor.cc:185-311) and
//     MarginalizationFactor::Evaluate (:343-393) in the canonical block order of SURVEY.md A.13.
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <memory>
#include <thread>

#include "imu.h"

namespace orc {

struct PlaneFeature {   // feature_manager/FeatureManager.h:84-100 PointPlaneFeature
  double score;
  V3d point;            // in lidar frame i
  double coeffs[4];     // (w, d) in the pivot lidar frame, already scaled by s
};

// Kept-block descriptor of a marginalization prior, in canonical order.
struct KeepBlock {
  int kind;   // 0 pose, 1 speed-bias, 2 extrinsic
  int index;  // opt-window index AFTER the address shift (Estimator.cc:2230-2238)
  int size;   // ambient size 7 / 9
  int idx;    // column offset inside the prior (local/tangent coordinates)
};

struct MargPrior {
  int n = 0;
  std::vector<KeepBlock> keep;
  std::vector<std::vector<double>> x0;  // keep_block_data
  Mat lin_jac;                          // n x n  linearized_jacobians
  std::vector<double> lin_res;          // n      linearized_residuals
};

struct WindowProblem {
  int Wo = 0;
  std::vector<std::array<double, 7>> pose;   // para_pose_[0..Wo]
  std::vector<std::array<double, 9>> sb;     // para_speed_bias_[0..Wo]
  std::array<double, 7> ex{};                // para_ex_pose_
  bool ex_constant = true;
  std::vector<std::shared_ptr<IntegrationBase>> pim;  // [i] links opt i -> i+1 (null = skipped)
  std::vector<std::vector<PlaneFeature>> feats;       // [i], i = 1..Wo used
  std::shared_ptr<MargPrior> prior;                   // null = none
  bool use_prior_factor = false;
  V3d prior_pos; Qd prior_rot;
  bool use_lidar = true, use_imu = true;
  // factor sharding across ranks (lio_est_set_factor_sharding): this rank evaluates items [lo, hi) of each frame
  int shard_rank = 0, shard_world = 1;
  int (*allreduce)(double *, int, void *) = nullptr;
  void *allreduce_user = nullptr;

  int D() const { return 15 * (Wo + 1) + (ex_constant ? 0 : 6); }
  int colPose(int i) const { return 15 * i; }
  int colSb(int i) const { return 15 * i + 6; }
  int colEx() const { return ex_constant ? -1 : 15 * (Wo + 1); }
};

// Column layout abstraction so the same accumulation serves the solve and the marginalization.
struct Layout {
  std::vector<int> pose, sb;  // -1 = not a variable in this system
  int ex = -1;
  int dim = 0;
};

// ρ for CauchyLoss(1.0): rho[0]=log(1+s), rho[1]=max(min_double,1/(1+s)), rho[2]=-(1/(1+s))^2
static inline void CauchyLoss(double s, double rho[3]) {
  const double sum = 1.0 + s, inv = 1.0 / sum;
  rho[0] = std::log(sum);
  rho[1] = std::max(std::numeric_limits<double>::min(), inv);
  rho[2] = -(inv * inv);
}

struct BlockRef { int col; int local; int ambient; const double *J; };

// Adds one residual block: robust correction (MarginalizationFactor.cc:69-95 == ceres::Corrector),
// then H(bi,bj) += Ji^T Jj, g(bi) += Ji^T r in local coordinates (first `local` columns of each
// ambient Jacobian: PoseLocalParameterization::ComputeJacobian = [I6;0]).
static inline double AccumulateBlock(int nres, double *r, std::vector<BlockRef> &blocks, std::vector<std::vector<double>> &Jstore,
                                     bool loss, Mat *H, std::vector<double> *g) {
  double sq = 0;
  for (int i = 0; i < nres; ++i) sq += r[i] * r[i];
  double cost = 0.5 * sq;
  if (loss) {
    double rho[3];
    CauchyLoss(sq, rho);
    cost = 0.5 * rho[0];
    double sqrt_rho1 = std::sqrt(rho[1]);
    double residual_scaling, alpha_sq_norm;
    if (sq == 0.0 || rho[2] <= 0.0) { residual_scaling = sqrt_rho1; alpha_sq_norm = 0.0; }
    else {
      const double Dd = 1.0 + 2.0 * sq * rho[2] / rho[1];
      const double alpha = 1.0 - std::sqrt(Dd);
      residual_scaling = sqrt_rho1 / (1 - alpha);
      alpha_sq_norm = alpha / sq;
    }
    if (H) {
      for (size_t b = 0; b < blocks.size(); ++b) {
        std::vector<double> &J = Jstore[b];
        int amb = blocks[b].ambient;
        if (alpha_sq_norm != 0.0) {
          std::vector<double> rtJ(amb, 0.0);
          for (int i = 0; i < nres; ++i) for (int c = 0; c < amb; ++c) rtJ[c] += r[i] * J[i * amb + c];
          for (int i = 0; i < nres; ++i) for (int c = 0; c < amb; ++c) J[i * amb + c] = sqrt_rho1 * (J[i * amb + c] - alpha_sq_norm * r[i] * rtJ[c]);
        } else {
          for (double &v : J) v *= sqrt_rho1;
        }
      }
    }
    for (int i = 0; i < nres; ++i) r[i] *= residual_scaling;
  }
  if (!H) return cost;
  for (size_t a = 0; a < blocks.size(); ++a) {
    if (blocks[a].col < 0) continue;
    const double *Ja = Jstore[a].data(); int la = blocks[a].local, aa = blocks[a].ambient, ca = blocks[a].col;
    for (size_t b = a; b < blocks.size(); ++b) {
      if (blocks[b].col < 0) continue;
      const double *Jb = Jstore[b].data(); int lb = blocks[b].local, ab = blocks[b].ambient, cb = blocks[b].col;
      for (int i = 0; i < la; ++i)
        for (int j = 0; j < lb; ++j) {
          double s = 0;
          for (int k = 0; k < nres; ++k) s += Ja[k * aa + i] * Jb[k * ab + j];
          (*H)(ca + i, cb + j) += s;
          if (a != b) (*H)(cb + j, ca + i) += s;
        }
    }
    for (int i = 0; i < la; ++i) { double s = 0; for (int k = 0; k < nres; ++k) s += Ja[k * aa + i] * r[k]; (*g)[ca + i] += s; }
  }
  return cost;
}

// MarginalizationFactor::Evaluate (:343-393)
static inline void MargPriorResidual(const MargPrior &pr, const WindowProblem &P, std::vector<double> &res) {
  std::vector<double> dx(pr.n, 0.0);
  for (size_t b = 0; b < pr.keep.size(); ++b) {
    const KeepBlock &kb = pr.keep[b];
    const double *x = kb.kind == 0 ? P.pose[kb.index].data() : (kb.kind == 1 ? P.sb[kb.index].data() : P.ex.data());
    const double *x0 = pr.x0[b].data();
    if (kb.size != 7) { for (int k = 0; k < kb.size; ++k) dx[kb.idx + k] = x[k] - x0[k]; }
    else {
      for (int k = 0; k < 3; ++k) dx[kb.idx + k] = x[k] - x0[k];
      Qd q0(x0[6], x0[3], x0[4], x0[5]), q(x[6], x[3], x[4], x[5]);
      Qd dq = q0.inverse() * q;
      V3d v = 2.0 * dq.normalized().vec();
      if (dq.w < 0) v = 2.0 * (-(dq.normalized().vec()));
      for (int k = 0; k < 3; ++k) dx[kb.idx + 3 + k] = v[k];
    }
  }
  res.assign(pr.n, 0.0);
  for (int i = 0; i < pr.n; ++i) { double s = pr.lin_res[i]; for (int j = 0; j < pr.n; ++j) s += pr.lin_jac(i, j) * dx[j]; res[i] = s; }
}

struct GroupCosts { double marg = 0, pim = 0, ppp = 0, prior = 0; };

// Evaluate the whole window problem in `lay` coordinates.  which: bit0 marg prior, bit1 imu, bit2 lidar,
// bit3 extrinsic prior factor.  imu_only_first: marginalization uses only ImuFactor(0->1).
static inline double EvaluateProblem(const WindowProblem &P, const Layout &lay, int which, bool imu_only_first, Mat *H,
                                     std::vector<double> *g, GroupCosts *gc = nullptr, int threads = 1) {
  double cost = 0;
  if (H) { *H = Mat(lay.dim, lay.dim); g->assign(lay.dim, 0.0); }
  // --- marginalization prior (added first: Estimator.cc:1779-1787), loss NULL
  if ((which & 1) && P.prior) {
    const MargPrior &pr = *P.prior;
    std::vector<double> res;
    MargPriorResidual(pr, P, res);
    std::vector<BlockRef> blocks;
    std::vector<std::vector<double>> Js;
    for (const KeepBlock &kb : pr.keep) {
      int col = kb.kind == 0 ? lay.pose[kb.index] : (kb.kind == 1 ? lay.sb[kb.index] : lay.ex);
      int local = kb.size == 7 ? 6 : kb.size;
      std::vector<double> J(size_t(pr.n) * kb.size, 0.0);
      if (H) for (int i = 0; i < pr.n; ++i) for (int c = 0; c < local; ++c) J[size_t(i) * kb.size + c] = pr.lin_jac(i, kb.idx + c);
      Js.push_back(std::move(J));
      blocks.push_back({col, local, kb.size, nullptr});
    }
    double c = AccumulateBlock(pr.n, res.data(), blocks, Js, false, H, g);
    cost += c; if (gc) gc->marg += c;
  }
  // --- IMU factors (Estimator.cc:1792-1827), loss NULL
  if (which & 2) {
    int last = imu_only_first ? 1 : P.Wo;
    for (int i = 0; i < last; ++i) {
      if (!P.pim[i]) continue;
      const double *par[4] = {P.pose[i].data(), P.sb[i].data(), P.pose[i + 1].data(), P.sb[i + 1].data()};
      double r[15];
      std::vector<std::vector<double>> Js(4);
      Js[0].resize(15 * 7); Js[1].resize(15 * 9); Js[2].resize(15 * 7); Js[3].resize(15 * 9);
      double *jp[4] = {Js[0].data(), Js[1].data(), Js[2].data(), Js[3].data()};
      ImuFactorEvaluate(*P.pim[i], par, r, H ? jp : nullptr);
      std::vector<BlockRef> blocks = {{lay.pose[i], 6, 7, nullptr}, {lay.sb[i], 9, 9, nullptr}, {lay.pose[i + 1], 6, 7, nullptr}, {lay.sb[i + 1], 9, 9, nullptr}};
      double c = AccumulateBlock(15, r, blocks, Js, false, H, g);
      cost += c; if (gc) gc->pim += c;
    }
  }
  // --- lidar factors (Estimator.cc:1831-1889), CauchyLoss(1.0); frame 0 has none (A.16)
  if ((which & 4) && P.use_lidar) {
    // The reference sums sequentially (Ceres num_threads=1, Estimator.cc:1913) in the solve and over
    // 4 round-robin pthreads in Marginalize (MarginalizationFactor.cc:245-269; partials added in
    // thread order 3,2,1,0).  `threads` selects which.
    struct Item { int frame; const PlaneFeature *f; };
    std::vector<Item> items;
    const bool sharded = P.shard_world > 1 && P.allreduce;
    for (int i = 1; i <= P.Wo; ++i) {
      const size_t nf = P.feats[i].size();
      size_t lo = 0, hi = nf;
      if (sharded) { lo = nf * size_t(P.shard_rank) / size_t(P.shard_world); hi = nf * size_t(P.shard_rank + 1) / size_t(P.shard_world); }
      for (size_t k = lo; k < hi; ++k) items.push_back({i, &P.feats[i][k]});
    }
    if (sharded) {
      // per-shard lidar normal equations, summed over ranks, then added to the (replicated) rest of the system
      Mat Hl(H ? lay.dim : 0, H ? lay.dim : 0);
      std::vector<double> gl(H ? lay.dim : 0, 0.0);
      double cl = 0;
      std::vector<std::vector<double>> Js(3, std::vector<double>(7));
      for (const Item &it : items) {
        const double *par[3] = {P.pose[0].data(), P.pose[it.frame].data(), P.ex.data()};
        double r[1];
        double *jp[3] = {Js[0].data(), Js[1].data(), Js[2].data()};
        PivotPointPlaneEvaluate(it.f->point, it.f->coeffs, par, r, H ? jp : nullptr);
        std::vector<BlockRef> blocks = {{lay.pose[0], 6, 7, nullptr}, {lay.pose[it.frame], 6, 7, nullptr}, {lay.ex, 6, 7, nullptr}};
        cl += AccumulateBlock(1, r, blocks, Js, true, H ? &Hl : nullptr, H ? &gl : nullptr);
      }
      std::vector<double> buf;
      if (H) { buf = Hl.a; buf.insert(buf.end(), gl.begin(), gl.end()); }
      buf.push_back(cl);
      P.allreduce(buf.data(), int(buf.size()), P.allreduce_user);
      if (H) {
        for (size_t k = 0; k < H->a.size(); ++k) H->a[k] += buf[k];
        for (int k = 0; k < lay.dim; ++k) (*g)[k] += buf[H->a.size() + k];
      }
      cost += buf.back(); if (gc) gc->ppp += buf.back();
      items.clear();
    }
    auto work = [&](int tid, int nth, Mat *Hl, std::vector<double> *gl, double *cl) {
      std::vector<std::vector<double>> Js(3, std::vector<double>(7));
      for (size_t k = tid; k < items.size(); k += nth) {
        int i = items[k].frame;
        const PlaneFeature &f = *items[k].f;
        const double *par[3] = {P.pose[0].data(), P.pose[i].data(), P.ex.data()};
        double r[1];
        double *jp[3] = {Js[0].data(), Js[1].data(), Js[2].data()};
        PivotPointPlaneEvaluate(f.point, f.coeffs, par, r, Hl ? jp : nullptr);
        std::vector<BlockRef> blocks = {{lay.pose[0], 6, 7, nullptr}, {lay.pose[i], 6, 7, nullptr}, {lay.ex, 6, 7, nullptr}};
        *cl += AccumulateBlock(1, r, blocks, Js, true, Hl, gl);
      }
    };
    if (sharded) {
      // already accumulated above
    } else if (threads <= 1 || !H) {
      double c = 0;
      work(0, 1, H, g, &c);
      cost += c; if (gc) gc->ppp += c;
    } else {
      std::vector<Mat> Hs(threads, Mat(lay.dim, lay.dim));
      std::vector<std::vector<double>> gs(threads, std::vector<double>(lay.dim, 0.0));
      std::vector<double> cs(threads, 0.0);
      std::vector<std::thread> th;
      for (int t = 0; t < threads; ++t) th.emplace_back(work, t, threads, &Hs[t], &gs[t], &cs[t]);
      for (auto &t : th) t.join();
      for (int t = threads - 1; t >= 0; --t) {
        for (size_t k = 0; k < H->a.size(); ++k) H->a[k] += Hs[t].a[k];
        for (int k = 0; k < lay.dim; ++k) (*g)[k] += gs[t][k];
        cost += cs[t]; if (gc) gc->ppp += cs[t];
      }
    }
  }
  // --- extrinsic prior (Estimator.cc:1891-1904), loss NULL
  if ((which & 8) && P.use_prior_factor) {
    double r[6];
    std::vector<std::vector<double>> Js(1, std::vector<double>(42));
    PriorFactorEvaluate(P.prior_pos, P.prior_rot, P.ex.data(), r, H ? Js[0].data() : nullptr);
    std::vector<BlockRef> blocks = {{lay.ex, 6, 7, nullptr}};
    double c = AccumulateBlock(6, r, blocks, Js, false, H, g);
    cost += c; if (gc) gc->prior += c;
  }
  return cost;
}

static inline Layout SolveLayout(const WindowProblem &P) {
  Layout l;
  l.pose.resize(P.Wo + 1); l.sb.resize(P.Wo + 1);
  for (int i = 0; i <= P.Wo; ++i) { l.pose[i] = P.colPose(i); l.sb[i] = P.colSb(i); }
  l.ex = P.colEx();
  l.dim = P.D();
  return l;
}

struct SolveSummary {
  int iterations = 0, successful = 0, termination = 0;
  double initial_cost = 0, final_cost = 0;
  std::vector<double> cost_trace;
};

static inline void PlusAll(const WindowProblem &P, const std::vector<double> &delta, WindowProblem &out) {
  out = P;
  for (int i = 0; i <= P.Wo; ++i) {
    PosePlus(P.pose[i].data(), &delta[P.colPose(i)], out.pose[i].data());
    for (int k = 0; k < 9; ++k) out.sb[i][k] = P.sb[i][k] + delta[P.colSb(i) + k];
  }
  if (!P.ex_constant) PosePlus(P.ex.data(), &delta[P.colEx()], out.ex.data());
}
static inline double AmbientNorm(const WindowProblem &P, const WindowProblem *other = nullptr) {
  double s = 0;
  auto acc = [&](const double *a, const double *b, int n) { for (int k = 0; k < n; ++k) { double d = b ? a[k] - b[k] : a[k]; s += d * d; } };
  for (int i = 0; i <= P.Wo; ++i) { acc(P.pose[i].data(), other ? other->pose[i].data() : nullptr, 7); acc(P.sb[i].data(), other ? other->sb[i].data() : nullptr, 9); }
  if (!P.ex_constant) acc(P.ex.data(), other ? other->ex.data() : nullptr, 7);
  return std::sqrt(s);
}

// Optional record of a solve for tests/golden (second-sourcing the trust-region logic): every linearisation the minimizer
// used (UNSCALED J^T J, J^T r, cost) and, per iteration, the state it entered with and what it decided.
struct DoglegDump {
  int n = 0;
  std::vector<std::vector<double>> H, g;   // one per linearisation, index 0 = the initial point
  std::vector<double> cost;
  struct It { int lin; double radius, mu, cand_cost, model_change, step_norm, x_norm, gmax; int valid, accepted, termination; std::vector<double> delta; };
  std::vector<It> its;
};
static inline DoglegDump *&dogleg_dump_sink() { static DoglegDump *p = nullptr; return p; }

// Ceres 1.14 TrustRegionMinimizer::Minimize with DoglegStrategy (TRADITIONAL_DOGLEG).
static inline SolveSummary SolveDogleg(WindowProblem &P, int max_num_iterations, double max_time_s) {
  DoglegDump *dump = dogleg_dump_sink();
  using clock = std::chrono::steady_clock;
  auto t_start = clock::now();
  SolveSummary sum;
  const int which = 1 | 2 | 4 | 8;
  Layout lay = SolveLayout(P);
  const int n = lay.dim;
  Mat H; std::vector<double> g;
  double x_cost = EvaluateProblem(P, lay, which, false, &H, &g);
  sum.initial_cost = x_cost; sum.cost_trace.push_back(x_cost);
  if (dump) { dump->n = n; dump->H.push_back(H.a); dump->g.push_back(g); dump->cost.push_back(x_cost); }
  // Jacobi scaling (iteration 0 only)
  std::vector<double> scale(n);
  for (int i = 0; i < n; ++i) scale[i] = 1.0 / (1.0 + std::sqrt(H(i, i)));
  auto scaleSystem = [&](Mat &Hs, std::vector<double> &gs) {
    for (int i = 0; i < n; ++i) { gs[i] *= scale[i]; for (int j = 0; j < n; ++j) Hs(i, j) *= scale[i] * scale[j]; }
  };
  auto gradMaxNorm = [&](const std::vector<double> &g_unscaled) {
    std::vector<double> neg(n);
    for (int i = 0; i < n; ++i) neg[i] = -g_unscaled[i];
    WindowProblem Pp; PlusAll(P, neg, Pp);
    double mx = 0;
    auto acc = [&](const double *a, const double *b, int m) { for (int k = 0; k < m; ++k) mx = std::max(mx, std::fabs(a[k] - b[k])); };
    for (int i = 0; i <= P.Wo; ++i) { acc(P.pose[i].data(), Pp.pose[i].data(), 7); acc(P.sb[i].data(), Pp.sb[i].data(), 9); }
    if (!P.ex_constant) acc(P.ex.data(), Pp.ex.data(), 7);
    return mx;
  };
  double gmax = gradMaxNorm(g);
  scaleSystem(H, g);
  double x_norm = AmbientNorm(P);
  // dogleg state
  double radius = 1e4, mu = 1e-8;
  const double min_mu = 1e-8, max_mu = 1.0, mu_inc = 10.0, min_diag = 1e-6, max_diag = 1e32;
  bool reuse = false;
  std::vector<double> diagonal(n), gradient(n), gn(n);
  double alpha = 0, dogleg_step_norm = 0;
  int consecutive_invalid = 0;
  int iteration = 0;
  while (true) {
    // FinalizeIterationAndCheckIfMinimizerCanContinue
    if (iteration >= max_num_iterations) { sum.termination = 0; break; }
    if (max_time_s > 0 && std::chrono::duration<double>(clock::now() - t_start).count() >= max_time_s) { sum.termination = 4; break; }
    if (gmax <= 1e-10) { sum.termination = 3; break; }
    if (radius <= 1e-32) { sum.termination = 1; break; }
    ++iteration;
    // ---- DoglegStrategy::ComputeStep
    bool linear_ok = true;
    if (!reuse) {
      reuse = true;
      for (int i = 0; i < n; ++i) diagonal[i] = std::sqrt(std::min(std::max(H(i, i), min_diag), max_diag));
      for (int i = 0; i < n; ++i) gradient[i] = g[i] / diagonal[i];
      // Cauchy point
      std::vector<double> sg(n);
      for (int i = 0; i < n; ++i) sg[i] = gradient[i] / diagonal[i];
      std::vector<double> Hsg = matvec(H, sg);
      double Jg2 = 0, g2 = 0;
      for (int i = 0; i < n; ++i) { Jg2 += sg[i] * Hsg[i]; g2 += gradient[i] * gradient[i]; }
      alpha = g2 / Jg2;
      // Gauss-Newton step with mu regularisation
      linear_ok = false;
      while (mu < max_mu) {
        Mat A = H;
        for (int i = 0; i < n; ++i) A(i, i) += diagonal[i] * diagonal[i] * mu;  // D^2, D = diagonal*sqrt(mu)
        Mat L;
        bool ok = cholesky(A, L);
        if (ok) {
          gn = g;
          chol_solve(L, gn);
          for (int i = 0; i < n; ++i) if (!std::isfinite(gn[i])) ok = false;
        }
        if (!ok) { mu *= mu_inc; continue; }
        linear_ok = true;
        break;
      }
      if (linear_ok) for (int i = 0; i < n; ++i) gn[i] *= -diagonal[i];
    }
    std::vector<double> step(n, 0.0);
    bool step_valid = linear_ok;
    double model_cost_change = 0;
    if (linear_ok) {
      double gnorm = 0, gnn = 0;
      for (int i = 0; i < n; ++i) { gnorm += gradient[i] * gradient[i]; gnn += gn[i] * gn[i]; }
      gnorm = std::sqrt(gnorm); gnn = std::sqrt(gnn);
      if (gnn <= radius) { step = gn; dogleg_step_norm = gnn; }
      else if (gnorm * alpha >= radius) { for (int i = 0; i < n; ++i) step[i] = -(radius / gnorm) * gradient[i]; dogleg_step_norm = radius; }
      else {
        double gdot = 0;
        for (int i = 0; i < n; ++i) gdot += gradient[i] * gn[i];
        const double b_dot_a = -alpha * gdot;
        const double a_sq = std::pow(alpha * gnorm, 2.0);
        const double bma_sq = a_sq - 2 * b_dot_a + std::pow(gnn, 2);
        const double c = b_dot_a - a_sq;
        const double d = std::sqrt(c * c + bma_sq * (std::pow(radius, 2.0) - a_sq));
        double beta = (c <= 0) ? (d - c) / bma_sq : (radius * radius - a_sq) / (d + c);
        double sn = 0;
        for (int i = 0; i < n; ++i) { step[i] = (-alpha * (1.0 - beta)) * gradient[i] + beta * gn[i]; sn += step[i] * step[i]; }
        dogleg_step_norm = std::sqrt(sn);
      }
      for (int i = 0; i < n; ++i) step[i] /= diagonal[i];
      // model_cost_change = -(step^T g + 0.5 step^T H step)
      std::vector<double> Hs = matvec(H, step);
      double sg = 0, sHs = 0;
      for (int i = 0; i < n; ++i) { sg += step[i] * g[i]; sHs += step[i] * Hs[i]; }
      model_cost_change = -(sg + 0.5 * sHs);
      if (!(model_cost_change > 0)) step_valid = false;
    }
    if (!step_valid) {
      if (dump) dump->its.push_back({int(dump->H.size()) - 1, radius, mu, 0.0, model_cost_change, 0.0, x_norm, gmax, 0, 0, -1, std::vector<double>(n, 0.0)});
      if (++consecutive_invalid >= 5) { sum.termination = 5; break; }
      mu *= mu_inc; reuse = false;  // StepIsInvalid
      sum.cost_trace.push_back(x_cost);
      continue;
    }
    consecutive_invalid = 0;
    std::vector<double> delta(n);
    for (int i = 0; i < n; ++i) delta[i] = step[i] * scale[i];
    WindowProblem cand;
    PlusAll(P, delta, cand);
    double cand_cost = EvaluateProblem(cand, lay, which, false, nullptr, nullptr);
    // ParameterToleranceReached
    double step_norm = AmbientNorm(P, &cand);
    if (dump) dump->its.push_back({int(dump->H.size()) - 1, radius, mu, cand_cost, model_cost_change, step_norm, x_norm, gmax, 1, 0, -1, delta});
    if (step_norm <= 1e-8 * (x_norm + 1e-8)) { sum.termination = 1; sum.cost_trace.push_back(x_cost); break; }
    // FunctionToleranceReached
    double cost_change = x_cost - cand_cost;
    if (std::fabs(cost_change) <= 1e-6 * x_cost) { sum.termination = 2; sum.cost_trace.push_back(x_cost); break; }
    double relative_decrease = cost_change / model_cost_change;
    if (getenv("LIO_ORACLE_DEBUG")) {
      double dn = 0; for (int i = 0; i < n; ++i) dn += delta[i] * delta[i];
      fprintf(stderr, "[oracle tr] it %d cost %.6g cand %.6g model %.6g rho %.4g radius %.4g mu %.3g |delta| %.4g step_norm %.4g\n", iteration, x_cost, cand_cost,
              model_cost_change, relative_decrease, radius, mu, std::sqrt(dn), step_norm);
    }
    if (relative_decrease > 1e-3) {
      // HandleSuccessfulStep
      P = cand;
      x_norm = AmbientNorm(P);
      x_cost = EvaluateProblem(P, lay, which, false, &H, &g);
      if (dump) { dump->H.push_back(H.a); dump->g.push_back(g); dump->cost.push_back(x_cost); dump->its.back().accepted = 1; }
      gmax = gradMaxNorm(g);
      scaleSystem(H, g);
      ++sum.successful;
      if (relative_decrease < 0.25) radius *= 0.5;
      if (relative_decrease > 0.75) radius = std::max(radius, 3.0 * dogleg_step_norm);
      mu = std::max(min_mu, 2.0 * mu / mu_inc);
      reuse = false;
    } else {
      radius *= 0.5; reuse = true;  // StepRejected
    }
    sum.cost_trace.push_back(x_cost);
  }
  sum.iterations = iteration;
  sum.final_cost = x_cost;
  return sum;
}

// The dense tail of MarginalizationInfo::Marginalize (MarginalizationFactor.cc:271-302) on an assembled (A, b) whose first m
// rows / columns are the dropped blocks: Amm^+ by eigen-decomposition with eigenvalues <= 1e-8 zeroed, Schur complement,
// second eigen-decomposition -> linearized_jacobians = sqrt(S) V^T, linearized_residuals = sqrt(S^+) V^T b.
static inline void MarginalizeSchur(const Mat &A, const std::vector<double> &b, int m, int n, Mat &lin_jac, std::vector<double> &lin_res) {
  const double eps = 1e-8;
  Mat Amm(m, m);
  for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) Amm(i, j) = 0.5 * (A(i, j) + A(j, i));
  std::vector<double> ev(m); Mat V(m, m);
  sym_eigen<double>(m, Amm.a.data(), ev.data(), V.a.data());
  Mat Amm_inv(m, m);
  for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) {
    double s = 0;
    for (int k = 0; k < m; ++k) s += V(i, k) * (ev[k] > eps ? 1.0 / ev[k] : 0.0) * V(j, k);
    Amm_inv(i, j) = s;
  }
  Mat Arm(n, m), Amr(m, n), Arr(n, n);
  std::vector<double> bmm(b.begin(), b.begin() + m), brr(b.begin() + m, b.end());
  for (int i = 0; i < n; ++i) for (int j = 0; j < m; ++j) { Arm(i, j) = A(m + i, j); Amr(j, i) = A(j, m + i); }
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Arr(i, j) = A(m + i, m + j);
  Mat T = matmul(Arm, Amm_inv);
  Mat TA = matmul(T, Amr);
  std::vector<double> Tb = matvec(T, bmm);
  Mat S(n, n); std::vector<double> bs(n);
  for (int i = 0; i < n; ++i) { bs[i] = brr[i] - Tb[i]; for (int j = 0; j < n; ++j) S(i, j) = Arr(i, j) - TA(i, j); }
  std::vector<double> ev2(n); Mat V2(n, n);
  sym_eigen<double>(n, S.a.data(), ev2.data(), V2.a.data());
  lin_jac = Mat(n, n); lin_res.assign(n, 0.0);
  for (int k = 0; k < n; ++k) {
    double Sk = ev2[k] > eps ? ev2[k] : 0.0, Sik = ev2[k] > eps ? 1.0 / ev2[k] : 0.0;
    double ss = std::sqrt(Sk), sis = std::sqrt(Sik);
    double vb = 0;
    for (int i = 0; i < n; ++i) { lin_jac(k, i) = ss * V2(i, k); vb += V2(i, k) * bs[i]; }
    lin_res[k] = sis * vb;
  }
}

// MarginalizationInfo::{PreMarginalize,Marginalize} + GetParameterBlocks in canonical order.
// Dropped: pose0 (6), sb0 (9).  Kept: pose1, sb1 (when the IMU factor 0->1 exists), pose2..poseWo, ex.
static inline std::shared_ptr<MargPrior> Marginalize(const WindowProblem &P, int threads = 4) {
  const int Wo = P.Wo;
  bool has_imu = P.use_imu && P.pim[0];
  bool sb0_present = has_imu || (P.prior != nullptr);
  Layout lay;
  lay.pose.assign(Wo + 1, -1); lay.sb.assign(Wo + 1, -1);
  int pos = 0;
  lay.pose[0] = pos; pos += 6;
  if (sb0_present) { lay.sb[0] = pos; pos += 9; }
  const int m = pos;
  std::vector<KeepBlock> keep;
  lay.pose[1] = pos; keep.push_back({0, 0, 7, pos - m}); pos += 6;
  if (has_imu) { lay.sb[1] = pos; keep.push_back({1, 0, 9, pos - m}); pos += 9; }
  for (int i = 2; i <= Wo; ++i) { lay.pose[i] = pos; keep.push_back({0, i - 1, 7, pos - m}); pos += 6; }
  lay.ex = pos; keep.push_back({2, 0, 7, pos - m}); pos += 6;
  lay.dim = pos;
  const int n = pos - m;
  // All factors are evaluated with the extrinsic as a variable here even when the solve held it
  // constant: ResidualBlockInfo knows nothing about SetParameterBlockConstant.
  WindowProblem Q = P;
  Q.ex_constant = false;
  Q.use_prior_factor = false;  // PriorFactor is not added to marginalization_info (Estimator.cc:2154-2218)
  Mat A; std::vector<double> b;
  EvaluateProblem(Q, lay, 1 | 2 | 4, true, &A, &b, nullptr, threads);
  auto pr = std::make_shared<MargPrior>();
  pr->n = n; pr->keep = keep;
  MarginalizeSchur(A, b, m, n, pr->lin_jac, pr->lin_res);
  // keep_block_data = parameter values at PreMarginalize time, re-addressed by addr_shift
  for (const KeepBlock &kb : keep) {
    const double *src;
    if (kb.kind == 0) src = P.pose[kb.index + 1].data();
    else if (kb.kind == 1) src = P.sb[kb.index + 1].data();
    else src = P.ex.data();
    pr->x0.emplace_back(src, src + kb.size);
  }
  return pr;
}

}  // namespace orc
