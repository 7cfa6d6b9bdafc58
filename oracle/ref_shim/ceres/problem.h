// Stand-ins for ceres::Problem and ceres::Solve, as far as the reference's Estimator::SolveOptimization uses them
// (Estimator.cc:1660-2020).  oracle/ref_shim: TEST INFRASTRUCTURE.
//
// What is independent here and what is not.  The PROBLEM — which parameter blocks exist, which are constant, which residual blocks
// are added with which cost function, loss and parameters, in which order — is entirely the reference's own code, and every
// residual and Jacobian is computed by the reference's own cost-function classes.  The MINIMIZER is not Ceres (a third-party
// dependency that is absent from /root/reference): it is the same restatement of Ceres 1.14's TrustRegionMinimizer +
// DoglegStrategy(TRADITIONAL_DOGLEG) as oracle/solver.h:SolveDogleg, written over generic parameter / residual blocks, with the
// same dense Cholesky (oracle/liomath.h).  The step sequence of the minimizer therefore stays UNPINNED; what a comparison of the
// reference's Estimator on these stand-ins with the oracle's estimator pins is everything around it.
// options.max_solver_time_in_seconds is ignored (a wall-clock limit cannot be compared).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>

#include "../../liomath.h"

namespace ceres {
namespace internal {
struct ResidualBlock {
  CostFunction *cost = nullptr;
  LossFunction *loss = nullptr;
  std::vector<double *> params;
};
}  // namespace internal
typedef internal::ResidualBlock *ResidualBlockId;
enum LinearSolverType { DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY, DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR, CGNR };
enum TrustRegionStrategyType { LEVENBERG_MARQUARDT, DOGLEG };
struct CRSMatrix {};

class Problem {
 public:
  struct EvaluateOptions {
    std::vector<double *> parameter_blocks;
    std::vector<ResidualBlockId> residual_blocks;
    bool apply_loss_function = true;
  };
  struct ParamBlock { double *ptr; int size; LocalParameterization *local; bool constant; };
  ~Problem() { for (internal::ResidualBlock *b : blocks_) delete b; }

  void AddParameterBlock(double *p, int size, LocalParameterization *local = nullptr) {
    if (index_.count(p)) return;
    index_[p] = int(params_.size());
    params_.push_back({p, size, local, false});
  }
  void SetParameterBlockConstant(double *p) { params_[index_.at(p)].constant = true; }
  ResidualBlockId AddResidualBlock(CostFunction *c, LossFunction *l, const std::vector<double *> &ps) {
    internal::ResidualBlock *b = new internal::ResidualBlock;
    b->cost = c; b->loss = l; b->params = ps;
    for (size_t k = 0; k < ps.size(); ++k) AddParameterBlock(ps[k], c->parameter_block_sizes()[k]);
    blocks_.push_back(b);
    return b;
  }
  template <typename... P> ResidualBlockId AddResidualBlock(CostFunction *c, LossFunction *l, double *p0, P *... rest) {
    return AddResidualBlock(c, l, std::vector<double *>{p0, rest...});
  }
  void RemoveResidualBlock(ResidualBlockId id) {
    for (size_t k = 0; k < blocks_.size(); ++k) if (blocks_[k] == id) { delete id; blocks_.erase(blocks_.begin() + k); return; }
  }
  static void evaluate_log_hook(double c);
  // cost only: that is all the reference asks of it (Estimator.cc:1931-2017)
  bool Evaluate(const EvaluateOptions &o, double *cost, std::vector<double> *, std::vector<double> *, CRSMatrix *) {
    const std::vector<ResidualBlockId> &bl = o.residual_blocks.empty() ? blocks_ : o.residual_blocks;
    double c = 0;
    for (ResidualBlockId b : bl) {
      std::vector<const double *> ps(b->params.begin(), b->params.end());
      c += EvalBlock(*b, ps.data(), o.apply_loss_function, nullptr, nullptr, nullptr);
    }
    if (cost) *cost = c;
    evaluate_log_hook(c);
    return true;
  }

  // One residual block at `ps`: returns its cost; with H / g also adds its share of J^T J and J^T r in local coordinates
  // (cols[k] = column of parameter k, -1 = constant).  The robust correction is ceres::Corrector; the order of the arithmetic is
  // that of oracle/solver.h:AccumulateBlock.
  double EvalBlock(const internal::ResidualBlock &b, const double *const *ps, bool apply_loss, const int *cols, orc::Mat *H, std::vector<double> *g) const {
    const int nres = b.cost->num_residuals();
    const std::vector<int> &sz = b.cost->parameter_block_sizes();
    const size_t np = sz.size();
    std::vector<double> r(nres);
    std::vector<std::vector<double>> J(np);
    std::vector<double *> jp(np, nullptr);
    if (H) for (size_t k = 0; k < np; ++k) { J[k].assign(size_t(nres) * sz[k], 0.0); jp[k] = J[k].data(); }
    b.cost->Evaluate(ps, r.data(), H ? jp.data() : nullptr);
    double sq = 0;
    for (int i = 0; i < nres; ++i) sq += r[i] * r[i];
    double cost = 0.5 * sq;
    if (b.loss && apply_loss) {
      double rho[3];
      b.loss->Evaluate(sq, rho);
      cost = 0.5 * rho[0];
      const double sqrt_rho1 = std::sqrt(rho[1]);
      double residual_scaling, alpha_sq_norm;
      if (sq == 0.0 || rho[2] <= 0.0) { residual_scaling = sqrt_rho1; alpha_sq_norm = 0.0; }
      else {
        const double Dd = 1.0 + 2.0 * sq * rho[2] / rho[1];
        const double alpha = 1.0 - std::sqrt(Dd);
        residual_scaling = sqrt_rho1 / (1 - alpha);
        alpha_sq_norm = alpha / sq;
      }
      if (H) {
        for (size_t k = 0; k < np; ++k) {
          const int amb = sz[k];
          std::vector<double> &Jk = J[k];
          if (alpha_sq_norm != 0.0) {
            std::vector<double> rtJ(amb, 0.0);
            for (int i = 0; i < nres; ++i) for (int c = 0; c < amb; ++c) rtJ[c] += r[i] * Jk[size_t(i) * amb + c];
            for (int i = 0; i < nres; ++i) for (int c = 0; c < amb; ++c) Jk[size_t(i) * amb + c] = sqrt_rho1 * (Jk[size_t(i) * amb + c] - alpha_sq_norm * r[i] * rtJ[c]);
          } else {
            for (double &v : Jk) v *= sqrt_rho1;
          }
        }
      }
      for (int i = 0; i < nres; ++i) r[i] *= residual_scaling;
    }
    if (!H) return cost;
    // ambient -> local Jacobians (J * ComputeJacobian(x)); stored nres x local, row-major
    std::vector<std::vector<double>> Jl(np);
    std::vector<int> loc(np);
    for (size_t k = 0; k < np; ++k) {
      const ParamBlock &pb = params_[index_.at(const_cast<double *>(b.params[k]))];
      if (!pb.local) { Jl[k] = J[k]; loc[k] = sz[k]; continue; }
      const int amb = sz[k], l = pb.local->LocalSize();
      std::vector<double> P(size_t(amb) * l);
      pb.local->ComputeJacobian(ps[k], P.data());
      Jl[k].assign(size_t(nres) * l, 0.0);
      // (a [I; 0] parameterization — the reference's PoseLocalParameterization — makes this a column selection, exactly)
      for (int i = 0; i < nres; ++i) for (int c = 0; c < l; ++c) {
        double s = 0; bool first = true;
        for (int a = 0; a < amb; ++a) { const double p = P[size_t(a) * l + c]; if (p == 0.0) continue; const double t = J[k][size_t(i) * amb + a] * p; s = first ? t : s + t; first = false; }
        Jl[k][size_t(i) * l + c] = s;
      }
      loc[k] = l;
    }
    for (size_t a = 0; a < np; ++a) {
      if (cols[a] < 0) continue;
      const double *Ja = Jl[a].data(); const int la = loc[a], ca = cols[a];
      for (size_t bb = a; bb < np; ++bb) {
        if (cols[bb] < 0) continue;
        const double *Jb = Jl[bb].data(); const int lb = loc[bb], cb = cols[bb];
        for (int i = 0; i < la; ++i)
          for (int j = 0; j < lb; ++j) {
            double s = 0;
            for (int k = 0; k < nres; ++k) s += Ja[size_t(k) * la + i] * Jb[size_t(k) * lb + j];
            (*H)(ca + i, cb + j) += s;
            if (a != bb) (*H)(cb + j, ca + i) += s;
          }
      }
      for (int i = 0; i < la; ++i) { double s = 0; for (int k = 0; k < nres; ++k) s += Ja[size_t(k) * la + i] * r[k]; (*g)[ca + i] += s; }
    }
    return cost;
  }

  std::vector<ParamBlock> params_;
  std::map<double *, int> index_;
  std::vector<internal::ResidualBlock *> blocks_;
};

struct Solver {
  struct Options {
    LinearSolverType linear_solver_type = DENSE_SCHUR;
    TrustRegionStrategyType trust_region_strategy_type = LEVENBERG_MARQUARDT;
    int max_num_iterations = 50, num_threads = 1;
    double max_solver_time_in_seconds = 1e9;
    bool minimizer_progress_to_stdout = false, use_nonmonotonic_steps = false, use_explicit_schur_complement = false;
  };
  struct Summary {
    int iterations = 0, successful = 0, termination = 0;
    double initial_cost = 0, final_cost = 0;
    std::vector<double> cost_trace;
    std::string BriefReport() const { return std::string(); }
    std::string FullReport() const { return std::string(); }
  };
};
inline Solver::Summary &last_summary() { static Solver::Summary s; return s; }
// what the test driver reads back: the residual blocks of the problem handed to the last Solve (cost functions are never freed by
// these stand-ins, so the pointers stay valid), and the cost of every Problem::Evaluate call since it was last cleared
inline std::vector<internal::ResidualBlock> &last_blocks() { static std::vector<internal::ResidualBlock> v; return v; }
inline std::vector<double> &evaluate_log() { static std::vector<double> v; return v; }
// the parameter blocks of the last Solve: address, whether constant, the values it started from and ended with
struct ParamRecord { double *ptr; bool constant; std::vector<double> initial, final_; };
inline std::vector<ParamRecord> &last_params() { static std::vector<ParamRecord> v; return v; }

inline void Problem::evaluate_log_hook(double c) { evaluate_log().push_back(c); }

// Ceres 1.14 TrustRegionMinimizer::Minimize with DoglegStrategy (TRADITIONAL_DOGLEG) — see the header of this file.
inline void Solve(const Solver::Options &opt, Problem *problem, Solver::Summary *out) {
  Problem &Q = *problem;
  last_blocks().clear();
  for (internal::ResidualBlock *b : Q.blocks_) last_blocks().push_back(*b);
  last_params().clear();
  for (const Problem::ParamBlock &pb : Q.params_) last_params().push_back({pb.ptr, pb.constant, std::vector<double>(pb.ptr, pb.ptr + pb.size), {}});
  Solver::Summary sum;
  const size_t NP = Q.params_.size();
  // the state: one ambient vector per parameter block (the user's memory is written at the end only)
  std::vector<std::vector<double>> x(NP), cand(NP);
  std::vector<int> col(NP, -1), loc(NP);
  int n = 0;
  for (size_t k = 0; k < NP; ++k) {
    const Problem::ParamBlock &pb = Q.params_[k];
    x[k].assign(pb.ptr, pb.ptr + pb.size);
    loc[k] = pb.local ? pb.local->LocalSize() : pb.size;
    if (!pb.constant) { col[k] = n; n += loc[k]; }
  }
  auto evaluate = [&](const std::vector<std::vector<double>> &at, orc::Mat *H, std::vector<double> *g) {
    if (H) { *H = orc::Mat(n, n); g->assign(n, 0.0); }
    double cost = 0;
    for (internal::ResidualBlock *b : Q.blocks_) {
      const size_t np = b->params.size();
      std::vector<const double *> ps(np);
      std::vector<int> cols(np);
      for (size_t k = 0; k < np; ++k) { const int id = Q.index_.at(b->params[k]); ps[k] = at[id].data(); cols[k] = col[id]; }
      cost += Q.EvalBlock(*b, ps.data(), true, cols.data(), H, g);
    }
    return cost;
  };
  auto plus = [&](const std::vector<std::vector<double>> &at, const std::vector<double> &delta, std::vector<std::vector<double>> &res) {
    res = at;
    for (size_t k = 0; k < NP; ++k) {
      if (col[k] < 0) continue;
      const Problem::ParamBlock &pb = Q.params_[k];
      if (pb.local) pb.local->Plus(at[k].data(), &delta[col[k]], res[k].data());
      else for (int i = 0; i < pb.size; ++i) res[k][i] = at[k][i] + delta[col[k] + i];
    }
  };
  auto ambientNorm = [&](const std::vector<std::vector<double>> &a, const std::vector<std::vector<double>> *b) {
    double s = 0;
    for (size_t k = 0; k < NP; ++k) {
      if (col[k] < 0) continue;
      for (size_t i = 0; i < a[k].size(); ++i) { const double d = b ? a[k][i] - (*b)[k][i] : a[k][i]; s += d * d; }
    }
    return std::sqrt(s);
  };
  orc::Mat H; std::vector<double> g;
  double x_cost = evaluate(x, &H, &g);
  sum.initial_cost = x_cost; sum.cost_trace.push_back(x_cost);
  if (const char *dump = std::getenv("REF_SHIM_DUMP_HG")) {   // debugging aid: the first linearisation of every solve, appended
    if (FILE *f = std::fopen(dump, "ab")) { double nn = n; std::fwrite(&nn, 8, 1, f); std::fwrite(H.a.data(), 8, H.a.size(), f); std::fwrite(g.data(), 8, g.size(), f); std::fclose(f); }
  }
  std::vector<double> scale(n);
  for (int i = 0; i < n; ++i) scale[i] = 1.0 / (1.0 + std::sqrt(H(i, i)));
  auto scaleSystem = [&](orc::Mat &Hs, std::vector<double> &gs) {
    for (int i = 0; i < n; ++i) { gs[i] *= scale[i]; for (int j = 0; j < n; ++j) Hs(i, j) *= scale[i] * scale[j]; }
  };
  auto gradMaxNorm = [&](const std::vector<double> &g_unscaled) {
    std::vector<double> neg(n);
    for (int i = 0; i < n; ++i) neg[i] = -g_unscaled[i];
    std::vector<std::vector<double>> xp; plus(x, neg, xp);
    double mx = 0;
    for (size_t k = 0; k < NP; ++k) { if (col[k] < 0) continue; for (size_t i = 0; i < x[k].size(); ++i) mx = std::max(mx, std::fabs(x[k][i] - xp[k][i])); }
    return mx;
  };
  double gmax = gradMaxNorm(g);
  scaleSystem(H, g);
  double x_norm = ambientNorm(x, nullptr);
  double radius = 1e4, mu = 1e-8;
  const double min_mu = 1e-8, max_mu = 1.0, mu_inc = 10.0, min_diag = 1e-6, max_diag = 1e32;
  bool reuse = false;
  std::vector<double> diagonal(n), gradient(n), gn(n);
  double alpha = 0, dogleg_step_norm = 0;
  int consecutive_invalid = 0, iteration = 0;
  while (true) {
    if (iteration >= opt.max_num_iterations) { sum.termination = 0; break; }
    if (gmax <= 1e-10) { sum.termination = 3; break; }
    if (radius <= 1e-32) { sum.termination = 1; break; }
    ++iteration;
    bool linear_ok = true;
    if (!reuse) {
      reuse = true;
      for (int i = 0; i < n; ++i) diagonal[i] = std::sqrt(std::min(std::max(H(i, i), min_diag), max_diag));
      for (int i = 0; i < n; ++i) gradient[i] = g[i] / diagonal[i];
      std::vector<double> sg(n);
      for (int i = 0; i < n; ++i) sg[i] = gradient[i] / diagonal[i];
      std::vector<double> Hsg = orc::matvec(H, sg);
      double Jg2 = 0, g2 = 0;
      for (int i = 0; i < n; ++i) { Jg2 += sg[i] * Hsg[i]; g2 += gradient[i] * gradient[i]; }
      alpha = g2 / Jg2;
      linear_ok = false;
      while (mu < max_mu) {
        orc::Mat A = H;
        for (int i = 0; i < n; ++i) A(i, i) += diagonal[i] * diagonal[i] * mu;
        orc::Mat L;
        bool ok = orc::cholesky(A, L);
        if (ok) {
          gn = g;
          orc::chol_solve(L, gn);
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
        const double beta = (c <= 0) ? (d - c) / bma_sq : (radius * radius - a_sq) / (d + c);
        double sn = 0;
        for (int i = 0; i < n; ++i) { step[i] = (-alpha * (1.0 - beta)) * gradient[i] + beta * gn[i]; sn += step[i] * step[i]; }
        dogleg_step_norm = std::sqrt(sn);
      }
      for (int i = 0; i < n; ++i) step[i] /= diagonal[i];
      std::vector<double> Hs = orc::matvec(H, step);
      double sg = 0, sHs = 0;
      for (int i = 0; i < n; ++i) { sg += step[i] * g[i]; sHs += step[i] * Hs[i]; }
      model_cost_change = -(sg + 0.5 * sHs);
      if (!(model_cost_change > 0)) step_valid = false;
    }
    if (!step_valid) {
      if (++consecutive_invalid >= 5) { sum.termination = 5; break; }
      mu *= mu_inc; reuse = false;
      sum.cost_trace.push_back(x_cost);
      continue;
    }
    consecutive_invalid = 0;
    std::vector<double> delta(n);
    for (int i = 0; i < n; ++i) delta[i] = step[i] * scale[i];
    plus(x, delta, cand);
    const double cand_cost = evaluate(cand, nullptr, nullptr);
    const double step_norm = ambientNorm(x, &cand);
    if (step_norm <= 1e-8 * (x_norm + 1e-8)) { sum.termination = 1; sum.cost_trace.push_back(x_cost); break; }
    const double cost_change = x_cost - cand_cost;
    if (std::fabs(cost_change) <= 1e-6 * x_cost) { sum.termination = 2; sum.cost_trace.push_back(x_cost); break; }
    const double relative_decrease = cost_change / model_cost_change;
    if (relative_decrease > 1e-3) {
      x = cand;
      x_norm = ambientNorm(x, nullptr);
      x_cost = evaluate(x, &H, &g);
      gmax = gradMaxNorm(g);
      scaleSystem(H, g);
      ++sum.successful;
      if (relative_decrease < 0.25) radius *= 0.5;
      if (relative_decrease > 0.75) radius = std::max(radius, 3.0 * dogleg_step_norm);
      mu = std::max(min_mu, 2.0 * mu / mu_inc);
      reuse = false;
    } else {
      radius *= 0.5; reuse = true;
    }
    sum.cost_trace.push_back(x_cost);
  }
  sum.iterations = iteration;
  sum.final_cost = x_cost;
  for (size_t k = 0; k < NP; ++k) if (col[k] >= 0) std::copy(x[k].begin(), x[k].end(), Q.params_[k].ptr);
  for (ParamRecord &r : last_params()) r.final_.assign(r.ptr, r.ptr + r.initial.size());
  last_summary() = sum;
  if (out) *out = sum;
}
}  // namespace ceres
