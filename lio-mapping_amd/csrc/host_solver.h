// host_solver.h — host half of Estimator::SolveOptimization (Estimator.cc:1648-2438) in the product.
//
// The GPU returns, per optimised frame i, the 16x16 moment matrix S_i = sum rho' z z^T and the robust
// cost (solve_kernels.hip).  Here:
//   * L_i (18x13) / l_i (13) are read off ppp_factor at basis inputs, H_i = L S L^T, g_i = L S l;
//   * the Wo IMU factors, the marginalization prior and the extrinsic prior are added on the host
//     (a few 15x30 blocks: microseconds);
//   * the trust-region loop is Ceres 1.14's TrustRegionMinimizer with the traditional dogleg strategy
//     and Jacobi scaling, working from (H, g) only — every quantity Ceres forms from J (column norms,
//     ||J v||^2, model cost change) is a quadratic form in H;
//   * Marginalize(): Schur complement + sqrt factor, MarginalizationFactor.cc:185-311, canonical block
//     order.
#pragma once
#include <array>
#include <chrono>
#include <functional>
#include <memory>

#include "host_factors.h"
#include "solve_kernels.h"

namespace lio {

struct KeepBlock { int kind; int index; int size; int idx; };  // kind 0 pose, 1 speed-bias, 2 extrinsic

struct MargPrior {
  int n = 0;
  std::vector<KeepBlock> keep;
  std::vector<std::vector<double>> x0;
  DMat lin_jac;                 // n x n
  std::vector<double> lin_res;  // n
  DMat JtJ;                     // cached lin_jac^T lin_jac
  std::vector<double> Jtr0;     // cached lin_jac^T lin_res
  // A prior computed by a batched solve (est_batch.h) lives on the device — where the next batched solve reads it — until a
  // host-side reader asks for it: materialize() downloads lin_jac / lin_res / JtJ / Jtr0 through `fetch`, once.
  bool on_device = false;
  std::function<void(MargPrior &)> fetch;
  void materialize() {
    if (on_device && fetch) fetch(*this);
    on_device = false; fetch = nullptr;
  }
  void finalize() {
    JtJ = DMat(n, n); Jtr0.assign(n, 0.0);
    // sum of n outer products of the rows of lin_jac: contiguous inner loops
    for (int k = 0; k < n; ++k) {
      const double *jr = &lin_jac.a[size_t(k) * n];
      const double rk = lin_res[k];
      for (int i = 0; i < n; ++i) {
        const double f = jr[i];
        double *hr = &JtJ.a[size_t(i) * n];
        for (int j = 0; j < n; ++j) hr[j] += f * jr[j];
        Jtr0[i] += f * rk;
      }
    }
  }
};

struct WindowParams {
  int Wo = 0;
  std::vector<std::array<double, 7>> pose;
  std::vector<std::array<double, 9>> sb;
  std::array<double, 7> ex{};
  bool ex_constant = true;
};

struct Layout {
  std::vector<int> pose, sb;
  int ex = -1, dim = 0;
};

struct FrameMoments { double S[256]; double cost; double count; };

// T_{pivot<-i} in fp64 from the parameter blocks (PivotPointPlaneFactor.cc:58-70)
LIO_HD void relative_lidar_pose(const double *pose_p, const double *pose_i, const double *pose_ex, double R[9], double t[3]) {
  V3d Pp, Pi, tlb; Qd Qp, Qi, qlb;
  unpack_pose(pose_p, Pp, Qp); unpack_pose(pose_i, Pi, Qi); unpack_pose(pose_ex, tlb, qlb);
  Qd Qlp = Qp * conj(qlb);
  V3d Plp = Pp - rotate(Qlp, tlb);
  Qd Qli = Qi * conj(qlb);
  V3d Pli = Pi - rotate(Qli, tlb);
  Qd Qlpi = conj(Qlp) * Qli;
  V3d Plpi = rotate(conj(Qlp), Pli - Plp);
  // rotate(q, v) for a (near-)unit q equals toRot(q) v up to rounding; the residual kernel needs a matrix.
  // Build it column by column from rotate() so it is the SAME map the factor applies.
  V3d c0 = rotate(Qlpi, V3d(1, 0, 0)), c1 = rotate(Qlpi, V3d(0, 1, 0)), c2 = rotate(Qlpi, V3d(0, 0, 1));
  R[0] = c0.x; R[1] = c1.x; R[2] = c2.x; R[3] = c0.y; R[4] = c1.y; R[5] = c2.y; R[6] = c0.z; R[7] = c1.z; R[8] = c2.z;
  t[0] = Plpi.x; t[1] = Plpi.y; t[2] = Plpi.z;
}

// L (18 x 13, row-major) and l (13): j = L z, r = l^T z, z = [w0*(p,1), w1*(p,1), w2*(p,1), d].
// The factor is linear in the plane normal w and affine in the point p, so L is read off the factor's own Jacobian
// expressions (ppp_eval) at p in {0, e0, e1, e2}; a unit normal e_a just selects row a of the 3x3 matrices those
// expressions multiply w with, so the four matrices per block are formed once per p instead of once per (p, a).
// Split in three so that the device can evaluate the four probe points on four lanes (solve_step.h): the pose-only part,
// one probe point, the combination.  lidar_linear_maps() below runs them in sequence (same arithmetic either way).
struct LidarMapPrep { PppPoses t; M3d A, ARi, RpTRi, Bx, S1, S2; };
LIO_HD LidarMapPrep lidar_map_prepare(const double *pose_p, const double *pose_i, const double *pose_ex) {
  LidarMapPrep m;
  m.t = ppp_prepare(pose_p, pose_i, pose_ex);
  const PppPoses &t = m.t;
  m.A = t.rlb * t.RpT;                 // translation blocks: pivot -w^T A, frame i  w^T A
  m.ARi = m.A * t.Ri;
  m.RpTRi = t.RpT * t.Ri;
  m.Bx = M3d::identity() - (t.rlb * m.RpTRi) * t.rlbT;
  m.S1 = skew(t.RpT * t.dP); m.S2 = skew(t.rlbT * t.tlb);
  return m;
}
// probe point b in {0: origin, 1..3: unit axes}: J[normal axis][column] (3 x 18) and r[normal axis]
LIO_HD void lidar_map_probe(const LidarMapPrep &m, int b, double *J /* 3 x 18 */, double *r /* 3 */) {
  const PppPoses &t = m.t;
  const V3d p(b == 1, b == 2, b == 3);
  const V3d q = t.rlbT * (p - t.tlb);
  const M3d Mp = t.rlb * (skew(t.RpT * (t.Ri * q)) + m.S1);
  const M3d Mi = m.ARi * (-skew(t.rlbT * p) + m.S2);
  const M3d Mx = t.rlb * (-skew(m.RpTRi * q) + m.RpTRi * skew(q) - m.S1);
  const V3d rv = rotate(t.Qlpi, p) + t.Plpi;
  for (int a = 0; a < 3; ++a) {
    for (int k = 0; k < 3; ++k) {
      J[a * 18 + k] = -m.A(a, k); J[a * 18 + 3 + k] = Mp(a, k);
      J[a * 18 + 6 + k] = m.A(a, k); J[a * 18 + 9 + k] = Mi(a, k);
      J[a * 18 + 12 + k] = m.Bx(a, k); J[a * 18 + 15 + k] = Mx(a, k);
    }
    r[a] = rv[a];
  }
}
// entry `e` of the packed output [L (18 x 13) | l (13)] from the four probes: J4 = [probe][axis][column], r4 = [probe][axis]
LIO_HD double lidar_map_entry(const double *J4 /* 4 x 3 x 18 */, const double *r4 /* 4 x 3 */, int e) {
  if (e < 18 * 13) {
    const int k = e / 13, c = e % 13;
    if (c == 12) return 0.0;
    const int a = c >> 2, b = c & 3;
    if (b == 3) return J4[(0 * 3 + a) * 18 + k];
    return J4[((b + 1) * 3 + a) * 18 + k] - J4[(0 * 3 + a) * 18 + k];
  }
  const int c = e - 18 * 13;
  if (c == 12) return 1.0;
  const int a = c >> 2, b = c & 3;
  if (b == 3) return r4[0 * 3 + a];
  return r4[(b + 1) * 3 + a] - r4[0 * 3 + a];
}
LIO_HD void lidar_linear_maps(const double *pose_p, const double *pose_i, const double *pose_ex, double L[18 * 13], double l[13]) {
  const LidarMapPrep m = lidar_map_prepare(pose_p, pose_i, pose_ex);
  double J4[4 * 3 * 18], r4[4 * 3];
  for (int b = 0; b < 4; ++b) lidar_map_probe(m, b, J4 + b * 54, r4 + b * 3);
  // the combination of lidar_map_entry, as plain loops (no division per entry): column 4 a + b of row k is the probe difference
  // (b < 3) or the origin probe (b == 3), column 12 is zero; l likewise with a trailing one
  for (int k = 0; k < 18; ++k) {
    double *row = L + k * 13;
    for (int a = 0; a < 3; ++a) {
      const double o = J4[(0 * 3 + a) * 18 + k];
      row[4 * a + 0] = J4[(1 * 3 + a) * 18 + k] - o;
      row[4 * a + 1] = J4[(2 * 3 + a) * 18 + k] - o;
      row[4 * a + 2] = J4[(3 * 3 + a) * 18 + k] - o;
      row[4 * a + 3] = o;
    }
    row[12] = 0.0;
  }
  for (int a = 0; a < 3; ++a) {
    const double o = r4[0 * 3 + a];
    l[4 * a + 0] = r4[1 * 3 + a] - o; l[4 * a + 1] = r4[2 * 3 + a] - o; l[4 * a + 2] = r4[3 * 3 + a] - o; l[4 * a + 3] = o;
  }
  l[12] = 1.0;
}

// One frame's lidar block of the normal equations from its moments: [Hb | gb] = (L S) [L^T | l], with
//   L   18 x 13 row-major (lidar_linear_maps),
//   Lt  13 x LIO_LT_LD: row k = column k of L, then l[k] in column 18, zero padding — so gb falls out as column 18 of the product,
//   S   16 x 16 row-major moments (13 x 13 used; the padding never meets a non-zero factor),
//   HG  18 x LIO_LT_LD: Hb in columns 0..17, gb in column 18.
// Both products run over padded rows of 16 / 24 doubles: whole vectors, no remainder loops (the 13- and 18-wide loops of the
// straightforward form ran at ~2 multiply-adds per cycle: 1 us per frame on the critical path of every linearisation).
#define LIO_LT_LD 24
inline void lidar_block_portable(const double *L, const double *Lt, const double *S, double *HG) {
  for (int a = 0; a < 18; ++a) {
    double ls[16];
    for (int b = 0; b < 16; ++b) ls[b] = 0.0;
    for (int k = 0; k < 13; ++k) { const double f = L[a * 13 + k]; const double *sr = S + k * 16; for (int b = 0; b < 16; ++b) ls[b] += f * sr[b]; }
    double *o = HG + a * LIO_LT_LD;
    for (int b = 0; b < LIO_LT_LD; ++b) o[b] = 0.0;
    for (int k = 0; k < 13; ++k) { const double f = ls[k]; const double *lr = Lt + k * LIO_LT_LD; for (int b = 0; b < LIO_LT_LD; ++b) o[b] += f * lr[b]; }
  }
}
__attribute__((target("avx512f,fma"))) inline void lidar_block_avx512(const double *L, const double *Lt, const double *S, double *HG) {
  for (int a = 0; a < 18; ++a) {
    __m512d s0 = _mm512_setzero_pd(), s1 = _mm512_setzero_pd();
    for (int k = 0; k < 13; ++k) {
      const __m512d f = _mm512_set1_pd(L[a * 13 + k]);
      s0 = _mm512_fmadd_pd(f, _mm512_loadu_pd(S + k * 16), s0);
      s1 = _mm512_fmadd_pd(f, _mm512_loadu_pd(S + k * 16 + 8), s1);
    }
    alignas(64) double ls[16];
    _mm512_store_pd(ls, s0); _mm512_store_pd(ls + 8, s1);
    __m512d h0 = _mm512_setzero_pd(), h1 = _mm512_setzero_pd(), h2 = _mm512_setzero_pd();
    for (int k = 0; k < 13; ++k) {
      const __m512d f = _mm512_set1_pd(ls[k]);
      const double *lr = Lt + k * LIO_LT_LD;
      h0 = _mm512_fmadd_pd(f, _mm512_loadu_pd(lr), h0);
      h1 = _mm512_fmadd_pd(f, _mm512_loadu_pd(lr + 8), h1);
      h2 = _mm512_fmadd_pd(f, _mm512_loadu_pd(lr + 16), h2);
    }
    double *o = HG + a * LIO_LT_LD;
    _mm512_storeu_pd(o, h0); _mm512_storeu_pd(o + 8, h1); _mm512_storeu_pd(o + 16, h2);
  }
}
inline void lidar_block(const double *L, const double *Lt, const double *S, double *HG) {
  static const bool has512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("fma");
  if (has512) lidar_block_avx512(L, Lt, S, HG); else lidar_block_portable(L, Lt, S, HG);
}

class WindowSystem {
 public:
  int Wo = 0;
  std::vector<std::shared_ptr<Preintegration>> pim;  // [i] links opt i -> i+1 (null = skipped)
  std::shared_ptr<MargPrior> prior;
  bool use_prior_factor = false, use_lidar = true;
  V3d prior_pos; Qd prior_rot;
  // lidar evaluation on the device: fills m[1..Wo] for the given parameters
  std::function<void(const WindowParams &, std::vector<FrameMoments> &)> lidar_eval;
  // split form: launch the device pass first, overlap the host-side prior / IMU factors with it, then collect
  std::function<void(const WindowParams &)> lidar_launch;
  std::function<void(std::vector<FrameMoments> &)> lidar_wait;
  // optional: frame i's moments as soon as THEY are in (the resident kernel posts a completion word per frame), so that the
  // frame blocks of the early frames are expanded while the late ones are still on their way; false = not available for this
  // pass, take lidar_wait
  std::function<bool(int, FrameMoments &)> lidar_wait_frame;
  // moments already known for the parameters of the NEXT evaluate call (consumed by it): the marginalization linearises
  // at the point the solver stopped at, whose lidar moments its last accepted step computed — they depend only on the relative
  // poses T_{pivot<-i} and the extrinsic, which the yaw re-anchoring of DoubleToVector leaves unchanged
  const std::vector<FrameMoments> *preset_moments = nullptr;
  // called inside evaluate() with the prior + IMU part of (H, g) — complete for the speed-bias rows, which no lidar factor
  // touches — right before the host blocks on the device pass: the solver factors the speed-bias block there (SplitFactor)
  std::function<void(const DMat &, const std::vector<double> &)> static_part_hook;
  // marginalize(): when set, the dense tail (Amm^+, Schur complement, both eigendecompositions, square-root factors) is handed
  // to it — the device path (marg_kernels.h).  (A, b, m, n, eps, lin_jac n x n, lin_res n) -> false: not handled, run the host code.
  std::function<bool(const double *, const double *, int, int, double, double *, double *)> marg_schur_hook;

  struct Costs { double marg = 0, pim = 0, ppp = 0, prior = 0; double total() const { return marg + pim + ppp + prior; } };

  static Layout solve_layout(const WindowParams &P) {
    Layout l;
    l.pose.resize(P.Wo + 1); l.sb.resize(P.Wo + 1);
    for (int i = 0; i <= P.Wo; ++i) { l.pose[i] = 15 * i; l.sb[i] = 15 * i + 6; }
    l.ex = P.ex_constant ? -1 : 15 * (P.Wo + 1);
    l.dim = 15 * (P.Wo + 1) + (P.ex_constant ? 0 : 6);
    return l;
  }

  static void add_block(DMat &H, std::vector<double> &g, const int *cols, const int *sizes, int nb, const double *Hb, const double *gb, int ld) {
    // Hb: (sum sizes)^2 dense, gb: sum sizes; blocks with col < 0 are constants
    int off_a = 0;
    for (int a = 0; a < nb; off_a += sizes[a], ++a) {
      if (cols[a] < 0) continue;
      int off_b = 0;
      for (int b = 0; b < nb; off_b += sizes[b], ++b) {
        if (cols[b] < 0) continue;
        for (int i = 0; i < sizes[a]; ++i)
          for (int j = 0; j < sizes[b]; ++j) H(cols[a] + i, cols[b] + j) += Hb[(off_a + i) * ld + off_b + j];
      }
      for (int i = 0; i < sizes[a]; ++i) g[cols[a] + i] += gb[off_a + i];
    }
  }

  // prior residual r = r0 + J0 dx  (MarginalizationFactor.cc:343-393); returns dx too
  static void prior_dx(const MargPrior &pr, const WindowParams &P, std::vector<double> &dx) {
    dx.assign(pr.n, 0.0);
    for (size_t b = 0; b < pr.keep.size(); ++b) {
      const KeepBlock &kb = pr.keep[b];
      const double *x = kb.kind == 0 ? P.pose[kb.index].data() : (kb.kind == 1 ? P.sb[kb.index].data() : P.ex.data());
      const double *x0 = pr.x0[b].data();
      if (kb.size != 7) { for (int k = 0; k < kb.size; ++k) dx[kb.idx + k] = x[k] - x0[k]; }
      else {
        for (int k = 0; k < 3; ++k) dx[kb.idx + k] = x[k] - x0[k];
        Qd q0(x0[6], x0[3], x0[4], x0[5]), q(x[6], x[3], x[4], x[5]);
        Qd dq = qinverse(q0) * q;
        V3d v = 2.0 * normalized(dq).vec();
        if (dq.w < 0) v = -v;
        for (int k = 0; k < 3; ++k) dx[kb.idx + 3 + k] = v[k];
      }
    }
  }

  // host-side phase clock of evaluate() (printed by the estimator under LIO_DEBUG_TIMING)
  struct EvalClock { double launch = 0, prior = 0, imu = 0, wait = 0, assemble = 0; int n = 0; };
  EvalClock eclk;
  static constexpr int LMAP_STRIDE = 18 * 13 + 13 * LIO_LT_LD;
  std::vector<double> lmaps_;                    // per frame: L (18 x 13) and [L^T | l] (13 x LIO_LT_LD) of the current evaluate() call
  std::vector<double> prior_scratch_;            // the prior's residual and gradient (no allocation per linearisation)
  // the prior's J^T J in the solve's layout (evaluate()).  The cache is keyed by the MargPrior's address, the layout's dimension and
  // its extrinsic column: valid because a WindowSystem lives for ONE solve (Estimator::SolveOptimizationHost builds it on the
  // stack, the marginalization builds its own), holds `prior` as a shared_ptr for that whole time (no other prior can take its
  // address meanwhile) and a MargPrior is immutable once built (lio_est_set_prior_factor installs a copy, never edits in place).
  // A WindowSystem that outlived its solve would have to clear prior_base_for_ whenever `prior` is assigned.
  DMat prior_base_;
  const MargPrior *prior_base_for_ = nullptr;
  int prior_base_ex_ = -2;
  std::vector<FrameMoments> moments_scratch_;    // landing zone of the device pass (no allocation per linearisation)
  static double clk_now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

  // which: bit0 prior, bit1 imu, bit2 lidar, bit3 extrinsic prior.  H/g may be null (cost only).
  Costs evaluate(const WindowParams &P, const Layout &lay, int which, bool imu_only_first, DMat *H, std::vector<double> *g,
                 std::vector<FrameMoments> *m_out = nullptr) {
    Costs c;
    const bool lidar_on = (which & 4) && use_lidar;
    const std::vector<FrameMoments> *preset = preset_moments;
    preset_moments = nullptr;
    if (preset && int(preset->size()) != Wo + 1) preset = nullptr;
    const bool split = lidar_on && !preset && lidar_launch && lidar_wait;
    double tk0 = clk_now();
    if (split) lidar_launch(P);  // asynchronous: the kernels run while the host evaluates the prior and the IMU factors
    { const double t = clk_now(); eclk.launch += t - tk0; tk0 = t; ++eclk.n; }
    // The prior's J^T J does not move during a solve: scattered into the solve's layout ONCE (prior_base_), every linearisation then
    // starts from a copy of that matrix instead of a cleared one plus 49 block additions.
    const bool prior_on = (which & 1) && prior;
    bool from_base = false;
    if (H) {
      if (!(H->r == lay.dim && H->c == lay.dim)) *H = DMat(lay.dim, lay.dim);  // (the dogleg loop recycles two buffers)
      if (prior_on) {
        if (prior_base_for_ != prior.get() || prior_base_.r != lay.dim || prior_base_ex_ != lay.ex) {
          const MargPrior &pr = *prior;
          prior_base_ = DMat(lay.dim, lay.dim);
          for (size_t a = 0; a < pr.keep.size(); ++a) {
            const KeepBlock &ka = pr.keep[a];
            const int ca = ka.kind == 0 ? lay.pose[ka.index] : (ka.kind == 1 ? lay.sb[ka.index] : lay.ex);
            if (ca < 0) continue;
            const int la = ka.size == 7 ? 6 : ka.size;
            for (size_t b = 0; b < pr.keep.size(); ++b) {
              const KeepBlock &kb = pr.keep[b];
              const int cb = kb.kind == 0 ? lay.pose[kb.index] : (kb.kind == 1 ? lay.sb[kb.index] : lay.ex);
              if (cb < 0) continue;
              const int lb = kb.size == 7 ? 6 : kb.size;
              for (int i = 0; i < la; ++i) for (int j = 0; j < lb; ++j) prior_base_(ca + i, cb + j) += pr.JtJ(ka.idx + i, kb.idx + j);
            }
          }
          prior_base_for_ = prior.get(); prior_base_ex_ = lay.ex;
        }
        std::memcpy(H->a.data(), prior_base_.a.data(), sizeof(double) * size_t(lay.dim) * lay.dim);
        from_base = true;
      } else {
        H->zero();
      }
      g->assign(lay.dim, 0.0);
    }
    if (prior_on) {
      const MargPrior &pr = *prior;
      std::vector<double> dx;
      prior_dx(pr, P, dx);
      // r = r0 + J0 dx;  cost = 0.5 |r|^2;  J^T r = Jtr0 + JtJ dx
      double cost = 0;
      prior_scratch_.resize(2 * size_t(pr.n));
      double *rb = prior_scratch_.data(), *gb = rb + pr.n;
      affine_matvec(pr.lin_jac.a.data(), pr.n, pr.n, dx.data(), pr.lin_res.data(), rb);
      for (int i = 0; i < pr.n; ++i) cost += rb[i] * rb[i];
      c.marg = 0.5 * cost;
      if (H) {
        affine_matvec(pr.JtJ.a.data(), pr.n, pr.n, dx.data(), pr.Jtr0.data(), gb);
        for (size_t a = 0; a < pr.keep.size(); ++a) {
          const KeepBlock &ka = pr.keep[a];
          int ca = ka.kind == 0 ? lay.pose[ka.index] : (ka.kind == 1 ? lay.sb[ka.index] : lay.ex);
          if (ca < 0) continue;
          int la = ka.size == 7 ? 6 : ka.size;
          for (size_t b = 0; b < pr.keep.size(); ++b) {
            const KeepBlock &kb = pr.keep[b];
            int cb = kb.kind == 0 ? lay.pose[kb.index] : (kb.kind == 1 ? lay.sb[kb.index] : lay.ex);
            if (cb < 0) continue;
            int lb = kb.size == 7 ? 6 : kb.size;
            if (!from_base) for (int i = 0; i < la; ++i) for (int j = 0; j < lb; ++j) (*H)(ca + i, cb + j) += pr.JtJ(ka.idx + i, kb.idx + j);
          }
          for (int i = 0; i < la; ++i) (*g)[ca + i] += gb[ka.idx + i];
        }
      }
    }
    { const double t = clk_now(); eclk.prior += t - tk0; tk0 = t; }
    if (which & 2) {
      int last = imu_only_first ? 1 : Wo;
      for (int i = 0; i < last; ++i) {
        if (!pim[i]) continue;
        if (H && host_has_avx512()) {
          // one block per factor: raw Jacobian once, whitening and J^T J in vector registers (host_factors.h: imu_block_avx512)
          const double *Sq = pim[i]->sqrt_info();
          if (Sq) {
            alignas(64) double Jl[15 * LIO_IMU_LD], Hb[30 * LIO_IMU_LD], gb[LIO_IMU_LD];
            double r0[15], rw[15];
            imu_raw_local(pim[i]->core(), P.pose[i].data(), P.sb[i].data(), P.pose[i + 1].data(), P.sb[i + 1].data(), r0, Jl);
            imu_block_avx512(Sq, Jl, r0, Hb, gb, rw);
            double cost = 0;
            for (int k = 0; k < 15; ++k) cost += rw[k] * rw[k];
            c.pim += 0.5 * cost;
            int cols[4] = {lay.pose[i], lay.sb[i], lay.pose[i + 1], lay.sb[i + 1]}, sizes[4] = {6, 9, 6, 9};
            add_block(*H, *g, cols, sizes, 4, Hb, gb, LIO_IMU_LD);
            continue;
          }
        }
        double r[15], J0[105], J1[135], J2[105], J3[135];
        imu_factor(*pim[i], P.pose[i].data(), P.sb[i].data(), P.pose[i + 1].data(), P.sb[i + 1].data(), r, H ? J0 : nullptr, H ? J1 : nullptr,
                   H ? J2 : nullptr, H ? J3 : nullptr);
        double cost = 0;
        for (int k = 0; k < 15; ++k) cost += r[k] * r[k];
        c.pim += 0.5 * cost;
        if (H) {
          double J[15 * 30];
          for (int k = 0; k < 15; ++k) {
            for (int q = 0; q < 6; ++q) { J[k * 30 + q] = J0[k * 7 + q]; J[k * 30 + 15 + q] = J2[k * 7 + q]; }
            for (int q = 0; q < 9; ++q) { J[k * 30 + 6 + q] = J1[k * 9 + q]; J[k * 30 + 21 + q] = J3[k * 9 + q]; }
          }
          double Hb[900], gb[30];
          // J^T J as a sum of 15 outer products: contiguous inner loops (vectorise without reassociation)
          for (int a = 0; a < 900; ++a) Hb[a] = 0.0;
          for (int a = 0; a < 30; ++a) gb[a] = 0.0;
          for (int k = 0; k < 15; ++k) {
            const double *jr = J + k * 30;
            for (int a = 0; a < 30; ++a) {
              const double f = jr[a];
              double *hr = Hb + a * 30;
              for (int b = 0; b < 30; ++b) hr[b] += f * jr[b];
              gb[a] += f * r[k];
            }
          }
          int cols[4] = {lay.pose[i], lay.sb[i], lay.pose[i + 1], lay.sb[i + 1]}, sizes[4] = {6, 9, 6, 9};
          add_block(*H, *g, cols, sizes, 4, Hb, gb, 30);
        }
      }
    }
    if (H && static_part_hook) static_part_hook(*H, *g);
    // (the extrinsic prior is evaluated here, ahead of the wait, like everything else that does not need the lidar moments)
    if ((which & 8) && use_prior_factor) {
      double r[6], J[42];
      prior_factor(prior_pos, prior_rot, P.ex.data(), r, H ? J : nullptr);
      double cost = 0;
      for (int k = 0; k < 6; ++k) cost += r[k] * r[k];
      c.prior = 0.5 * cost;
      if (H && lay.ex >= 0) {
        double Hb[36], gb[6];
        for (int a = 0; a < 6; ++a) {
          for (int b = 0; b < 6; ++b) { double s = 0; for (int k = 0; k < 6; ++k) s += J[k * 7 + a] * J[k * 7 + b]; Hb[a * 6 + b] = s; }
          double s = 0; for (int k = 0; k < 6; ++k) s += J[k * 7 + a] * r[k];
          gb[a] = s;
        }
        int cols[1] = {lay.ex}, sizes[1] = {6};
        add_block(*H, *g, cols, sizes, 1, Hb, gb, 6);
      }
    }
    // The 18 x 13 linear maps of the frames' lidar factors depend on the poses only: they are formed HERE, while the device
    // pass is still in flight, instead of behind the wait (0.6 us per frame off the critical path of every linearisation).
    const bool lidar_h = lidar_on && (preset || split || lidar_eval) && H;
    if (lidar_h) {
      lmaps_.assign(size_t(Wo + 1) * LMAP_STRIDE, 0.0);
      for (int i = 1; i <= Wo; ++i) {
        double *L = &lmaps_[size_t(i) * LMAP_STRIDE], *Lt = L + 18 * 13;
        double l[13];
        lidar_linear_maps(P.pose[0].data(), P.pose[i].data(), P.ex.data(), L, l);
        for (int k = 0; k < 13; ++k) {
          for (int a = 0; a < 18; ++a) Lt[k * LIO_LT_LD + a] = L[a * 13 + k];   // [L^T | l | 0]: see lidar_block
          Lt[k * LIO_LT_LD + 18] = l[k];
        }
      }
    }
    { const double t = clk_now(); eclk.imu += t - tk0; tk0 = t; }
    if (lidar_on && (preset || split || lidar_eval)) {
      std::vector<FrameMoments> &m = m_out ? *m_out : moments_scratch_;   // the device pass lands in the caller's vector: no copy
      m.resize(Wo + 1);
      bool per_frame = split && !preset && lidar_wait_frame && lidar_wait_frame(1, m[1]);
      if (preset) m = *preset; else if (per_frame) {} else if (split) lidar_wait(m); else lidar_eval(P, m);
      { const double t = clk_now(); eclk.wait += t - tk0; tk0 = t; }
      for (int i = 1; i <= Wo; ++i) {
        if (per_frame && i > 1) {
          const double tw = clk_now();
          if (!lidar_wait_frame(i, m[i])) throw std::runtime_error("lidar_wait_frame gave up in the middle of a pass");
          const double t = clk_now(); eclk.wait += t - tw; tk0 += t - tw;   // (keeps the wait out of the assemble clock)
        }
        c.ppp += m[i].cost;
        if (!H || m[i].count == 0) continue;
        const double *L = &lmaps_[size_t(i) * LMAP_STRIDE], *Lt = L + 18 * 13;
        double HG[18 * LIO_LT_LD], gb[18];
        lidar_block(L, Lt, m[i].S, HG);
        for (int a = 0; a < 18; ++a) gb[a] = HG[a * LIO_LT_LD + 18];
        int cols[3] = {lay.pose[0], lay.pose[i], lay.ex}, sizes[3] = {6, 6, 6};
        add_block(*H, *g, cols, sizes, 3, HG, gb, LIO_LT_LD);
      }
    }
    { const double t = clk_now(); eclk.assemble += t - tk0; tk0 = t; }
    return c;
  }
};

struct SolveSummary {
  int iterations = 0, successful = 0, termination = 0;
  double initial_cost = 0, final_cost = 0;
  std::vector<double> trace;
  double ms_chol = 0, ms_eval = 0;  // LIO_DEBUG_TIMING breakdown
  std::vector<FrameMoments> final_moments;  // lidar moments at the point the solver stopped at
  WindowSystem::Costs initial_costs;
};

inline void plus_all(const WindowParams &P, const Layout &lay, const std::vector<double> &delta, WindowParams &out) {
  out = P;
  for (int i = 0; i <= P.Wo; ++i) {
    pose_plus(P.pose[i].data(), &delta[lay.pose[i]], out.pose[i].data());
    for (int k = 0; k < 9; ++k) out.sb[i][k] = P.sb[i][k] + delta[lay.sb[i] + k];
  }
  if (lay.ex >= 0) pose_plus(P.ex.data(), &delta[lay.ex], out.ex.data());
}
inline double ambient_norm(const WindowParams &P, const WindowParams *o, double *maxabs = nullptr) {
  double s = 0, mx = 0;
  auto acc = [&](const double *a, const double *b, int n) {
    for (int k = 0; k < n; ++k) { double d = b ? a[k] - b[k] : a[k]; s += d * d; mx = std::max(mx, std::fabs(d)); }
  };
  for (int i = 0; i <= P.Wo; ++i) { acc(P.pose[i].data(), o ? o->pose[i].data() : nullptr, 7); acc(P.sb[i].data(), o ? o->sb[i].data() : nullptr, 9); }
  if (!P.ex_constant) acc(P.ex.data(), o ? o->ex.data() : nullptr, 7);
  if (maxabs) *maxabs = mx;
  return std::sqrt(s);
}

// The dense factorisation split along the structure of the window: no lidar factor touches a speed-bias block, so with the
// speed-bias columns ordered first
//     A = [A11 A12; A12^T A22],   A11 = U11^T U11,   W = U11^-T A12,   A22 - W^T W = U22^T U22,
// everything up to W^T W depends only on the prior and the IMU factors of the candidate — which the host has evaluated while
// the device pass over the lidar factors is still running (WindowSystem::static_part_hook).  Once the moments arrive only the
// pose / extrinsic block (6 (Wo + 1) [+ 6] columns instead of 15 (Wo + 1) [+ 6]) remains to be factored: ~1/12 of the flops
// of the full factorisation sit on the critical path.  Same system, same solution (an elimination order, like Ceres'
// DENSE_SCHUR itself, Appendix B.3); the factor is speculative on the candidate being accepted with the regularisation the
// minimizer will then use, and is simply dropped otherwise.
struct SplitFactor {
  bool valid = false;
  double mu = 0;
  int n1 = 0, n1p = 0, n2 = 0, n = 0, ld = 0;   // speed-bias columns, the same padded to whole 8-row bands, the rest, n1p + n2, row stride
  std::vector<int> perm;          // position in the permuted system -> column of the solver's layout (-1: padding)
  std::vector<double> M, T, shift, xs;   // M: the permuted system / its factor, n x ld, column n = the right-hand side; T: what the
                                         // speed-bias rows of the factor take from the rows below them (chol_gram_avx512)
  void set_layout(const Layout &lay) {
    std::vector<char> is_sb(lay.dim, 0);
    for (int c : lay.sb) if (c >= 0) for (int k = 0; k < 9; ++k) is_sb[c + k] = 1;
    perm.clear();
    for (int i = 0; i < lay.dim; ++i) if (is_sb[i]) perm.push_back(i);
    n1 = int(perm.size()); n1p = (n1 + 7) & ~7;
    perm.resize(n1p, -1);
    for (int i = 0; i < lay.dim; ++i) if (!is_sb[i]) perm.push_back(i);
    n = int(perm.size()); n2 = n - n1p;
    ld = (n + 1 + 7) & ~7;
    M.assign(size_t(n) * ld, 0.0); T.assign(size_t(n) * ld, 0.0); shift.assign(n, 0.0); xs.assign(n, 0.0);
    valid = false;
  }
  // Hs, gs: UNSCALED prior + IMU part at the candidate (complete in the speed-bias rows); scale: the solver's fixed Jacobi scaling.
  // Factors the speed-bias rows of the permuted system [sb | rest] — U11, W = U11^-T A12 and U11^-T g1 fall out of the same band
  // sweep — and forms -W^T [W | y1], all of it while the device pass over the lidar factors is in flight.
  __attribute__((target("avx512f,fma"))) void prefactor(const DMat &Hs, const std::vector<double> &gs, const std::vector<double> &scale, double mu_) {
    valid = false; mu = mu_;
    for (int p = 0; p < n1p; ++p) {
      double *row = &M[size_t(p) * ld];
      const int i = perm[p];
      if (i < 0) { for (int q = p; q <= n; ++q) row[q] = 0.0; row[p] = 1.0; shift[p] = 0.0; continue; }
      const double si = scale[i];
      const double *hrow = &Hs.a[size_t(i) * Hs.c];
      for (int q = p; q < n; ++q) { const int j = perm[q]; row[q] = j < 0 ? 0.0 : hrow[j] * (si * scale[j]); }
      row[n] = gs[i] * si;
      const double d = std::sqrt(std::min(std::max(row[p], 1e-6), 1e32));
      shift[p] = d * d * mu;
    }
    if (!chol_upper_panels_avx512(M.data(), M.data(), n, n + 1, ld, shift.data(), 0, n1p, 0)) return;
    chol_gram_avx512(M.data(), T.data(), n, n + 1, ld, n1p, 0, n1p);
    valid = true;
  }
  // H, g: the SCALED full system of the accepted candidate; diag as the minimizer computed it.  x <- (H + mu D^2)^-1 g: only the
  // pose / extrinsic rows are left to factor.
  __attribute__((target("avx512f,fma"))) bool finish(const DMat &H, const std::vector<double> &g, const std::vector<double> &diag, std::vector<double> &x) {
    const int nfull = n & ~7;
    for (int p = n1p; p < n; ++p) {
      const int i = perm[p];
      double *row = &M[size_t(p) * ld];
      const double *trow = &T[size_t(p) * ld];
      const double *hrow = &H.a[size_t(i) * H.c];
      const int q0 = p < nfull ? (p & ~7) : p;   // chol_band_update reads a band from its first column on
      for (int q = q0; q < n; ++q) row[q] = hrow[perm[q]] + trow[q];
      row[n] = g[i] + trow[n];
      shift[p] = diag[i] * diag[i] * mu;
    }
    if (!chol_upper_panels_avx512(M.data(), M.data(), n, n + 1, ld, shift.data(), n1p, n, n1p)) return false;
    for (int p = 0; p < n; ++p) xs[p] = M[size_t(p) * ld + n];
    upper_backsolve_avx512(M.data(), n, ld, xs.data());
    x.assign(H.r, 0.0);
    for (int p = 0; p < n; ++p) if (perm[p] >= 0) x[perm[p]] = xs[p];
    return true;
  }
};

// Ceres 1.14 TrustRegionMinimizer + DoglegStrategy (TRADITIONAL_DOGLEG), jacobi_scaling = true.
// first_eval (optional) lets the caller reuse the linearisation it already made for the group costs.
struct Linearization { DMat H; std::vector<double> g; WindowSystem::Costs costs; std::vector<FrameMoments> m; bool valid = false; };

// phase clock of the loop's own arithmetic (tools/micro/host_eval_timing.hip builds with -DLIO_DOGLEG_CLOCK; off otherwise)
#ifdef LIO_DOGLEG_CLOCK
struct DoglegClock { double t[8] = {0, 0, 0, 0, 0, 0, 0, 0}; double last = 0; };
inline DoglegClock &dogleg_clock() { static DoglegClock c; return c; }
#define LIO_DCLK_START() (dogleg_clock().last = WindowSystem::clk_now())
#define LIO_DCLK(k) do { const double t_ = WindowSystem::clk_now(); dogleg_clock().t[k] += t_ - dogleg_clock().last; dogleg_clock().last = t_; } while (0)
#else
#define LIO_DCLK_START() ((void)0)
#define LIO_DCLK(k) ((void)0)
#endif

inline SolveSummary solve_dogleg(WindowSystem &sys, WindowParams &P, int max_iterations, double max_time_s, Linearization *first = nullptr) {
  using clock = std::chrono::steady_clock;
  const auto t0 = clock::now();
  SolveSummary sum;
  const int which = 1 | 2 | 4 | 8;
  Layout lay = WindowSystem::solve_layout(P);
  const int n = lay.dim;
  DMat H; std::vector<double> g;
  std::vector<FrameMoments> m_cur, m_cand;  // lidar moments at the current point / at the candidate
  WindowSystem::Costs c0;
  if (first && first->valid && first->H.r == n) { H = std::move(first->H); g = std::move(first->g); c0 = first->costs; m_cur = std::move(first->m); }
  else c0 = sys.evaluate(P, lay, which, false, &H, &g, &m_cur);
  sum.initial_costs = c0;
  double x_cost = c0.total();
  sum.initial_cost = x_cost; sum.trace.push_back(x_cost);
  std::vector<double> scale(n);
  for (int i = 0; i < n; ++i) scale[i] = 1.0 / (1.0 + std::sqrt(H(i, i)));
  auto grad_max = [&](const std::vector<double> &gu) {
    std::vector<double> neg(n);
    for (int i = 0; i < n; ++i) neg[i] = -gu[i];
    WindowParams Pp; plus_all(P, lay, neg, Pp);
    double mx = 0; ambient_norm(P, &Pp, &mx);
    return mx;
  };
  auto apply_scale = [&](DMat &Hs, std::vector<double> &gs) {
    for (int i = 0; i < n; ++i) { gs[i] *= scale[i]; double si = scale[i]; double *row = &Hs.a[size_t(i) * n]; for (int j = 0; j < n; ++j) row[j] *= si * scale[j]; }
  };
  double gmax = grad_max(g);
  apply_scale(H, g);
  double x_norm = ambient_norm(P, nullptr);
  double radius = 1e4, mu = 1e-8, alpha = 0, dogleg_norm = 0;
  const double min_mu = 1e-8, max_mu = 1.0, mu_inc = 10.0;
  bool reuse = false;
  std::vector<double> diag(n), grad(n), gn(n), step(n), tmp(n), shift(n);
  std::vector<double> A(size_t(n) * n);
  int invalid = 0, it = 0;
  DMat Hc; std::vector<double> gc;  // candidate linearisation; swapped with (H, g) on acceptance, never reallocated
  SplitFactor spec;                 // speed-bias block of the candidate, factored under its device pass
  // On by default where the host has AVX-512 (LIO_SPLIT_FACTOR=0 turns it off).  History: the first form (round 2: scalar loops, a
  // 10.6 us pre-factor on the EPYC 9575F against 4.6 us saved behind the wait) lost, because the host was not the long pole of an
  // evaluation then.  Since the resident kernel a pass takes ~15 us from ring to moments and the host's serial tail
  // behind it is a third of a linearisation; the pre-factor now runs on the band kernels of the blocked Cholesky (hlinalg.h) and
  // skips the structural zeros of the block-bidiagonal speed-bias factor.
  static const bool use_split = [] { const char *e = std::getenv("LIO_SPLIT_FACTOR"); return (e ? std::atoi(e) != 0 : true) && host_has_avx512(); }();
  if (use_split) spec.set_layout(lay);
  while (true) {
    if (it >= max_iterations) { sum.termination = 0; break; }
    if (max_time_s > 0 && std::chrono::duration<double>(clock::now() - t0).count() >= max_time_s) { sum.termination = 4; break; }
    if (gmax <= 1e-10) { sum.termination = 3; break; }
    if (radius <= 1e-32) { sum.termination = 1; break; }
    ++it;
    bool lin_ok = true;
    LIO_DCLK_START();
    if (!reuse) {
      reuse = true;
      for (int i = 0; i < n; ++i) diag[i] = std::sqrt(std::min(std::max(H(i, i), 1e-6), 1e32));
      double g2 = 0, Jg2 = 0;
      for (int i = 0; i < n; ++i) { grad[i] = g[i] / diag[i]; tmp[i] = grad[i] / diag[i]; g2 += grad[i] * grad[i]; }
      Jg2 = sym_quad(H.a.data(), tmp.data(), n, n);
      alpha = g2 / Jg2;
      lin_ok = false;
      LIO_DCLK(0);   // diag + Cauchy step length
      while (mu < max_mu) {
        const auto tc0 = clock::now();
        bool ok;
        if (spec.valid && spec.mu == mu) {   // only the pose / extrinsic block is left to factor
          ok = spec.finish(H, g, diag, gn);
          spec.valid = false;
        } else {
          for (int i = 0; i < n; ++i) shift[i] = diag[i] * diag[i] * mu;
          ok = chol_upper_from(H.a.data(), A.data(), n, n, shift.data());   // A = chol(H + mu D^2), H untouched
          if (ok) {
            gn = g;
            chol_upper_solve(A.data(), n, n, gn.data());
          }
        }
        if (ok) for (int i = 0; i < n; ++i) if (!std::isfinite(gn[i])) ok = false;
        sum.ms_chol += std::chrono::duration<double, std::milli>(clock::now() - tc0).count();
        if (!ok) { mu *= mu_inc; continue; }
        lin_ok = true;
        break;
      }
      if (lin_ok) for (int i = 0; i < n; ++i) gn[i] *= -diag[i];
      LIO_DCLK(1);   // factorisation + Gauss-Newton step
    }
    bool valid = lin_ok;
    double model_change = 0;
    if (lin_ok) {
      double gnorm = 0, gnn = 0, gdot = 0;
      for (int i = 0; i < n; ++i) { gnorm += grad[i] * grad[i]; gnn += gn[i] * gn[i]; gdot += grad[i] * gn[i]; }
      gnorm = std::sqrt(gnorm); gnn = std::sqrt(gnn);
      if (gnn <= radius) { step = gn; dogleg_norm = gnn; }
      else if (gnorm * alpha >= radius) { for (int i = 0; i < n; ++i) step[i] = -(radius / gnorm) * grad[i]; dogleg_norm = radius; }
      else {
        const double b_dot_a = -alpha * gdot, a_sq = std::pow(alpha * gnorm, 2.0);
        const double bma = a_sq - 2 * b_dot_a + std::pow(gnn, 2);
        const double cc = b_dot_a - a_sq;
        const double d = std::sqrt(cc * cc + bma * (std::pow(radius, 2.0) - a_sq));
        const double beta = (cc <= 0) ? (d - cc) / bma : (radius * radius - a_sq) / (d + cc);
        double sn = 0;
        for (int i = 0; i < n; ++i) { step[i] = (-alpha * (1.0 - beta)) * grad[i] + beta * gn[i]; sn += step[i] * step[i]; }
        dogleg_norm = std::sqrt(sn);
      }
      double sg = 0, sHs = 0;
      for (int i = 0; i < n; ++i) step[i] /= diag[i];
      sHs = sym_quad(H.a.data(), step.data(), n, n);
      for (int i = 0; i < n; ++i) sg += step[i] * g[i];
      model_change = -(sg + 0.5 * sHs);
      if (!(model_change > 0)) valid = false;
      LIO_DCLK(2);   // dogleg combination + model change
    }
    if (!valid) {
      if (++invalid >= 5) { sum.termination = 5; break; }
      mu *= mu_inc; reuse = false;
      sum.trace.push_back(x_cost);
      continue;
    }
    invalid = 0;
    std::vector<double> delta(n);
    for (int i = 0; i < n; ++i) delta[i] = step[i] * scale[i];
    WindowParams cand;
    plus_all(P, lay, delta, cand);
    // Evaluate cost AND linearisation at the candidate in one device pass: if the step is accepted the
    // Jacobian evaluation Ceres performs next (HandleSuccessfulStep) is already done.
    LIO_DCLK(3);   // candidate parameters
    const auto te0 = clock::now();
    if (use_split && spec.n1 > 0) {
      // if this candidate is accepted the next factorisation uses mu' = max(min_mu, 2 mu / mu_inc) (below): factor its
      // speed-bias block now, while the device evaluates the lidar factors
      const double mu_next = std::max(min_mu, 2.0 * mu / mu_inc);
      sys.static_part_hook = [&spec, &scale, mu_next](const DMat &Hs, const std::vector<double> &gs) { spec.prefactor(Hs, gs, scale, mu_next); };
    }
    double cand_cost = sys.evaluate(cand, lay, which, false, &Hc, &gc, &m_cand).total();
    sys.static_part_hook = nullptr;
    sum.ms_eval += std::chrono::duration<double, std::milli>(clock::now() - te0).count();
    LIO_DCLK_START();
    double step_norm = ambient_norm(P, &cand);
    if (step_norm <= 1e-8 * (x_norm + 1e-8)) { sum.termination = 1; sum.trace.push_back(x_cost); break; }
    double cost_change = x_cost - cand_cost;
    if (std::fabs(cost_change) <= 1e-6 * x_cost) { sum.termination = 2; sum.trace.push_back(x_cost); break; }
    double rho = cost_change / model_change;
    if (rho > 1e-3) {
      P = cand;
      x_norm = ambient_norm(P, nullptr);
      x_cost = cand_cost;
      std::swap(H, Hc); g.swap(gc);
      m_cur.swap(m_cand);
      gmax = grad_max(g);
      apply_scale(H, g);
      ++sum.successful;
      if (rho < 0.25) radius *= 0.5;
      if (rho > 0.75) radius = std::max(radius, 3.0 * dogleg_norm);
      mu = std::max(min_mu, 2.0 * mu / mu_inc);
      reuse = false;
    } else {
      radius *= 0.5; reuse = true;
      spec.valid = false;   // the pre-factored block belonged to the rejected candidate
    }
    sum.trace.push_back(x_cost);
    LIO_DCLK(4);   // acceptance: norms, gradient maximum, scaling of the new (H, g)
  }
  sum.iterations = it;
  sum.final_cost = x_cost;
  sum.final_moments = std::move(m_cur);
  return sum;
}

// MarginalizationInfo::{PreMarginalize, Marginalize, GetParameterBlocks}.  `sys` must carry the OLD prior,
// pim[0] (or null) and the lidar evaluator; P are the parameters after DoubleToVector/VectorToDouble.
inline std::shared_ptr<MargPrior> marginalize(WindowSystem &sys, const WindowParams &Pin) {
  const double eps = 1e-8;
  const int Wo = Pin.Wo;
  WindowParams P = Pin;
  P.ex_constant = false;
  const bool has_imu = sys.pim[0] != nullptr;
  const bool sb0_present = has_imu || sys.prior != nullptr;
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
  DMat A; std::vector<double> b;
  bool saved = sys.use_prior_factor;
  sys.use_prior_factor = false;
  sys.evaluate(P, lay, 1 | 2 | 4, true, &A, &b);
  sys.use_prior_factor = saved;
  auto pr = std::make_shared<MargPrior>();
  pr->n = n; pr->keep = keep;
  pr->lin_jac = DMat(n, n); pr->lin_res.assign(n, 0.0);
  auto finish = [&] {
    for (const KeepBlock &kb : keep) {
      const double *src = kb.kind == 0 ? Pin.pose[kb.index + 1].data() : (kb.kind == 1 ? Pin.sb[kb.index + 1].data() : Pin.ex.data());
      pr->x0.emplace_back(src, src + kb.size);
    }
    pr->finalize();
    return pr;
  };
  if (sys.marg_schur_hook && sys.marg_schur_hook(A.a.data(), b.data(), m, n, eps, pr->lin_jac.a.data(), pr->lin_res.data())) return finish();
  std::vector<double> Amm(size_t(m) * m), ev(m), V(size_t(m) * m);
  for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) Amm[size_t(i) * m + j] = 0.5 * (A(i, j) + A(j, i));
  sym_eig(Amm.data(), m, ev.data(), V.data());
  DMat Ainv(m, m);
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < m; ++j) {
      double s = 0;
      for (int k = 0; k < m; ++k) s += V[size_t(i) * m + k] * (ev[k] > eps ? 1.0 / ev[k] : 0.0) * V[size_t(j) * m + k];
      Ainv(i, j) = s;
    }
  // T = Arm * Amm_inv  (n x m)
  DMat T(n, m);
  for (int i = 0; i < n; ++i) for (int j = 0; j < m; ++j) { double s = 0; for (int k = 0; k < m; ++k) s += A(m + i, k) * Ainv(k, j); T(i, j) = s; }
  std::vector<double> S(size_t(n) * n), bs(n);
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < m; ++k) s += T(i, k) * A(k, m + j); S[size_t(i) * n + j] = A(m + i, m + j) - s; }
    double s = 0; for (int k = 0; k < m; ++k) s += T(i, k) * b[k];
    bs[i] = b[m + i] - s;
  }
  std::vector<double> ev2(n), V2(size_t(n) * n);
  sym_eig(S.data(), n, ev2.data(), V2.data());
  for (int k = 0; k < n; ++k) {
    double Sk = ev2[k] > eps ? ev2[k] : 0.0, Sik = ev2[k] > eps ? 1.0 / ev2[k] : 0.0;
    double ss = std::sqrt(Sk), sis = std::sqrt(Sik), vb = 0;
    for (int i = 0; i < n; ++i) { pr->lin_jac(k, i) = ss * V2[size_t(i) * n + k]; vb += V2[size_t(i) * n + k] * bs[i]; }
    pr->lin_res[k] = sis * vb;
  }
  return finish();
}

}  // namespace lio
