// host_init.h — IMU initialisation of the estimator (host side; <= window_size+1 frames of small dense algebra).
// Reference: src/imu_processor/ImuInitializer.cc
//   :35-47 TangentBasis, :49-91 EstimateGyroBias, :93-177 ApproximateGravity, :179-329 RefineGravityAccBias,
//   :331-397 EstimateExtrinsicRotation, :399-436 Initialization.
// The normal equations here are tiny (3x3, (3n+2)x(3n+2) with n <= 16, 4x4) and solved once per initialisation
// attempt: they stay on the host next to the solver state.  Dense SPD solves use Cholesky (Eigen's ldlt() in the
// reference); the null vector of the 4n x 4 extrinsic system is the lowest eigenvector of its 4x4 Gram matrix.
#pragma once
#include <memory>
#include <vector>

#include "hlinalg.h"
#include "host_factors.h"

namespace lio {

struct LaserFrame {  // LaserTransform (include/imu_processor/ImuInitializer.h:60-71)
  double time = 0;
  Rigid<float> transform;
  std::shared_ptr<Preintegration> pim;
};

namespace init_detail {

inline Qd castd(const Quat<float> &q) { return Qd(double(q.w), double(q.x), double(q.y), double(q.z)); }
inline V3d castd(const Vec3<float> &v) { return V3d(double(v.x), double(v.y), double(v.z)); }
inline V3d unit(const V3d &v) { double n2 = dot(v, v); return n2 > 0 ? v / std::sqrt(n2) : v; }

// symmetric positive (semi-)definite solve; falls back to pivoted elimination when the factorisation breaks down
inline std::vector<double> spd_solve(std::vector<double> A, std::vector<double> b, int n) {
  std::vector<double> L = A, x = b;
  if (chol_factor(L.data(), n, n)) {
    chol_solve_inplace(L.data(), n, n, x.data());
    return x;
  }
  for (int k = 0; k < n; ++k) {
    int piv = k;
    for (int i = k + 1; i < n; ++i) if (std::fabs(A[i * n + k]) > std::fabs(A[piv * n + k])) piv = i;
    if (piv != k) { for (int j = 0; j < n; ++j) std::swap(A[k * n + j], A[piv * n + j]); std::swap(b[k], b[piv]); }
    const double d = A[k * n + k];
    if (d == 0.0) continue;
    for (int i = k + 1; i < n; ++i) {
      const double f = A[i * n + k] / d;
      for (int j = k; j < n; ++j) A[i * n + j] -= f * A[k * n + j];
      b[i] -= f * b[k];
    }
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int j = i + 1; j < n; ++j) s -= A[i * n + j] * x[j];
    x[i] = A[i * n + i] != 0.0 ? s / A[i * n + i] : 0.0;
  }
  return x;
}

inline void tangent_basis(const V3d &g0, V3d &b, V3d &c) {
  const V3d a = unit(g0);
  V3d tmp(0, 0, 1);
  if (a.x == 0.0 && a.y == 0.0 && a.z == 1.0) tmp = V3d(1, 0, 0);
  b = unit(tmp - a * dot(a, tmp));
  c = cross(a, b);
}

inline double g_norm_of(const std::vector<LaserFrame> &all) { return (all[0].pim ? all[0].pim : all[1].pim)->noise.g_norm; }

}  // namespace init_detail

inline void estimate_gyro_bias(std::vector<LaserFrame> &all, std::vector<V3d> &Bgs) {
  using namespace init_detail;
  std::vector<double> A(9, 0.0), b(3, 0.0);
  const size_t ws = all.size() - 1;
  for (size_t i = 0; i < ws; ++i) {
    const LaserFrame &fi = all[i], &fj = all[i + 1];
    const Qd q_ij = castd(conj(fi.transform.rot) * fj.transform.rot);
    const M3d J = get3(fj.pim->jac, 15, kOR, kOBG);
    const V3d r = 2.0 * (conj(fj.pim->dq) * q_ij).vec();
    const M3d JtJ = transpose(J) * J;
    const V3d Jtr = transpose(J) * r;
    for (int a = 0; a < 3; ++a) { for (int c = 0; c < 3; ++c) A[a * 3 + c] += JtJ(a, c); }
    b[0] += Jtr.x; b[1] += Jtr.y; b[2] += Jtr.z;
  }
  const std::vector<double> x = spd_solve(A, b, 3);
  const V3d dbg(x[0], x[1], x[2]);
  for (size_t i = 0; i <= ws; ++i) Bgs[i] = Bgs[i] + dbg;
  for (size_t i = 0; i < ws; ++i) all[i + 1].pim->repropagate(V3d(), Bgs[0]);
}

inline bool approximate_gravity(const std::vector<LaserFrame> &all, V3d &g, const Rigid<float> &lb) {
  using namespace init_detail;
  const size_t ws = all.size() - 1;
  if (ws < 5) return false;
  double A = 0;  // every block is a multiple of I3
  V3d b;
  const V3d plb = castd(lb.pos);
  const M3d rlb = toRot(castd(lb.rot));
  for (size_t i = 0; i + 1 < ws; ++i) {
    const LaserFrame &f1 = all[i], &f2 = all[i + 1], &f3 = all[i + 2];
    const double dt12 = f2.pim->sum_dt, dt23 = f3.pim->sum_dt;
    const V3d pl1 = castd(f1.transform.pos), pl2 = castd(f2.transform.pos), pl3 = castd(f3.transform.pos);
    const M3d rl1 = toRot(castd(f1.transform.rot)), rl2 = toRot(castd(f2.transform.rot)), rl3 = toRot(castd(f3.transform.rot));
    const double a = 0.5 * (dt12 * dt12 * dt23 + dt23 * dt23 * dt12);
    const V3d tb = (pl2 - pl1) * dt23 - (pl3 - pl2) * dt12 + ((rl2 - rl1) * plb) * dt23 - ((rl3 - rl2) * plb) * dt12 +
                   ((rl2 * rlb) * f3.pim->dp) * dt12 + ((rl1 * rlb) * f2.pim->dv) * (dt12 * dt23) - ((rl1 * rlb) * f2.pim->dp) * dt23;
    A += a * a;
    b = b - tb * a;
  }
  g = A != 0.0 ? b / A : V3d();   // the common factor 10000 of :158-159 cancels
  return std::fabs(norm(g) - g_norm_of(all)) <= 1.0;
}

inline void refine_gravity(std::vector<LaserFrame> &all, std::vector<V3d> &Vs, V3d &g, const Rigid<float> &lb, M3d &R_WI) {
  using namespace init_detail;
  const size_t nv = all.size();
  const int ns = int(nv) * 3 + 2;
  std::vector<double> A(size_t(ns) * ns, 0.0), b(ns, 0.0), x(ns, 0.0);
  const double g_norm = g_norm_of(all);
  const V3d plb = castd(lb.pos);
  const M3d rlb = toRot(castd(normalized(lb.rot)));
  g = unit(g) * g_norm;
  for (int round = 0; round < 5; ++round) {   // A and b carry over between rounds, scaled by 1000 each time (:297-299)
    V3d lx, ly;
    tangent_basis(g, lx, ly);
    for (size_t i = 0; i + 1 < nv; ++i) {
      const LaserFrame &f1 = all[i], &f2 = all[i + 1];
      const double dt = f2.pim->sum_dt;
      const V3d pl1 = castd(f1.transform.pos), pl2 = castd(f2.transform.pos);
      const M3d rl1 = toRot(castd(normalized(f1.transform.rot))), rl2 = toRot(castd(normalized(f2.transform.rot)));
      double J[6][8] = {{0}}, r[6];
      const double lxv[3] = {lx.x, lx.y, lx.z}, lyv[3] = {ly.x, ly.y, ly.z};
      for (int d = 0; d < 3; ++d) {
        J[d][d] = dt; J[d][6] = 0.5 * lxv[d] * dt * dt; J[d][7] = 0.5 * lyv[d] * dt * dt;
        J[3 + d][d] = 1.0; J[3 + d][3 + d] = -1.0; J[3 + d][6] = lxv[d] * dt; J[3 + d][7] = lyv[d] * dt;
      }
      const V3d r0 = pl2 - pl1 - (rl1 * rlb) * f2.pim->dp - (rl1 - rl2) * plb - g * (0.5 * dt * dt);
      const V3d r1 = -((rl1 * rlb) * f2.pim->dv) - g * dt;
      r[0] = r0.x; r[1] = r0.y; r[2] = r0.z; r[3] = r1.x; r[4] = r1.y; r[5] = r1.z;
      // scatter J^T J / J^T r: columns 0..5 -> states of frames i, i+1; columns 6,7 -> the two gravity tangents
      auto col = [&](int c) { return c < 6 ? int(i) * 3 + c : ns - 2 + (c - 6); };
      for (int p = 0; p < 8; ++p) {
        double sr = 0;
        for (int m = 0; m < 6; ++m) sr += J[m][p] * r[m];
        b[col(p)] += sr;
        for (int q = 0; q < 8; ++q) {
          double s = 0;
          for (int m = 0; m < 6; ++m) s += J[m][p] * J[m][q];
          A[size_t(col(p)) * ns + col(q)] += s;
        }
      }
    }
    for (double &v : A) v *= 1000.0;
    for (double &v : b) v *= 1000.0;
    x = spd_solve(A, b, ns);
    g = unit(g + lx * x[ns - 2] + ly * x[ns - 1]) * g_norm;
  }
  // rotation taking the inertial -z axis onto the refined gravity direction (:303-312)
  const V3d gI(0.0, 0.0, -1.0), gW = unit(g);
  const V3d ax = cross(gI, gW);
  const double s = norm(ax), ang = std::atan2(s, dot(gI, gW));
  const V3d om = (ax / s) * ang;
  const double th2 = dot(om, om), th = std::sqrt(th2);
  double im, re;
  if (th < 1e-10) { im = 0.5 - th2 / 48.0 + th2 * th2 / 3840.0; re = 1.0 - th2 / 8.0 + th2 * th2 / 384.0; }
  else { im = std::sin(0.5 * th) / th; re = std::cos(0.5 * th); }
  R_WI = toRot(Qd(re, im * om.x, im * om.y, im * om.z));
  for (size_t i = 0; i < nv; ++i) Vs[i] = V3d(x[i * 3], x[i * 3 + 1], x[i * 3 + 2]);
}

// true when the calibration is accepted (second-smallest singular value > 0.25, :389-395)
inline bool estimate_extrinsic_rotation(const std::vector<LaserFrame> &all, Rigid<float> &lb) {
  using namespace init_detail;
  const Qd q_bl = castd(rinverse(lb).rot);
  const size_t ws = all.size() - 1;
  double G[16] = {0};
  for (size_t i = 0; i < ws; ++i) {
    const LaserFrame &fi = all[i], &fj = all[i + 1];
    const Qd q_imu = fj.pim->dq;
    const Qd q_laser = castd(conj(fi.transform.rot) * fj.transform.rot);
    const Qd q_pred = (conj(q_bl) * q_imu) * q_bl;
    const Qd d = q_laser * conj(q_pred);
    const double ang_deg = 180 / M_PI * 2.0 * std::atan2(norm(d.vec()), std::fabs(d.w));
    const double w = ang_deg > 5.0 ? 5.0 / ang_deg : 1.0;
    // B = w (L(q_laser) - R(q_imu)), quaternion coefficient order x,y,z,w (math_utils.h:139-161)
    double B[4][4];
    const M3d Sl = skew(q_laser.vec()), Sr = skew(q_imu.vec());
    const double lv[3] = {q_laser.x, q_laser.y, q_laser.z}, rv[3] = {q_imu.x, q_imu.y, q_imu.z};
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) B[r][c] = w * (((r == c) ? q_laser.w : 0.0) + Sl(r, c) - (((r == c) ? q_imu.w : 0.0) - Sr(r, c)));
      B[r][3] = w * (lv[r] - rv[r]);
      B[3][r] = w * (-lv[r] + rv[r]);
    }
    B[3][3] = w * (q_laser.w - q_imu.w);
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { double sacc = 0; for (int m = 0; m < 4; ++m) sacc += B[m][r] * B[m][c]; G[r * 4 + c] += sacc; }
  }
  double ev[4], V[16];
  if (!sym_eig(G, 4, ev, V)) return false;
  const Qd q(V[3 * 4 + 0], V[0 * 4 + 0], V[1 * 4 + 0], V[2 * 4 + 0]);
  const Quat<float> qf(float(q.w), float(q.x), float(q.y), float(q.z));
  lb.rot = fromRot(toRot(qf));   // Quaternionf = Matrix3f (:382)
  return std::sqrt(std::max(ev[1], 0.0)) > 0.25;
}

inline bool imu_initialization(std::vector<LaserFrame> &all, std::vector<V3d> &Vs, std::vector<V3d> &Bgs, V3d &g, const Rigid<float> &lb, M3d &R_WI) {
  estimate_gyro_bias(all, Bgs);
  if (!approximate_gravity(all, g, lb)) return false;
  refine_gravity(all, Vs, g, lb, R_WI);
  return true;
}

}  // namespace lio
