// oracle/imu_init.h — TEST INFRASTRUCTURE ONLY (CPU oracle).  Not part of the product.
//
// Restates src/imu_processor/ImuInitializer.cc:
//   :35-47   TangentBasis            :49-91    EstimateGyroBias
//   :93-177  ApproximateGravity      :179-329  RefineGravityAccBias (A, b are NOT reset between the 5 rounds)
//   :331-397 EstimateExtrinsicRotation         :399-436 Initialization
// Eigen's ldlt()/JacobiSVD are replaced by the documented algorithms (partial-pivot elimination on the symmetric
// system; smallest right singular vector = eigenvector of A^T A) — parity with Eigen itself is UNPINNED.
// Everything ELSE in this file is pinned (round 3) against the reference's own ImuInitializer.cc compiled where it lies (oracle/ref_factors.cc,
// oracle/ref_shim, `make ref`): tests/golden/ref_imu_init_vectors.npz, tests/test_ref_imu_init_vectors.py.
#pragma once
#include <memory>
#include <vector>

#include "imu.h"

namespace orc {

typedef Twist<float> Transformf;

struct LaserTransform {  // include/imu_processor/ImuInitializer.h:60-71
  double time = 0;
  Transformf transform;
  std::shared_ptr<IntegrationBase> pre_integration;
};

// dense solve by Gaussian elimination with partial pivoting (A destroyed)
inline std::vector<double> DenseSolve(Mat A, std::vector<double> b) {
  const int n = A.r;
  for (int k = 0; k < n; ++k) {
    int piv = k;
    for (int i = k + 1; i < n; ++i) if (std::fabs(A(i, k)) > std::fabs(A(piv, k))) piv = i;
    if (piv != k) { for (int j = 0; j < n; ++j) std::swap(A(k, j), A(piv, j)); std::swap(b[k], b[piv]); }
    const double d = A(k, k);
    if (d == 0.0) continue;
    for (int i = k + 1; i < n; ++i) {
      const double f = A(i, k) / d;
      if (f == 0.0) continue;
      for (int j = k; j < n; ++j) A(i, j) -= f * A(k, j);
      b[i] -= f * b[k];
    }
  }
  std::vector<double> x(n, 0.0);
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int j = i + 1; j < n; ++j) s -= A(i, j) * x[j];
    x[i] = A(i, i) != 0.0 ? s / A(i, i) : 0.0;
  }
  return x;
}

inline void TangentBasis(const V3d &g0, V3d &b, V3d &c) {
  V3d a = g0.normalized();
  V3d tmp(0, 0, 1);
  if (a.x == tmp.x && a.y == tmp.y && a.z == tmp.z) tmp = V3d(1, 0, 0);
  b = (tmp - a * a.dot(tmp)).normalized();
  c = a.cross(b);
}

inline void EstimateGyroBias(std::vector<LaserTransform> &all, std::vector<V3d> &Bgs) {
  M3d A = M3d::Zero();
  V3d b;
  const size_t window_size = all.size() - 1;
  for (size_t i = 0; i < window_size; ++i) {
    const LaserTransform &li = all[i], &lj = all[i + 1];
    Qd q_ij = (li.transform.rot.conjugate() * lj.transform.rot).cast<double>();
    M3d tmp_A = getBlock(lj.pre_integration->jacobian_, O_R, O_BG);
    V3d tmp_b = (lj.pre_integration->delta_q_.conjugate() * q_ij).vec() * 2.0;
    A = A + tmp_A.transpose() * tmp_A;
    b = b + tmp_A.transpose() * tmp_b;
  }
  Mat Am(3, 3);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Am(i, j) = A(i, j);
  std::vector<double> x = DenseSolve(Am, {b.x, b.y, b.z});
  V3d delta_bg(x[0], x[1], x[2]);
  for (size_t i = 0; i <= window_size; ++i) Bgs[i] += delta_bg;
  for (size_t i = 0; i < window_size; ++i) all[i + 1].pre_integration->Repropagate(V3d(), Bgs[0]);
}

inline bool ApproximateGravity(std::vector<LaserTransform> &all, V3d &g, const Transformf &transform_lb) {
  const size_t window_size = all.size() - 1;
  if (window_size < 5) return false;
  double A = 0;  // tmp_A is a multiple of I3: the 3x3 system is diagonal with equal entries
  V3d b;
  for (size_t i = 0; i + 1 < window_size; ++i) {
    const LaserTransform &li = all[i], &lj = all[i + 1], &lk = all[i + 2];
    const double dt12 = lj.pre_integration->sum_dt_, dt23 = lk.pre_integration->sum_dt_;
    const V3d dp12 = lj.pre_integration->delta_p_, dp23 = lk.pre_integration->delta_p_, dv12 = lj.pre_integration->delta_v_;
    const V3d pl1 = li.transform.pos.cast<double>(), pl2 = lj.transform.pos.cast<double>(), pl3 = lk.transform.pos.cast<double>();
    const V3d plb = transform_lb.pos.cast<double>();
    const M3d rl1 = li.transform.rot.cast<double>().toRotationMatrix(), rl2 = lj.transform.rot.cast<double>().toRotationMatrix(),
              rl3 = lk.transform.rot.cast<double>().toRotationMatrix(), rlb = transform_lb.rot.cast<double>().toRotationMatrix();
    const double a = 0.5 * (dt12 * dt12 * dt23 + dt23 * dt23 * dt12);
    V3d tmp_b = (pl2 - pl1) * dt23 - (pl3 - pl2) * dt12 + ((rl2 - rl1) * plb) * dt23 - ((rl3 - rl2) * plb) * dt12 +
                (rl2 * (rlb * dp23)) * dt12 + (rl1 * (rlb * dv12)) * dt12 * dt23 - (rl1 * (rlb * dp12)) * dt23;
    A += a * a;
    b = b - tmp_b * a;
  }
  A *= 10000.0;
  b = b * 10000.0;
  g = A != 0.0 ? b / A : V3d();
  const double g_norm = all.front().pre_integration ? all.front().pre_integration->config_.g_norm : all[1].pre_integration->config_.g_norm;
  return std::fabs(g.norm() - g_norm) <= 1.0;
}

inline void RefineGravityAccBias(std::vector<LaserTransform> &all, std::vector<V3d> &Vs, V3d &g_refined, const Transformf &transform_lb, M3d &R_WI) {
  const size_t nv = all.size();
  const int ns = int(nv) * 3 + 2;
  Mat A(ns, ns);
  std::vector<double> b(ns, 0.0), x(ns, 0.0);
  const double g_norm = all.front().pre_integration ? all.front().pre_integration->config_.g_norm : all[1].pre_integration->config_.g_norm;
  g_refined = g_refined.normalized() * g_norm;
  for (int k = 0; k < 5; ++k) {
    V3d lx, ly;
    TangentBasis(g_refined, lx, ly);
    for (size_t i = 0; i + 1 < nv; ++i) {
      const LaserTransform &li = all[i], &lj = all[i + 1];
      double tA[6][8] = {{0}};
      double tb[6] = {0};
      const double dt12 = lj.pre_integration->sum_dt_;
      const V3d dp12 = lj.pre_integration->delta_p_, dv12 = lj.pre_integration->delta_v_;
      const V3d pl1 = li.transform.pos.cast<double>(), pl2 = lj.transform.pos.cast<double>(), plb = transform_lb.pos.cast<double>();
      const M3d rl1 = li.transform.rot.normalized().cast<double>().toRotationMatrix(),
                rl2 = lj.transform.rot.normalized().cast<double>().toRotationMatrix(),
                rlb = transform_lb.rot.normalized().cast<double>().toRotationMatrix();
      for (int d = 0; d < 3; ++d) {
        tA[d][d] = dt12;
        tA[d][6] = 0.5 * lx[d] * dt12 * dt12; tA[d][7] = 0.5 * ly[d] * dt12 * dt12;
        tA[3 + d][d] = 1.0; tA[3 + d][3 + d] = -1.0;
        tA[3 + d][6] = lx[d] * dt12; tA[3 + d][7] = ly[d] * dt12;
      }
      V3d b0 = pl2 - pl1 - rl1 * (rlb * dp12) - (rl1 - rl2) * plb - g_refined * (0.5 * dt12 * dt12);
      V3d b1 = -(rl1 * (rlb * dv12)) - g_refined * dt12;
      for (int d = 0; d < 3; ++d) { tb[d] = b0[d]; tb[3 + d] = b1[d]; }
      double rA[8][8], rb[8];
      for (int r = 0; r < 8; ++r) {
        for (int c = 0; c < 8; ++c) { double s = 0; for (int m = 0; m < 6; ++m) s += tA[m][r] * tA[m][c]; rA[r][c] = s; }
        double s = 0; for (int m = 0; m < 6; ++m) s += tA[m][r] * tb[m]; rb[r] = s;
      }
      const int o = int(i) * 3;
      for (int r = 0; r < 6; ++r) {
        for (int c = 0; c < 6; ++c) A(o + r, o + c) += rA[r][c];
        b[o + r] += rb[r];
        for (int c = 0; c < 2; ++c) { A(o + r, ns - 2 + c) += rA[r][6 + c]; A(ns - 2 + c, o + r) += rA[6 + c][r]; }
      }
      for (int r = 0; r < 2; ++r) {
        for (int c = 0; c < 2; ++c) A(ns - 2 + r, ns - 2 + c) += rA[6 + r][6 + c];
        b[ns - 2 + r] += rb[6 + r];
      }
    }
    for (double &v : A.a) v *= 1000.0;
    for (double &v : b) v *= 1000.0;
    x = DenseSolve(A, b);
    const double dgx = x[ns - 2], dgy = x[ns - 1];
    g_refined = (g_refined + lx * dgx + ly * dgy).normalized() * g_norm;
  }
  const V3d gI_n(0.0, 0.0, -1.0);
  const V3d gW_n = g_refined.normalized();
  const V3d gIxgW = gI_n.cross(gW_n);
  const V3d v_WI = gIxgW / gIxgW.norm();
  const double ang_WI = std::atan2(gIxgW.norm(), gI_n.dot(gW_n));
  // Sophus::SO3d::exp(ang * v).unit_quaternion() (so3.hpp:534-568)
  const V3d omega = v_WI * ang_WI;
  const double theta_sq = omega.squaredNorm(), theta = std::sqrt(theta_sq), half_theta = 0.5 * theta;
  double imag, real;
  if (theta < 1e-10) {
    const double t4 = theta_sq * theta_sq;
    imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * t4;
    real = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * t4;
  } else {
    imag = std::sin(half_theta) / theta;
    real = std::cos(half_theta);
  }
  R_WI = Qd(real, imag * omega.x, imag * omega.y, imag * omega.z).toRotationMatrix();
  for (size_t i = 0; i < nv; ++i) Vs[i] = V3d(x[i * 3], x[i * 3 + 1], x[i * 3 + 2]);
}

// returns true when the second-smallest singular value exceeds 0.25 (:389-395)
inline bool EstimateExtrinsicRotation(std::vector<LaserTransform> &all, Transformf &transform_lb) {
  const Transformf transform_bl = transform_lb.inverse();
  const Qd rot_bl = transform_bl.rot.cast<double>();
  const size_t window_size = all.size() - 1;
  double AtA[16] = {0};
  for (size_t i = 0; i < window_size; ++i) {
    const LaserTransform &li = all[i], &lj = all[i + 1];
    const Qd dq_imu = lj.pre_integration->delta_q_;
    const Qd dq_laser = (li.transform.rot.conjugate() * lj.transform.rot).cast<double>();
    const Qd dq_laser_from_imu = rot_bl.conjugate() * dq_imu * rot_bl;
    const double angular_distance = 180 / M_PI * dq_laser.angularDistance(dq_laser_from_imu);
    const double huber = angular_distance > 5.0 ? 5.0 / angular_distance : 1.0;
    double L4[4][4], R4[4][4], B[4][4];
    LeftQuat4(dq_laser, L4); RightQuat4(dq_imu, R4);
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) B[r][c] = huber * (L4[r][c] - R4[r][c]);
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { double s = 0; for (int m = 0; m < 4; ++m) s += B[m][r] * B[m][c]; AtA[r * 4 + c] += s; }
  }
  double E[4], V[16];
  sym_eigen<double>(4, AtA, E, V);
  Qd q(V[3 * 4 + 0], V[0 * 4 + 0], V[1 * 4 + 0], V[2 * 4 + 0]);  // coefficient order x,y,z,w; column of the smallest eigenvalue
  transform_lb.rot = Q<float>::FromMatrix(q.cast<float>().toRotationMatrix());
  const double s2 = std::sqrt(std::max(E[1], 0.0));
  return s2 > 0.25;
}

inline bool Initialization(std::vector<LaserTransform> &all, std::vector<V3d> &Vs, std::vector<V3d> &Bgs, V3d &g, const Transformf &transform_lb,
                           M3d &R_WI) {
  EstimateGyroBias(all, Bgs);
  if (!ApproximateGravity(all, g, transform_lb)) return false;
  RefineGravityAccBias(all, Vs, g, transform_lb, R_WI);
  return true;
}

}  // namespace orc
