// oracle/imu.h — TEST INFRASTRUCTURE ONLY (CPU oracle).  Not part of the product.
//
// CPU restatement of
//   include/imu_processor/IntegrationBase.h:77-357  (mid-point pre-integration, Evaluate)
//   include/factor/ImuFactor.h:53-168               (ImuFactor::Evaluate)
//   src/factor/PivotPointPlaneFactor.cc:43-137      (PivotPointPlaneFactor::Evaluate)
//   src/factor/PriorFactor.cc:35-67                 (PriorFactor::Evaluate)
//   src/factor/PoseLocalParameterization.cc:35-59   (Plus / ComputeJacobian)
// Pinned against the reference's only fixture for this path (test/data/imu_pose_vel.txt, intent of
// test_imu_factor.cc:435-444: residual at ground truth ~ 0) in tests/test_oracle_imu.py, and — since round 3 — against the
// outputs of those five reference sources THEMSELVES, compiled where they lie against stand-in headers (oracle/ref_shim,
// oracle/ref_factors.cc, `make ref`; committed as tests/golden/ref_factor_vectors.npz): tests/test_ref_factor_vectors.py,
// bit for bit.
#pragma once
#include <memory>

#include "liomath.h"

namespace orc {

typedef V3<double> V3d;
typedef M3<double> M3d;
typedef Q<double> Qd;

enum { O_P = 0, O_R = 3, O_V = 6, O_BA = 9, O_BG = 12 };  // IntegrationBase.h:56-62

struct PimConfig {
  double acc_n = 0.1, gyr_n = 0.01, acc_w = 0.0002, gyr_w = 2.0e-5, g_norm = 9.805;
};

static inline void setBlock(Mat &M, int r, int c, const M3d &B) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M(r + i, c + j) = B(i, j);
}
static inline M3d getBlock(const Mat &M, int r, int c) {
  M3d B;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) B(i, j) = M(r + i, c + j);
  return B;
}

struct IntegrationBase {
  double dt_ = 0;
  V3d acc0_, gyr0_, acc1_, gyr1_;
  V3d linearized_acc_, linearized_gyr_;
  V3d linearized_ba_, linearized_bg_;
  Mat jacobian_, covariance_, noise_;
  double sum_dt_ = 0;
  V3d delta_p_, delta_v_;
  Qd delta_q_;
  std::vector<double> dt_buf_;
  std::vector<V3d> acc_buf_, gyr_buf_;
  PimConfig config_;
  V3d g_vec_;

  IntegrationBase(const V3d &acc0, const V3d &gyr0, const V3d &ba, const V3d &bg, const PimConfig &cfg)
      : acc0_(acc0), gyr0_(gyr0), linearized_acc_(acc0), linearized_gyr_(gyr0), linearized_ba_(ba), linearized_bg_(bg),
        jacobian_(Mat::Identity(15)), covariance_(15, 15), noise_(18, 18), config_(cfg) {
    g_vec_ = V3d(0, 0, -cfg.g_norm);
    double an = cfg.acc_n * cfg.acc_n, gn = cfg.gyr_n * cfg.gyr_n, aw = cfg.acc_w * cfg.acc_w, gw = cfg.gyr_w * cfg.gyr_w;
    for (int i = 0; i < 3; ++i) {
      noise_(i, i) = an; noise_(3 + i, 3 + i) = gn; noise_(6 + i, 6 + i) = an;
      noise_(9 + i, 9 + i) = gn; noise_(12 + i, 12 + i) = aw; noise_(15 + i, 15 + i) = gw;
    }
  }

  void push_back(double dt, const V3d &acc, const V3d &gyr) {
    dt_buf_.push_back(dt); acc_buf_.push_back(acc); gyr_buf_.push_back(gyr);
    Propagate(dt, acc, gyr);
  }

  void Repropagate(const V3d &ba, const V3d &bg) {  // :110-125
    sum_dt_ = 0.0;
    acc0_ = linearized_acc_; gyr0_ = linearized_gyr_;
    delta_p_ = V3d(); delta_q_ = Qd(); delta_v_ = V3d();
    linearized_ba_ = ba; linearized_bg_ = bg;
    jacobian_ = Mat::Identity(15); covariance_.setZero();
    for (size_t i = 0; i < dt_buf_.size(); ++i) Propagate(dt_buf_[i], acc_buf_[i], gyr_buf_[i]);
  }

  // :127-209 (update_jacobian always true on the hot path)
  void Propagate(double dt, const V3d &acc1, const V3d &gyr1) {
    dt_ = dt; acc1_ = acc1; gyr1_ = gyr1;
    const V3d &ba = linearized_ba_, &bg = linearized_bg_;
    V3d un_acc_0 = delta_q_ * (acc0_ - ba);
    V3d un_gyr = 0.5 * (gyr0_ + gyr1) - bg;
    Qd rq = delta_q_ * Qd(1, un_gyr.x * dt / 2, un_gyr.y * dt / 2, un_gyr.z * dt / 2);
    V3d un_acc_1 = rq * (acc1 - ba);
    V3d un_acc = 0.5 * (un_acc_0 + un_acc_1);
    V3d rp = delta_p_ + delta_v_ * dt + 0.5 * un_acc * dt * dt;
    V3d rv = delta_v_ + un_acc * dt;

    V3d w_x = 0.5 * (gyr0_ + gyr1) - bg;
    V3d a_0_x = acc0_ - ba, a_1_x = acc1 - ba;
    M3d R_w_x = Skew(w_x), R_a_0_x = Skew(a_0_x), R_a_1_x = Skew(a_1_x);
    M3d I = M3d::Identity();
    M3d Rq = delta_q_.toRotationMatrix(), Rr = rq.toRotationMatrix();
    Mat F(15, 15);
    setBlock(F, 0, 0, I);
    setBlock(F, 0, 3, (Rq * R_a_0_x) * (-0.25) * dt * dt + ((Rr * R_a_1_x) * (I - R_w_x * dt)) * (-0.25) * dt * dt);
    setBlock(F, 0, 6, I * dt);
    setBlock(F, 0, 9, (Rq + Rr) * (-0.25) * dt * dt);
    setBlock(F, 0, 12, (Rr * R_a_1_x) * (-0.1667) * dt * dt * (-dt));  // :173 (A.10)
    setBlock(F, 3, 3, I - R_w_x * dt);
    setBlock(F, 3, 12, I * (-1.0) * dt);
    setBlock(F, 6, 3, (Rq * R_a_0_x) * (-0.5) * dt + ((Rr * R_a_1_x) * (I - R_w_x * dt)) * (-0.5) * dt);
    setBlock(F, 6, 6, I);
    setBlock(F, 6, 9, (Rq + Rr) * (-0.5) * dt);
    setBlock(F, 6, 12, (Rr * R_a_1_x) * (-0.5) * dt * (-dt));
    setBlock(F, 9, 9, I);
    setBlock(F, 12, 12, I);
    Mat V(15, 18);
    setBlock(V, 0, 0, Rq * 0.5 * dt * dt);  // :189 (A.10)
    M3d v03 = ((-Rr) * 0.25 * R_a_1_x) * dt * dt * 0.5 * dt;
    setBlock(V, 0, 3, v03);
    setBlock(V, 0, 6, Rr * 0.5 * dt * dt);  // :192
    setBlock(V, 0, 9, v03);
    setBlock(V, 3, 3, I * 0.5 * dt);
    setBlock(V, 3, 9, I * 0.5 * dt);
    setBlock(V, 6, 0, Rq * 0.5 * dt);
    M3d v63 = ((-Rr) * 0.5 * R_a_1_x) * dt * 0.5 * dt;
    setBlock(V, 6, 3, v63);
    setBlock(V, 6, 6, Rr * 0.5 * dt);
    setBlock(V, 6, 9, v63);
    setBlock(V, 9, 12, I * dt);
    setBlock(V, 12, 15, I * dt);
    jacobian_ = matmul(F, jacobian_);
    covariance_ = matmul(matmul(F, covariance_), F.transpose());
    Mat VQVt = matmul(matmul(V, noise_), V.transpose());
    for (size_t i = 0; i < covariance_.a.size(); ++i) covariance_.a[i] += VQVt.a[i];

    delta_p_ = rp; delta_q_ = rq; delta_v_ = rv;
    delta_q_.normalize();  // :302
    sum_dt_ += dt_;
    acc0_ = acc1_; gyr0_ = gyr1_;
  }

  // :309-357
  void Evaluate(const V3d &Pi, const Qd &Qi, const V3d &Vi, const V3d &Bai, const V3d &Bgi, const V3d &Pj, const Qd &Qj,
                const V3d &Vj, const V3d &Baj, const V3d &Bgj, double res[15]) const {
    M3d dp_dba = getBlock(jacobian_, O_P, O_BA), dp_dbg = getBlock(jacobian_, O_P, O_BG);
    M3d dq_dbg = getBlock(jacobian_, O_R, O_BG);
    M3d dv_dba = getBlock(jacobian_, O_V, O_BA), dv_dbg = getBlock(jacobian_, O_V, O_BG);
    V3d dba = Bai - linearized_ba_, dbg = Bgi - linearized_bg_;
    Qd cq = delta_q_ * DeltaQ(dq_dbg * dbg);
    V3d cv = delta_v_ + dv_dba * dba + dv_dbg * dbg;
    V3d cp = delta_p_ + dp_dba * dba + dp_dbg * dbg;
    V3d rp = Qi.inverse() * (-0.5 * g_vec_ * sum_dt_ * sum_dt_ + Pj - Pi - Vi * sum_dt_) - cp;
    V3d rr = 2.0 * (cq.inverse() * (Qi.inverse() * Qj)).vec();
    V3d rv = Qi.inverse() * (-1.0 * g_vec_ * sum_dt_ + Vj - Vi) - cv;
    V3d rba = Baj - Bai, rbg = Bgj - Bgi;
    for (int k = 0; k < 3; ++k) { res[O_P + k] = rp[k]; res[O_R + k] = rr[k]; res[O_V + k] = rv[k]; res[O_BA + k] = rba[k]; res[O_BG + k] = rbg[k]; }
  }
};

static inline void unpackPose(const double *p, V3d &P, Qd &Qq) { P = V3d(p[0], p[1], p[2]); Qq = Qd(p[6], p[3], p[4], p[5]); }

// ImuFactor.h:53-168.  Jacobians row-major 15x7, 15x9, 15x7, 15x9; null pointers are skipped.
// sqrt_info = LLT(cov^-1).matrixL().transpose()  (recomputed at every call in the reference, A.11)
inline bool ImuSqrtInfo(const IntegrationBase &pim, Mat &sqrt_info) {
  Mat cinv, L;
  if (!inverse(pim.covariance_, cinv)) return false;
  if (!cholesky(cinv, L)) return false;
  sqrt_info = L.transpose();
  return true;
}

inline bool ImuFactorEvaluate(const IntegrationBase &pim, const double *const *par, double *residuals, double **jac) {
  V3d Pi, Pj; Qd Qi, Qj;
  unpackPose(par[0], Pi, Qi); unpackPose(par[2], Pj, Qj);
  V3d Vi(par[1][0], par[1][1], par[1][2]), Bai(par[1][3], par[1][4], par[1][5]), Bgi(par[1][6], par[1][7], par[1][8]);
  V3d Vj(par[3][0], par[3][1], par[3][2]), Baj(par[3][3], par[3][4], par[3][5]), Bgj(par[3][6], par[3][7], par[3][8]);
  double r[15];
  pim.Evaluate(Pi, Qi, Vi, Bai, Bgi, Pj, Qj, Vj, Baj, Bgj, r);
  Mat S;
  if (!ImuSqrtInfo(pim, S)) return false;
  for (int i = 0; i < 15; ++i) { double s = 0; for (int k = 0; k < 15; ++k) s += S(i, k) * r[k]; residuals[i] = s; }
  if (!jac) return true;
  double sum_dt = pim.sum_dt_;
  M3d dp_dba = getBlock(pim.jacobian_, O_P, O_BA), dp_dbg = getBlock(pim.jacobian_, O_P, O_BG);
  M3d dq_dbg = getBlock(pim.jacobian_, O_R, O_BG);
  M3d dv_dba = getBlock(pim.jacobian_, O_V, O_BA), dv_dbg = getBlock(pim.jacobian_, O_V, O_BG);
  const V3d &g = pim.g_vec_;
  M3d RiT = Qi.inverse().toRotationMatrix();
  Qd cq = pim.delta_q_ * DeltaQ(dq_dbg * (Bgi - pim.linearized_bg_));
  auto whiten = [&](const Mat &J, int cols, double *out) {
    for (int i = 0; i < 15; ++i)
      for (int j = 0; j < cols; ++j) { double s = 0; for (int k = 0; k < 15; ++k) s += S(i, k) * J(k, j); out[i * cols + j] = s; }
  };
  if (jac[0]) {
    Mat J(15, 7);
    setBlock(J, O_P, O_P, -RiT);
    setBlock(J, O_P, O_R, Skew(Qi.inverse() * (-0.5 * g * sum_dt * sum_dt + Pj - Pi - Vi * sum_dt)));
    double L4[4][4], R4[4][4];
    LeftQuat4(Qj.inverse() * Qi, L4); RightQuat4(cq, R4);
    M3d LR;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += L4[i][k] * R4[k][j]; LR(i, j) = s; }
    setBlock(J, O_R, O_R, -LR);
    setBlock(J, O_V, O_R, Skew(Qi.inverse() * (-1.0 * g * sum_dt + Vj - Vi)));
    whiten(J, 7, jac[0]);
  }
  if (jac[1]) {
    Mat J(15, 9);
    setBlock(J, O_P, O_V - O_V, -RiT * sum_dt);
    setBlock(J, O_P, O_BA - O_V, -dp_dba);
    setBlock(J, O_P, O_BG - O_V, -dp_dbg);
    setBlock(J, O_R, O_BG - O_V, -LeftQuatTL3(Qj.inverse() * Qi * cq) * dq_dbg);
    setBlock(J, O_V, O_V - O_V, -RiT);
    setBlock(J, O_V, O_BA - O_V, -dv_dba);
    setBlock(J, O_V, O_BG - O_V, -dv_dbg);
    setBlock(J, O_BA, O_BA - O_V, -M3d::Identity());
    setBlock(J, O_BG, O_BG - O_V, -M3d::Identity());
    whiten(J, 9, jac[1]);
  }
  if (jac[2]) {
    Mat J(15, 7);
    setBlock(J, O_P, O_P, RiT);
    setBlock(J, O_R, O_R, LeftQuatTL3(cq.inverse() * Qi.inverse() * Qj));
    whiten(J, 7, jac[2]);
  }
  if (jac[3]) {
    Mat J(15, 9);
    setBlock(J, O_V, O_V - O_V, RiT);
    setBlock(J, O_BA, O_BA - O_V, M3d::Identity());
    setBlock(J, O_BG, O_BG - O_V, M3d::Identity());
    whiten(J, 9, jac[3]);
  }
  return true;
}

// PivotPointPlaneFactor.cc:43-137 (sqrt_info_static = 1.0, :35).  Jacobians 1x7 row-major, col 6 = 0.
inline bool PivotPointPlaneEvaluate(const V3d &point, const double coeff[4], const double *const *par, double *residuals,
                                    double **jac) {
  V3d Pp, Pi, tlb; Qd Qp, Qi, qlb;
  unpackPose(par[0], Pp, Qp); unpackPose(par[1], Pi, Qi); unpackPose(par[2], tlb, qlb);
  Qd Qlp = Qp * qlb.conjugate();
  V3d Plp = Pp - Qlp * tlb;
  Qd Qli = Qi * qlb.conjugate();
  V3d Pli = Pi - Qli * tlb;
  Qd Qlpi = Qlp.conjugate() * Qli;
  V3d Plpi = Qlp.conjugate() * (Pli - Plp);
  V3d w(coeff[0], coeff[1], coeff[2]);
  double b = coeff[3];
  residuals[0] = w.dot(Qlpi * point + Plpi) + b;
  if (!jac) return true;
  M3d Ri = Qi.toRotationMatrix(), Rp = Qp.toRotationMatrix(), rlb = qlb.toRotationMatrix();
  auto rowTimes = [](const V3d &r, const M3d &M) {  // r^T M
    return V3d(r.x * M(0, 0) + r.y * M(1, 0) + r.z * M(2, 0), r.x * M(0, 1) + r.y * M(1, 1) + r.z * M(2, 1),
               r.x * M(0, 2) + r.y * M(1, 2) + r.z * M(2, 2));
  };
  if (jac[0]) {
    V3d l = -rowTimes(w, rlb * Rp.transpose());
    V3d rr = rowTimes(w, rlb * (Skew(Rp.transpose() * (Ri * (rlb.transpose() * (point - tlb)))) + Skew(Rp.transpose() * (Pi - Pp))));
    double *J = jac[0];
    J[0] = l.x; J[1] = l.y; J[2] = l.z; J[3] = rr.x; J[4] = rr.y; J[5] = rr.z; J[6] = 0;
  }
  if (jac[1]) {
    V3d l = rowTimes(w, rlb * Rp.transpose());
    V3d rr = rowTimes(w, ((rlb * Rp.transpose()) * Ri) * (-Skew(rlb.transpose() * point) + Skew(rlb.transpose() * tlb)));
    double *J = jac[1];
    J[0] = l.x; J[1] = l.y; J[2] = l.z; J[3] = rr.x; J[4] = rr.y; J[5] = rr.z; J[6] = 0;
  }
  if (jac[2]) {
    M3d I = M3d::Identity();
    M3d RpTRi = Rp.transpose() * Ri;
    V3d l = rowTimes(w, I - (rlb * RpTRi) * rlb.transpose());
    V3d q = rlb.transpose() * (point - tlb);
    V3d rr = rowTimes(w, rlb * (-Skew(RpTRi * q) + RpTRi * Skew(q) - Skew(Rp.transpose() * (Pi - Pp))));
    double *J = jac[2];
    J[0] = l.x; J[1] = l.y; J[2] = l.z; J[3] = rr.x; J[4] = rr.y; J[5] = rr.z; J[6] = 0;
  }
  return true;
}

// PriorFactor.cc:35-67: sqrt_info = diag(1000 I3, 0.1 I3)
inline bool PriorFactorEvaluate(const V3d &pos0, const Qd &rot0, const double *pose, double *res, double *J67) {
  V3d P; Qd Qq;
  unpackPose(pose, P, Qq);
  V3d dp = P - pos0;
  V3d dr = 2.0 * (rot0.inverse() * Qq).vec();
  for (int k = 0; k < 3; ++k) { res[k] = 1000.0 * dp[k]; res[3 + k] = 0.1 * dr[k]; }
  if (J67) {
    for (int i = 0; i < 42; ++i) J67[i] = 0;
    M3d B = LeftQuatTL3(Qq.inverse() * rot0);
    for (int i = 0; i < 3; ++i) {
      J67[i * 7 + i] = 1000.0;
      for (int j = 0; j < 3; ++j) J67[(3 + i) * 7 + 3 + j] = 0.1 * B(i, j);
    }
  }
  return true;
}

// PoseLocalParameterization.cc:35-50
inline void PosePlus(const double *x, const double *d, double *out) {
  Qd q(x[6], x[3], x[4], x[5]);
  Qd dq = DeltaQ(V3d(d[3], d[4], d[5]));
  Qd qn = (q * dq).normalized();
  out[0] = x[0] + d[0]; out[1] = x[1] + d[1]; out[2] = x[2] + d[2];
  out[3] = qn.x; out[4] = qn.y; out[5] = qn.z; out[6] = qn.w;
}

}  // namespace orc
