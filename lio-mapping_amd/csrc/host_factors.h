// host_factors.h — host-side (C++) factors of the product.
//   Preintegration  <- IntegrationBase            include/imu_processor/IntegrationBase.h:77-357
//   imu_factor      <- ImuFactor::Evaluate        include/factor/ImuFactor.h:53-168
//   ppp_factor      <- PivotPointPlaneFactor      src/factor/PivotPointPlaneFactor.cc:43-137
//   prior_factor    <- PriorFactor::Evaluate      src/factor/PriorFactor.cc:35-67
//   pose_plus       <- PoseLocalParameterization  src/factor/PoseLocalParameterization.cc:35-50
// IMU work is sequential 15x15 recurrences at 200 Hz and Wo factors per solve: it stays on the host
// (SURVEY.md §2.1); the per-point lidar factors run on the GPU (solve_kernels.hip) and only their
// 18-column structure (ppp_factor at basis inputs) is taken from here.
#pragma once
#include <cstring>
#include <memory>
#include <vector>

#include "hlinalg.h"
#include "hmath.h"

namespace lio {

typedef Vec3<double> V3d;
typedef Mat3<double> M3d;
typedef Quat<double> Qd;

struct PimNoise { double acc_n = 0.1, gyr_n = 0.01, acc_w = 0.0002, gyr_w = 2.0e-5, g_norm = 9.805; };

enum { kOP = 0, kOR = 3, kOV = 6, kOBA = 9, kOBG = 12 };

LIO_HD void put3(double *M, int ld, int r, int c, const M3d &B) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[(r + i) * ld + c + j] = B(i, j);
}
LIO_HD M3d get3(const double *M, int ld, int r, int c) {
  M3d B;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) B(i, j) = M[(r + i) * ld + c + j];
  return B;
}

// The frozen quantities of a pre-integration that the factor reads, as a POD view (also what the device-resident solver
// is handed: solve_step.h).  jac points at the 15x15 row-major bias Jacobian.
struct PimCore {
  double dp[3], dq[4] /* w, x, y, z */, dv[3], ba[3], bg[3], g[3], sum_dt;
  const double *jac;
};

// IntegrationBase::Evaluate (IntegrationBase.h:309-357): unwhitened 15-d residual
LIO_HD void pim_residual(const PimCore &c, const V3d &Pi, const Qd &Qi, const V3d &Vi, const V3d &Bai, const V3d &Bgi, const V3d &Pj,
                         const Qd &Qj, const V3d &Vj, const V3d &Baj, const V3d &Bgj, double r[15]) {
  const V3d dp(c.dp[0], c.dp[1], c.dp[2]), dv(c.dv[0], c.dv[1], c.dv[2]), ba(c.ba[0], c.ba[1], c.ba[2]), bg(c.bg[0], c.bg[1], c.bg[2]);
  const V3d g_vec(c.g[0], c.g[1], c.g[2]);
  const Qd dq(c.dq[0], c.dq[1], c.dq[2], c.dq[3]);
  const double sum_dt = c.sum_dt;
  M3d dp_dba = get3(c.jac, 15, kOP, kOBA), dp_dbg = get3(c.jac, 15, kOP, kOBG), dq_dbg = get3(c.jac, 15, kOR, kOBG);
  M3d dv_dba = get3(c.jac, 15, kOV, kOBA), dv_dbg = get3(c.jac, 15, kOV, kOBG);
  V3d dba = Bai - ba, dbg = Bgi - bg;
  Qd cq = dq * deltaQ(dq_dbg * dbg);
  V3d cv = dv + dv_dba * dba + dv_dbg * dbg;
  V3d cp = dp + dp_dba * dba + dp_dbg * dbg;
  Qd Qii = qinverse(Qi);
  V3d rp = rotate(Qii, (-0.5) * g_vec * sum_dt * sum_dt + Pj - Pi - Vi * sum_dt) - cp;
  V3d rr = 2.0 * (qinverse(cq) * (Qii * Qj)).vec();
  V3d rv = rotate(Qii, (-1.0) * g_vec * sum_dt + Vj - Vi) - cv;
  V3d rba = Baj - Bai, rbg = Bgj - Bgi;
  for (int k = 0; k < 3; ++k) { r[kOP + k] = rp[k]; r[kOR + k] = rr[k]; r[kOV + k] = rv[k]; r[kOBA + k] = rba[k]; r[kOBG + k] = rbg[k]; }
}

class Preintegration {
 public:
  V3d acc0, gyr0, lin_acc, lin_gyr, ba, bg, dp, dv, g_vec;
  Qd dq;
  double jac[225], cov[225];
  double sum_dt = 0;
  PimNoise noise;
  std::vector<double> dt_buf;
  std::vector<V3d> acc_buf, gyr_buf;

  Preintegration(const V3d &a0, const V3d &g0, const V3d &ba_, const V3d &bg_, const PimNoise &n)
      : acc0(a0), gyr0(g0), lin_acc(a0), lin_gyr(g0), ba(ba_), bg(bg_), noise(n) {
    g_vec = V3d(0, 0, -n.g_norm);
    reset_state();
  }
  void reset_state() {
    std::memset(jac, 0, sizeof(jac)); std::memset(cov, 0, sizeof(cov));
    for (int i = 0; i < 15; ++i) jac[i * 15 + i] = 1.0;
    dp = V3d(); dv = V3d(); dq = Qd(); sum_dt = 0; sqrt_info_valid_ = false;
  }
  void push_back(double dt, const V3d &a, const V3d &g) {
    dt_buf.push_back(dt); acc_buf.push_back(a); gyr_buf.push_back(g);
    propagate(dt, a, g);
  }
  void repropagate(const V3d &nba, const V3d &nbg) {
    acc0 = lin_acc; gyr0 = lin_gyr; ba = nba; bg = nbg;
    reset_state();
    for (size_t i = 0; i < dt_buf.size(); ++i) propagate(dt_buf[i], acc_buf[i], gyr_buf[i]);
  }
  // mid-point integration with the reference's constants
  void propagate(double dt, const V3d &a1, const V3d &g1) {
    sqrt_info_valid_ = false;
    V3d un_acc_0 = rotate(dq, acc0 - ba);
    V3d un_gyr = 0.5 * (gyr0 + g1) - bg;
    Qd rq = dq * Qd(1, un_gyr.x * dt / 2, un_gyr.y * dt / 2, un_gyr.z * dt / 2);
    V3d un_acc_1 = rotate(rq, a1 - ba);
    V3d un_acc = 0.5 * (un_acc_0 + un_acc_1);
    V3d rp = dp + dv * dt + 0.5 * un_acc * dt * dt;
    V3d rv = dv + un_acc * dt;
    M3d Rw = skew(un_gyr), Ra0 = skew(acc0 - ba), Ra1 = skew(a1 - ba);
    M3d I = M3d::identity(), Rq = toRot(dq), Rr = toRot(rq);
    double F[225] = {0}, V[15 * 18] = {0};
    put3(F, 15, 0, 0, I);
    put3(F, 15, 0, 3, (Rq * Ra0) * (-0.25) * dt * dt + ((Rr * Ra1) * (I - Rw * dt)) * (-0.25) * dt * dt);
    put3(F, 15, 0, 6, I * dt);
    put3(F, 15, 0, 9, (Rq + Rr) * (-0.25) * dt * dt);
    put3(F, 15, 0, 12, (Rr * Ra1) * (-0.1667) * dt * dt * (-dt));
    put3(F, 15, 3, 3, I - Rw * dt);
    put3(F, 15, 3, 12, I * (-1.0) * dt);
    put3(F, 15, 6, 3, (Rq * Ra0) * (-0.5) * dt + ((Rr * Ra1) * (I - Rw * dt)) * (-0.5) * dt);
    put3(F, 15, 6, 6, I);
    put3(F, 15, 6, 9, (Rq + Rr) * (-0.5) * dt);
    put3(F, 15, 6, 12, (Rr * Ra1) * (-0.5) * dt * (-dt));
    put3(F, 15, 9, 9, I);
    put3(F, 15, 12, 12, I);
    put3(V, 18, 0, 0, Rq * 0.5 * dt * dt);
    M3d v03 = (((-Rr) * 0.25) * Ra1) * dt * dt * 0.5 * dt;
    put3(V, 18, 0, 3, v03);
    put3(V, 18, 0, 6, Rr * 0.5 * dt * dt);
    put3(V, 18, 0, 9, v03);
    put3(V, 18, 3, 3, I * 0.5 * dt);
    put3(V, 18, 3, 9, I * 0.5 * dt);
    put3(V, 18, 6, 0, Rq * 0.5 * dt);
    M3d v63 = (((-Rr) * 0.5) * Ra1) * dt * 0.5 * dt;
    put3(V, 18, 6, 3, v63);
    put3(V, 18, 6, 6, Rr * 0.5 * dt);
    put3(V, 18, 6, 9, v63);
    put3(V, 18, 9, 12, I * dt);
    put3(V, 18, 12, 15, I * dt);
    double q18[18];
    double an = noise.acc_n * noise.acc_n, gn = noise.gyr_n * noise.gyr_n, aw = noise.acc_w * noise.acc_w, gw = noise.gyr_w * noise.gyr_w;
    for (int i = 0; i < 3; ++i) { q18[i] = an; q18[3 + i] = gn; q18[6 + i] = an; q18[9 + i] = gn; q18[12 + i] = aw; q18[15 + i] = gw; }
    // J <- F J, P <- F P F^T + V Q V^T (IntegrationBase.h:205-206).  Written as row updates (axpy over the contiguous j) so
    // the compiler vectorises them, with the structural zeros of F (144 of 225) and V (186 of 270) skipped; every element is
    // still the sum of the same products in ascending k, i.e. bit-identical to the plain triple loop of the oracle.
    double nj[225] = {0}, FP[225] = {0}, nc[225] = {0}, vq[225] = {0}, FT[225], VT[18 * 15];
    for (int i = 0; i < 15; ++i) {
      for (int k = 0; k < 15; ++k) FT[k * 15 + i] = F[i * 15 + k];
      for (int k = 0; k < 18; ++k) VT[k * 15 + i] = V[i * 18 + k];
    }
    for (int i = 0; i < 15; ++i)
      for (int k = 0; k < 15; ++k) {
        const double f = F[i * 15 + k];
        if (f == 0.0) continue;
        double *o1 = nj + i * 15, *o2 = FP + i * 15;
        const double *r1 = jac + k * 15, *r2 = cov + k * 15;
        for (int j = 0; j < 15; ++j) { o1[j] += f * r1[j]; o2[j] += f * r2[j]; }
      }
    for (int i = 0; i < 15; ++i) {
      double *o = nc + i * 15, *ov = vq + i * 15;
      for (int k = 0; k < 15; ++k) {
        const double f = FP[i * 15 + k];
        const double *r = FT + k * 15;
        for (int j = 0; j < 15; ++j) o[j] += f * r[j];
      }
      for (int k = 0; k < 18; ++k) {
        const double w = V[i * 18 + k];
        if (w == 0.0) continue;
        const double wq = w * q18[k];
        const double *r = VT + k * 15;
        for (int j = 0; j < 15; ++j) ov[j] += wq * r[j];
      }
      for (int j = 0; j < 15; ++j) o[j] = o[j] + ov[j];
    }
    std::memcpy(jac, nj, sizeof(jac)); std::memcpy(cov, nc, sizeof(cov));
    dp = rp; dq = normalized(rq); dv = rv;
    sum_dt += dt;
    acc0 = a1; gyr0 = g1;
  }
  PimCore core() const {
    PimCore c;
    for (int k = 0; k < 3; ++k) { c.dp[k] = dp[k]; c.dv[k] = dv[k]; c.ba[k] = ba[k]; c.bg[k] = bg[k]; c.g[k] = g_vec[k]; }
    c.dq[0] = dq.w; c.dq[1] = dq.x; c.dq[2] = dq.y; c.dq[3] = dq.z;
    c.sum_dt = sum_dt; c.jac = jac;
    return c;
  }
  void evaluate(const V3d &Pi, const Qd &Qi, const V3d &Vi, const V3d &Bai, const V3d &Bgi, const V3d &Pj, const Qd &Qj, const V3d &Vj,
                const V3d &Baj, const V3d &Bgj, double r[15]) const {
    pim_residual(core(), Pi, Qi, Vi, Bai, Bgi, Pj, Qj, Vj, Baj, Bgj, r);
  }
  // upper-triangular whitening matrix L^T with L L^T = cov^-1; cached: cov is frozen once the frame is pushed
  const double *sqrt_info() const {
    if (!sqrt_info_valid_) {
      double cinv[225];
      sqrt_info_ok_ = gj_inverse(cov, 15, cinv) && chol_factor(cinv, 15, 15);
      for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) sqrt_info_[i * 15 + j] = (j >= i) ? cinv[j * 15 + i] : 0.0;
      sqrt_info_valid_ = true;
    }
    return sqrt_info_ok_ ? sqrt_info_ : nullptr;
  }

 private:
  mutable double sqrt_info_[225];
  mutable bool sqrt_info_valid_ = false, sqrt_info_ok_ = false;
};

LIO_HD void unpack_pose(const double *p, V3d &P, Qd &Q) { P = V3d(p[0], p[1], p[2]); Q = Qd(p[6], p[3], p[4], p[5]); }

// 4x4 quaternion matrices, top-left 3x3 of Left(a) * Right(b)
LIO_HD M3d left_tl3(const Qd &q) { return M3d::identity() * q.w + skew(q.vec()); }
LIO_HD M3d left_right_tl3(const Qd &a, const Qd &b) {
  double L[4][4], R[4][4];
  M3d la = left_tl3(a), rb = M3d::identity() * b.w - skew(b.vec());
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { L[i][j] = la(i, j); R[i][j] = rb(i, j); }
  V3d va = a.vec(), vb = b.vec();
  for (int j = 0; j < 3; ++j) { L[3][j] = -va[j]; L[j][3] = va[j]; R[3][j] = -vb[j]; R[j][3] = vb[j]; }
  L[3][3] = a.w; R[3][3] = b.w;
  M3d out;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += L[i][k] * R[k][j]; out(i, j) = s; }
  return out;
}

// Unwhitened Jacobian block `which` of ImuFactor::Evaluate (ImuFactor.h:88-160) in AMBIENT layout, row-major, zero-filled:
// 0: d r / d pose_i (15x7), 1: d r / d speed-bias_i (15x9), 2: d r / d pose_j (15x7), 3: d r / d speed-bias_j (15x9).
LIO_HD void imu_raw_jacobian(const PimCore &c, int which, const double *pose_i, const double *sb_i, const double *pose_j, const double *sb_j,
                             double *J) {
  V3d Pi, Pj; Qd Qi, Qj;
  unpack_pose(pose_i, Pi, Qi); unpack_pose(pose_j, Pj, Qj);
  const V3d Vi(sb_i[0], sb_i[1], sb_i[2]), Bgi(sb_i[6], sb_i[7], sb_i[8]);
  const V3d Vj(sb_j[0], sb_j[1], sb_j[2]);
  const double sum_dt = c.sum_dt;
  const V3d g(c.g[0], c.g[1], c.g[2]), pbg(c.bg[0], c.bg[1], c.bg[2]);
  const Qd pdq(c.dq[0], c.dq[1], c.dq[2], c.dq[3]);
  M3d dp_dba = get3(c.jac, 15, kOP, kOBA), dp_dbg = get3(c.jac, 15, kOP, kOBG), dq_dbg = get3(c.jac, 15, kOR, kOBG);
  M3d dv_dba = get3(c.jac, 15, kOV, kOBA), dv_dbg = get3(c.jac, 15, kOV, kOBG);
  Qd Qii = qinverse(Qi);
  M3d RiT = toRot(Qii);
  Qd cq = pdq * deltaQ(dq_dbg * (Bgi - pbg));
  const int cols = (which & 1) ? 9 : 7;
  for (int k = 0; k < 15 * cols; ++k) J[k] = 0.0;
  if (which == 0) {
    put3(J, 7, kOP, 0, -RiT);
    put3(J, 7, kOP, 3, skew(rotate(Qii, (-0.5) * g * sum_dt * sum_dt + Pj - Pi - Vi * sum_dt)));
    put3(J, 7, kOR, 3, -left_right_tl3(qinverse(Qj) * Qi, cq));
    put3(J, 7, kOV, 3, skew(rotate(Qii, (-1.0) * g * sum_dt + Vj - Vi)));
  } else if (which == 1) {
    put3(J, 9, kOP, 0, -RiT * sum_dt);
    put3(J, 9, kOP, 3, -dp_dba);
    put3(J, 9, kOP, 6, -dp_dbg);
    put3(J, 9, kOR, 6, -left_tl3(qinverse(Qj) * Qi * cq) * dq_dbg);
    put3(J, 9, kOV, 0, -RiT);
    put3(J, 9, kOV, 3, -dv_dba);
    put3(J, 9, kOV, 6, -dv_dbg);
    put3(J, 9, kOBA, 3, -M3d::identity());
    put3(J, 9, kOBG, 6, -M3d::identity());
  } else if (which == 2) {
    put3(J, 7, kOP, 0, RiT);
    put3(J, 7, kOR, 3, left_tl3(qinverse(cq) * Qii * Qj));
  } else {
    put3(J, 9, kOV, 0, RiT);
    put3(J, 9, kOBA, 3, M3d::identity());
    put3(J, 9, kOBG, 6, M3d::identity());
  }
}

// Whitened residual (15) and Jacobians in AMBIENT layout (15x7, 15x9, 15x7, 15x9; null = skip).
inline bool imu_factor(const Preintegration &pim, const double *pose_i, const double *sb_i, const double *pose_j, const double *sb_j,
                       double *res, double *J0, double *J1, double *J2, double *J3) {
  V3d Pi, Pj; Qd Qi, Qj;
  unpack_pose(pose_i, Pi, Qi); unpack_pose(pose_j, Pj, Qj);
  V3d Vi(sb_i[0], sb_i[1], sb_i[2]), Bai(sb_i[3], sb_i[4], sb_i[5]), Bgi(sb_i[6], sb_i[7], sb_i[8]);
  V3d Vj(sb_j[0], sb_j[1], sb_j[2]), Baj(sb_j[3], sb_j[4], sb_j[5]), Bgj(sb_j[6], sb_j[7], sb_j[8]);
  double r[15];
  const PimCore core = pim.core();
  pim_residual(core, Pi, Qi, Vi, Bai, Bgi, Pj, Qj, Vj, Baj, Bgj, r);
  const double *S = pim.sqrt_info();
  if (!S) return false;
  for (int i = 0; i < 15; ++i) { double s = 0; for (int k = i; k < 15; ++k) s += S[i * 15 + k] * r[k]; res[i] = s; }
  if (!J0 && !J1 && !J2 && !J3) return true;
  auto whiten = [&](const double *J, int cols, double *out) {
    // row-axpy form (same k-ascending summation order as the dot form, but the inner loop is contiguous
    // and vectorises under strict IEEE semantics)
    for (int i = 0; i < 15; ++i) {
      double *o = out + i * cols;
      for (int j = 0; j < cols; ++j) o[j] = 0.0;
      for (int k = i; k < 15; ++k) {
        const double f = S[i * 15 + k];
        const double *jr = J + k * cols;
        for (int j = 0; j < cols; ++j) o[j] += f * jr[j];
      }
    }
  };
  double *outs[4] = {J0, J1, J2, J3};
  for (int which = 0; which < 4; ++which) {
    if (!outs[which]) continue;
    double J[135];
    imu_raw_jacobian(core, which, pose_i, sb_i, pose_j, sb_j, J);
    whiten(J, (which & 1) ? 9 : 7, outs[which]);
  }
  return true;
}

// ---- the same factor as ONE block of the normal equations (what WindowSystem::evaluate adds per IMU factor), for hosts with
// AVX-512.  imu_raw_local forms the unwhitened residual and the 15 x 30 Jacobian in the LOCAL column layout
// [pose_i 6 | speed-bias_i 9 | pose_j 6 | speed-bias_j 9] (padded to 32) with the shared quantities — R_i^T, the corrected delta
// rotation, the bias Jacobians — computed once (imu_raw_jacobian above forms them per block); the entries are the expressions of
// imu_raw_jacobian, the local columns of a pose being the first six ambient ones.  imu_block_avx512 whitens with the upper
// factor S of the information matrix (rows of 4 vectors), then H_b = J_w^T J_w and g_b = J_w^T r_w in register tiles of 6 x 4
// vectors.  The whitened residual is summed exactly as imu_factor sums it, so the cost is bit-identical; H_b and g_b differ from
// the scalar path in the last bits (fused multiply-adds).
#define LIO_IMU_LD 32
inline void imu_raw_local(const PimCore &c, const double *pose_i, const double *sb_i, const double *pose_j, const double *sb_j, double r[15],
                          double *J /* 15 x LIO_IMU_LD, zero-filled here */) {
  V3d Pi, Pj; Qd Qi, Qj;
  unpack_pose(pose_i, Pi, Qi); unpack_pose(pose_j, Pj, Qj);
  const V3d Vi(sb_i[0], sb_i[1], sb_i[2]), Bai(sb_i[3], sb_i[4], sb_i[5]), Bgi(sb_i[6], sb_i[7], sb_i[8]);
  const V3d Vj(sb_j[0], sb_j[1], sb_j[2]), Baj(sb_j[3], sb_j[4], sb_j[5]), Bgj(sb_j[6], sb_j[7], sb_j[8]);
  pim_residual(c, Pi, Qi, Vi, Bai, Bgi, Pj, Qj, Vj, Baj, Bgj, r);
  const double sum_dt = c.sum_dt;
  const V3d g(c.g[0], c.g[1], c.g[2]), pbg(c.bg[0], c.bg[1], c.bg[2]);
  const Qd pdq(c.dq[0], c.dq[1], c.dq[2], c.dq[3]);
  const M3d dp_dba = get3(c.jac, 15, kOP, kOBA), dp_dbg = get3(c.jac, 15, kOP, kOBG), dq_dbg = get3(c.jac, 15, kOR, kOBG);
  const M3d dv_dba = get3(c.jac, 15, kOV, kOBA), dv_dbg = get3(c.jac, 15, kOV, kOBG);
  const Qd Qii = qinverse(Qi);
  const M3d RiT = toRot(Qii);
  const Qd cq = pdq * deltaQ(dq_dbg * (Bgi - pbg));
  for (int k = 0; k < 15 * LIO_IMU_LD; ++k) J[k] = 0.0;
  const int cPi = 0, cRi = 3, cVi = 6, cBai = 9, cBgi = 12, cPj = 15, cRj = 18, cVj = 21, cBaj = 24, cBgj = 27;
  const M3d I = M3d::identity();
  // d r / d pose_i
  put3(J, LIO_IMU_LD, kOP, cPi, -RiT);
  put3(J, LIO_IMU_LD, kOP, cRi, skew(rotate(Qii, (-0.5) * g * sum_dt * sum_dt + Pj - Pi - Vi * sum_dt)));
  put3(J, LIO_IMU_LD, kOR, cRi, -left_right_tl3(qinverse(Qj) * Qi, cq));
  put3(J, LIO_IMU_LD, kOV, cRi, skew(rotate(Qii, (-1.0) * g * sum_dt + Vj - Vi)));
  // d r / d speed-bias_i
  put3(J, LIO_IMU_LD, kOP, cVi, -RiT * sum_dt);
  put3(J, LIO_IMU_LD, kOP, cBai, -dp_dba);
  put3(J, LIO_IMU_LD, kOP, cBgi, -dp_dbg);
  put3(J, LIO_IMU_LD, kOR, cBgi, -left_tl3(qinverse(Qj) * Qi * cq) * dq_dbg);
  put3(J, LIO_IMU_LD, kOV, cVi, -RiT);
  put3(J, LIO_IMU_LD, kOV, cBai, -dv_dba);
  put3(J, LIO_IMU_LD, kOV, cBgi, -dv_dbg);
  put3(J, LIO_IMU_LD, kOBA, cBai, -I);
  put3(J, LIO_IMU_LD, kOBG, cBgi, -I);
  // d r / d pose_j
  put3(J, LIO_IMU_LD, kOP, cPj, RiT);
  put3(J, LIO_IMU_LD, kOR, cRj, left_tl3(qinverse(cq) * Qii * Qj));
  // d r / d speed-bias_j
  put3(J, LIO_IMU_LD, kOV, cVj, RiT);
  put3(J, LIO_IMU_LD, kOBA, cBaj, I);
  put3(J, LIO_IMU_LD, kOBG, cBgj, I);
}

// Hb: 30 x LIO_IMU_LD (columns 30, 31 zero), gb: LIO_IMU_LD, rw: the whitened residual (15)
__attribute__((target("avx512f,fma"))) inline void imu_block_avx512(const double *S /* 15 x 15 upper */, const double *J /* 15 x LIO_IMU_LD */,
                                                                      const double *r, double *Hb, double *gb, double *rw) {
  alignas(64) double Jw[15 * LIO_IMU_LD];
  for (int i = 0; i < 15; ++i) {
    __m512d a0 = _mm512_setzero_pd(), a1 = a0, a2 = a0, a3 = a0;
    double sres = 0;
    for (int k = i; k < 15; ++k) {
      const double f = S[i * 15 + k];
      const __m512d fv = _mm512_set1_pd(f);
      const double *jr = J + k * LIO_IMU_LD;
      a0 = _mm512_fmadd_pd(fv, _mm512_loadu_pd(jr), a0); a1 = _mm512_fmadd_pd(fv, _mm512_loadu_pd(jr + 8), a1);
      a2 = _mm512_fmadd_pd(fv, _mm512_loadu_pd(jr + 16), a2); a3 = _mm512_fmadd_pd(fv, _mm512_loadu_pd(jr + 24), a3);
      sres += f * r[k];
    }
    double *o = Jw + i * LIO_IMU_LD;
    _mm512_store_pd(o, a0); _mm512_store_pd(o + 8, a1); _mm512_store_pd(o + 16, a2); _mm512_store_pd(o + 24, a3);
    rw[i] = sres;
  }
  {
    __m512d g0 = _mm512_setzero_pd(), g1 = g0, g2 = g0, g3 = g0;
    for (int k = 0; k < 15; ++k) {
      const __m512d fv = _mm512_set1_pd(rw[k]);
      const double *jr = Jw + k * LIO_IMU_LD;
      g0 = _mm512_fmadd_pd(fv, _mm512_load_pd(jr), g0); g1 = _mm512_fmadd_pd(fv, _mm512_load_pd(jr + 8), g1);
      g2 = _mm512_fmadd_pd(fv, _mm512_load_pd(jr + 16), g2); g3 = _mm512_fmadd_pd(fv, _mm512_load_pd(jr + 24), g3);
    }
    _mm512_storeu_pd(gb, g0); _mm512_storeu_pd(gb + 8, g1); _mm512_storeu_pd(gb + 16, g2); _mm512_storeu_pd(gb + 24, g3);
  }
  for (int a0 = 0; a0 < 30; a0 += 6) {
    __m512d acc[6][4];
#pragma GCC unroll 6
    for (int q = 0; q < 6; ++q) { acc[q][0] = _mm512_setzero_pd(); acc[q][1] = acc[q][0]; acc[q][2] = acc[q][0]; acc[q][3] = acc[q][0]; }
    for (int k = 0; k < 15; ++k) {
      const double *jr = Jw + k * LIO_IMU_LD;
      const __m512d v0 = _mm512_load_pd(jr), v1 = _mm512_load_pd(jr + 8), v2 = _mm512_load_pd(jr + 16), v3 = _mm512_load_pd(jr + 24);
#pragma GCC unroll 6
      for (int q = 0; q < 6; ++q) {
        const __m512d fv = _mm512_set1_pd(jr[a0 + q]);
        acc[q][0] = _mm512_fmadd_pd(fv, v0, acc[q][0]); acc[q][1] = _mm512_fmadd_pd(fv, v1, acc[q][1]);
        acc[q][2] = _mm512_fmadd_pd(fv, v2, acc[q][2]); acc[q][3] = _mm512_fmadd_pd(fv, v3, acc[q][3]);
      }
    }
#pragma GCC unroll 6
    for (int q = 0; q < 6; ++q) {
      double *o = Hb + (a0 + q) * LIO_IMU_LD;
      _mm512_storeu_pd(o, acc[q][0]); _mm512_storeu_pd(o + 8, acc[q][1]); _mm512_storeu_pd(o + 16, acc[q][2]); _mm512_storeu_pd(o + 24, acc[q][3]);
    }
  }
}

// residual + 1x7 Jacobians (null = skip)
// The pose-dependent part of PivotPointPlaneFactor::Evaluate (PivotPointPlaneFactor.cc:58-70 and the rotation matrices of
// :85-128), shared by every residual of a (pivot, frame i, extrinsic) triple.
struct PppPoses {
  V3d Pp, Pi, tlb, Plpi, dP;       // dP = Pi - Pp
  Qd Qlpi;
  M3d Ri, RpT, rlb, rlbT;
};
LIO_HD PppPoses ppp_prepare(const double *pose_p, const double *pose_i, const double *pose_ex) {
  PppPoses t;
  Qd Qp, Qi, qlb;
  unpack_pose(pose_p, t.Pp, Qp); unpack_pose(pose_i, t.Pi, Qi); unpack_pose(pose_ex, t.tlb, qlb);
  Qd Qlp = Qp * conj(qlb);
  V3d Plp = t.Pp - rotate(Qlp, t.tlb);
  Qd Qli = Qi * conj(qlb);
  V3d Pli = t.Pi - rotate(Qli, t.tlb);
  t.Qlpi = conj(Qlp) * Qli;
  t.Plpi = rotate(conj(Qlp), Pli - Plp);
  t.Ri = toRot(Qi); t.rlb = toRot(qlb);
  t.RpT = transpose(toRot(Qp)); t.rlbT = transpose(t.rlb);
  t.dP = t.Pi - t.Pp;
  return t;
}
LIO_HD void ppp_eval(const PppPoses &t, const V3d &point, const double coeff[4], double *res, double *Jp, double *Ji, double *Jex) {
  V3d w(coeff[0], coeff[1], coeff[2]);
  *res = dot(w, rotate(t.Qlpi, point) + t.Plpi) + coeff[3];
  if (!Jp && !Ji && !Jex) return;
  const M3d &Ri = t.Ri, &RpT = t.RpT, &rlb = t.rlb, &rlbT = t.rlbT;
  const V3d &tlb = t.tlb;
  auto put = [](double *J, const V3d &l, const V3d &r) { J[0] = l.x; J[1] = l.y; J[2] = l.z; J[3] = r.x; J[4] = r.y; J[5] = r.z; J[6] = 0; };
  if (Jp) put(Jp, -rowmul(w, rlb * RpT), rowmul(w, rlb * (skew(RpT * (Ri * (rlbT * (point - tlb)))) + skew(RpT * t.dP))));
  if (Ji) put(Ji, rowmul(w, rlb * RpT), rowmul(w, ((rlb * RpT) * Ri) * (-skew(rlbT * point) + skew(rlbT * tlb))));
  if (Jex) {
    M3d RpTRi = RpT * Ri;
    V3d q = rlbT * (point - tlb);
    put(Jex, rowmul(w, M3d::identity() - (rlb * RpTRi) * rlbT), rowmul(w, rlb * (-skew(RpTRi * q) + RpTRi * skew(q) - skew(RpT * t.dP))));
  }
}
LIO_HD void ppp_factor(const V3d &point, const double coeff[4], const double *pose_p, const double *pose_i, const double *pose_ex,
                       double *res, double *Jp, double *Ji, double *Jex) {
  ppp_eval(ppp_prepare(pose_p, pose_i, pose_ex), point, coeff, res, Jp, Ji, Jex);
}

LIO_HD void prior_factor(const V3d &pos0, const Qd &rot0, const double *pose, double *res, double *J67) {
  V3d P; Qd Q;
  unpack_pose(pose, P, Q);
  V3d dp = P - pos0;
  V3d dr = 2.0 * (qinverse(rot0) * Q).vec();
  for (int k = 0; k < 3; ++k) { res[k] = 1000.0 * dp[k]; res[3 + k] = 0.1 * dr[k]; }
  if (J67) {
    for (int k = 0; k < 42; ++k) J67[k] = 0.0;
    M3d B = left_tl3(qinverse(Q) * rot0);  // as written at PriorFactor.cc:56 (skew sign differs from the true derivative)
    for (int i = 0; i < 3; ++i) {
      J67[i * 7 + i] = 1000.0;
      for (int j = 0; j < 3; ++j) J67[(3 + i) * 7 + 3 + j] = 0.1 * B(i, j);
    }
  }
}

LIO_HD void pose_plus(const double *x, const double *d, double *out) {
  Qd q(x[6], x[3], x[4], x[5]);
  Qd qn = normalized(q * deltaQ(V3d(d[3], d[4], d[5])));
  out[0] = x[0] + d[0]; out[1] = x[1] + d[1]; out[2] = x[2] + d[2];
  out[3] = qn.x; out[4] = qn.y; out[5] = qn.z; out[6] = qn.w;
}

}  // namespace lio
