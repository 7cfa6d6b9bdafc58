// C entry points over the REFERENCE's own factor math, compiled from the sources where they lie under /root/reference
// (include/imu_processor/IntegrationBase.h, include/factor/ImuFactor.h, src/factor/PivotPointPlaneFactor.cc,
// src/factor/PriorFactor.cc, src/factor/PoseLocalParameterization.cc, include/utils/math_utils.h) against the stand-in headers of
// oracle/ref_shim (Eigen's dense API subset, Ceres' two base classes, the ROS/glog logging macros).  TEST INFRASTRUCTURE: built by
// `make -C oracle ref` into oracle/_ref/libref_factors.so when /root/reference exists; tests/golden/make_ref_factor_vectors.py
// turns its outputs into committed vectors, tests/test_oracle_ref_factors.py checks the oracle (and with it the product's host
// code, which is bit-identical to the oracle) against them.  Nothing of the reference is copied: this file only calls it.
#include <memory>

#include "factor/ImuFactor.h"
#include "factor/PivotPointPlaneFactor.h"
#include "factor/PoseLocalParameterization.h"
#include "factor/PriorFactor.h"
#include "imu_processor/IntegrationBase.h"

namespace {
Eigen::Vector3d v3(const double *p) { return Eigen::Vector3d(p[0], p[1], p[2]); }
Eigen::Quaterniond qxyzw(const double *p) { return Eigen::Quaterniond(p[3], p[0], p[1], p[2]); }
struct Pim { std::shared_ptr<lio::IntegrationBase> p; };
}  // namespace

extern "C" {

// noise = acc_n, gyr_n, acc_w, gyr_w, g_norm (IntegrationBaseConfig)
void *ref_pim_create(const double *acc0, const double *gyr0, const double *ba, const double *bg, const double *noise) {
  lio::IntegrationBaseConfig c;
  c.acc_n = noise[0]; c.gyr_n = noise[1]; c.acc_w = noise[2]; c.gyr_w = noise[3]; c.g_norm = noise[4];
  Pim *h = new Pim;
  h->p = std::make_shared<lio::IntegrationBase>(v3(acc0), v3(gyr0), v3(ba), v3(bg), c);
  return h;
}
void ref_pim_destroy(void *h) { delete static_cast<Pim *>(h); }
void ref_pim_push(void *h, double dt, const double *acc, const double *gyr) { static_cast<Pim *>(h)->p->push_back(dt, v3(acc), v3(gyr)); }
void ref_pim_repropagate(void *h, const double *ba, const double *bg) { static_cast<Pim *>(h)->p->Repropagate(v3(ba), v3(bg)); }
// dp[3], dq[4] = x y z w, dv[3], jacobian / covariance 15 x 15 row-major, sum_dt
void ref_pim_get(void *h, double *dp, double *dq, double *dv, double *jac, double *cov, double *sum_dt) {
  const lio::IntegrationBase &b = *static_cast<Pim *>(h)->p;
  for (int k = 0; k < 3; ++k) { dp[k] = b.delta_p_(k); dv[k] = b.delta_v_(k); }
  dq[0] = b.delta_q_.x(); dq[1] = b.delta_q_.y(); dq[2] = b.delta_q_.z(); dq[3] = b.delta_q_.w();
  for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) { jac[i * 15 + j] = b.jacobian_(i, j); cov[i * 15 + j] = b.covariance_(i, j); }
  *sum_dt = b.sum_dt_;
}
// IntegrationBase::Evaluate: the UNWHITENED residual; poses as [p, q_xyzw], speed-bias as [v, ba, bg]
void ref_pim_evaluate(void *h, const double *pose_i, const double *sb_i, const double *pose_j, const double *sb_j, double *res15) {
  Eigen::Matrix<double, 15, 1> r = static_cast<Pim *>(h)->p->Evaluate(v3(pose_i), qxyzw(pose_i + 3), v3(sb_i), v3(sb_i + 3), v3(sb_i + 6),
                                                                       v3(pose_j), qxyzw(pose_j + 3), v3(sb_j), v3(sb_j + 3), v3(sb_j + 6));
  for (int k = 0; k < 15; ++k) res15[k] = r(k);
}
// ImuFactor::Evaluate: whitened residual and the four ambient Jacobians (15x7, 15x9, 15x7, 15x9, row-major)
int ref_imu_factor(void *h, const double *pose_i, const double *sb_i, const double *pose_j, const double *sb_j, double *res15, double *J0, double *J1,
                   double *J2, double *J3) {
  lio::ImuFactor f(static_cast<Pim *>(h)->p);
  const double *params[4] = {pose_i, sb_i, pose_j, sb_j};
  double *jac[4] = {J0, J1, J2, J3};
  return f.Evaluate(params, res15, (J0 || J1 || J2 || J3) ? jac : nullptr) ? 1 : 0;
}
// PivotPointPlaneFactor::Evaluate: residual and the three 1x7 Jacobians
int ref_ppp_factor(const double *point, const double *coeff, const double *pose_p, const double *pose_i, const double *pose_ex, double *res, double *Jp,
                   double *Ji, double *Jex) {
  lio::PivotPointPlaneFactor f(v3(point), Eigen::Vector4d(coeff[0], coeff[1], coeff[2], coeff[3]));
  const double *params[3] = {pose_p, pose_i, pose_ex};
  double *jac[3] = {Jp, Ji, Jex};
  return f.Evaluate(params, res, (Jp || Ji || Jex) ? jac : nullptr) ? 1 : 0;
}
// PriorFactor::Evaluate: residual (6) and the 6x7 Jacobian (row-major); rot0 as x y z w
int ref_prior_factor(const double *pos0, const double *rot0, const double *pose, double *res6, double *J42) {
  lio::PriorFactor f(v3(pos0), Eigen::Quaterniond(rot0[3], rot0[0], rot0[1], rot0[2]));
  const double *params[1] = {pose};
  double *jac[1] = {J42};
  return f.Evaluate(params, res6, J42 ? jac : nullptr) ? 1 : 0;
}
// PoseLocalParameterization (its members are private in the reference: called through the Ceres interface, as Ceres does)
void ref_pose_plus(const double *x, const double *delta, double *out) {
  lio::PoseLocalParameterization p;
  static_cast<const ceres::LocalParameterization &>(p).Plus(x, delta, out);
}
void ref_pose_jacobian(const double *x, double *J42) {
  lio::PoseLocalParameterization p;
  static_cast<const ceres::LocalParameterization &>(p).ComputeJacobian(x, J42);
}

}  // extern "C"
