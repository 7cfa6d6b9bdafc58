// C entry points over the REFERENCE's own factor math, compiled from the sources where they lie under /root/reference
// (include/imu_processor/IntegrationBase.h, include/factor/ImuFactor.h, src/factor/PivotPointPlaneFactor.cc,
// src/factor/PriorFactor.cc, src/factor/PoseLocalParameterization.cc, include/utils/math_utils.h) against the stand-in headers of
// oracle/ref_shim (Eigen's dense API subset, Ceres' two base classes, the ROS/glog logging macros).  TEST INFRASTRUCTURE: built by
// `make -C oracle ref` into oracle/_ref/libref_factors.so when /root/reference exists; tests/golden/make_ref_factor_vectors.py
// turns its outputs into committed vectors, tests/test_oracle_ref_factors.py checks the oracle (and with it the product's host
// code, which is bit-identical to the oracle) against them.  Nothing of the reference is copied: this file only calls it.
#include <array>
#include <cstring>
#include <memory>
#include <unordered_map>
#include <vector>

#include "factor/ImuFactor.h"
#include "factor/MarginalizationFactor.h"
#include "factor/PivotPointPlaneFactor.h"
#include "factor/PoseLocalParameterization.h"
#include "factor/PriorFactor.h"
#include "imu_processor/ImuInitializer.h"
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


// ---- ImuInitializer (src/imu_processor/ImuInitializer.cc): transforms = n x (q xyzw, p) floats, pims = n handles of ref_pim_create
// (pims[0] may be null: the reference never reads the motion of frame 0), T_lb = q xyzw, p
static void fill_frames(int n, const float *transforms, void *const *pims, lio::CircularBuffer<lio::PairTimeLaserTransform> &all) {
  for (int i = 0; i < n; ++i) {
    const float *t = transforms + 7 * i;
    lio::LaserTransform lt(double(i), lio::Transform(Eigen::Quaternionf(t[3], t[0], t[1], t[2]), Eigen::Vector3f(t[4], t[5], t[6])));
    if (pims[i]) lt.pre_integration = static_cast<Pim *>(pims[i])->p;
    all.push(lio::PairTimeLaserTransform(double(i), lt));
  }
}
int ref_imu_estimate_extrinsic_rotation(int n, const float *transforms, void *const *pims, float *T_lb) {
  lio::CircularBuffer<lio::PairTimeLaserTransform> all(size_t(n) + 1);
  fill_frames(n, transforms, pims, all);
  lio::Transform lb(Eigen::Quaternionf(T_lb[3], T_lb[0], T_lb[1], T_lb[2]), Eigen::Vector3f(T_lb[4], T_lb[5], T_lb[6]));
  const bool ok = lio::ImuInitializer::EstimateExtrinsicRotation(all, lb);
  T_lb[0] = lb.rot.x(); T_lb[1] = lb.rot.y(); T_lb[2] = lb.rot.z(); T_lb[3] = lb.rot.w();
  return ok ? 1 : 0;
}
int ref_imu_initialization(int n, const float *transforms, void *const *pims, const float *T_lb, double *Vs_out, double *Bgs_inout, double *g_out,
                           double *R_WI_out) {
  lio::CircularBuffer<lio::PairTimeLaserTransform> all(size_t(n) + 1);
  fill_frames(n, transforms, pims, all);
  lio::CircularBuffer<Eigen::Vector3d> Vs(size_t(n) + 1), Bas(size_t(n) + 1), Bgs(size_t(n) + 1);
  for (int i = 0; i < n; ++i) { Vs.push(Eigen::Vector3d(0, 0, 0)); Bas.push(Eigen::Vector3d(0, 0, 0)); Bgs.push(v3(Bgs_inout + 3 * i)); }
  lio::Transform lb(Eigen::Quaternionf(T_lb[3], T_lb[0], T_lb[1], T_lb[2]), Eigen::Vector3f(T_lb[4], T_lb[5], T_lb[6]));
  Eigen::Vector3d g(0, 0, 0);
  Eigen::Matrix3d R_WI;
  R_WI.setIdentity();
  const bool ok = lio::ImuInitializer::Initialization(all, Vs, Bas, Bgs, g, lb, R_WI);
  for (int i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) { Vs_out[3 * i + k] = Vs[i](k); Bgs_inout[3 * i + k] = Bgs[i](k); }
  for (int k = 0; k < 3; ++k) g_out[k] = g(k);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R_WI_out[3 * i + j] = R_WI(i, j);
  return ok ? 1 : 0;
}

// ---- MarginalizationInfo driven the way Estimator::SolveOptimization drives it (Estimator.cc:2152-2245): the previous prior as a
// MarginalizationFactor dropping pose 0 / speed-bias 0, the ImuFactor of the first interval dropping both, one PivotPointPlaneFactor
// per lidar feature of the frames 1..Wo under CauchyLoss(1.0) dropping the pivot pose; PreMarginalize, Marginalize,
// GetParameterBlocks with the address shift i -> i - 1.
//   poses: (Wo + 1) x 7, sbs: (Wo + 1) x 9, ex: 7 (ambient layouts of the reference's para_* arrays)
//   prev (n_prev > 0): linearized_jacobians n_prev x n_prev row-major, linearized_residuals, and its kept blocks in ITS order:
//     prev_kind[k] = 0 pose / 1 speed-bias / 2 extrinsic, prev_index[k] = the window slot the block refers to NOW (after the shift),
//     prev_x0 = keep_block_data concatenated (ambient)
//   pim: handle of ref_pim_create (null: no IMU factor)
//   feat_n[i - 1], points (3 each), coeffs (4 each) of frame i = 1..Wo, concatenated
// out: returns n (< 0 on failure); m; lin_jac (n x n row-major), lin_res (n); the kept blocks in the reference's order:
//   kind / index (AFTER the shift, i.e. in the next window) / offset into the n residual columns / ambient size; x0 concatenated.
int ref_marginalize(int Wo, const double *poses, const double *sbs, const double *ex, int n_prev, const double *prev_jac, const double *prev_res,
                    int n_prev_blocks, const int *prev_kind, const int *prev_index, const double *prev_x0, void *pim, const int *feat_n,
                    const double *points, const double *coeffs, int *m_out, double *lin_jac, double *lin_res, int *n_blocks_out, int *kind_out,
                    int *index_out, int *offset_out, int *size_out, double *x0_out, int capacity_n) {
  std::vector<std::array<double, 7> > para_pose(Wo + 1);
  std::vector<std::array<double, 9> > para_sb(Wo + 1);
  std::array<double, 7> para_ex;
  for (int i = 0; i <= Wo; ++i) { std::memcpy(para_pose[i].data(), poses + 7 * i, 56); std::memcpy(para_sb[i].data(), sbs + 9 * i, 72); }
  std::memcpy(para_ex.data(), ex, 56);
  auto addr_of = [&](int kind, int index) -> double * { return kind == 0 ? para_pose[index].data() : (kind == 1 ? para_sb[index].data() : para_ex.data()); };

  lio::MarginalizationInfo *info = new lio::MarginalizationInfo();
  lio::MarginalizationInfo *last = nullptr;
  std::vector<double *> last_blocks;
  if (n_prev > 0) {
    last = new lio::MarginalizationInfo();
    last->n = n_prev; last->m = 0;
    last->linearized_jacobians = Eigen::MatrixXd(n_prev, n_prev);
    last->linearized_residuals = Eigen::VectorXd(n_prev);
    for (int i = 0; i < n_prev; ++i) { last->linearized_residuals(i) = prev_res[i]; for (int j = 0; j < n_prev; ++j) last->linearized_jacobians(i, j) = prev_jac[size_t(i) * n_prev + j]; }
    int off = 0, xo = 0;
    for (int k = 0; k < n_prev_blocks; ++k) {
      const int size = prev_kind[k] == 1 ? 9 : 7;
      last->keep_block_size.push_back(size);
      last->keep_block_idx.push_back(off);          // (m = 0: the offsets are the columns)
      double *d = new double[size];
      std::memcpy(d, prev_x0 + xo, sizeof(double) * size);
      last->keep_block_data.push_back(d);
      last->parameter_block_data[long(k) + 1] = d;  // owned by `last` (its destructor frees parameter_block_data)
      last_blocks.push_back(addr_of(prev_kind[k], prev_index[k]));
      off += size == 7 ? 6 : size; xo += size;
    }
    std::vector<int> drop_set;
    for (int i = 0; i < int(last_blocks.size()); ++i) if (last_blocks[i] == para_pose[0].data() || last_blocks[i] == para_sb[0].data()) drop_set.push_back(i);
    lio::MarginalizationFactor *mf = new lio::MarginalizationFactor(last);
    info->AddResidualBlockInfo(new lio::ResidualBlockInfo(mf, NULL, last_blocks, drop_set));
  }
  if (pim) {
    lio::ImuFactor *f = new lio::ImuFactor(static_cast<Pim *>(pim)->p);
    info->AddResidualBlockInfo(new lio::ResidualBlockInfo(f, NULL, std::vector<double *>{para_pose[0].data(), para_sb[0].data(), para_pose[1].data(), para_sb[1].data()},
                                                          std::vector<int>{0, 1}));
  }
  ceres::LossFunction *loss = new ceres::CauchyLoss(1.0);
  size_t at = 0;
  for (int i = 1; i <= Wo; ++i)
    for (int j = 0; j < feat_n[i - 1]; ++j, ++at) {
      lio::PivotPointPlaneFactor *f = new lio::PivotPointPlaneFactor(v3(points + 3 * at), Eigen::Vector4d(coeffs[4 * at], coeffs[4 * at + 1], coeffs[4 * at + 2], coeffs[4 * at + 3]));
      info->AddResidualBlockInfo(new lio::ResidualBlockInfo(f, loss, std::vector<double *>{para_pose[0].data(), para_pose[i].data(), para_ex.data()}, std::vector<int>{0}));
    }
  info->PreMarginalize();
  info->Marginalize();
  std::unordered_map<long, double *> addr_shift;
  for (int i = 1; i <= Wo; ++i) { addr_shift[reinterpret_cast<long>(para_pose[i].data())] = para_pose[i - 1].data(); addr_shift[reinterpret_cast<long>(para_sb[i].data())] = para_sb[i - 1].data(); }
  addr_shift[reinterpret_cast<long>(para_ex.data())] = para_ex.data();
  std::vector<double *> blocks = info->GetParameterBlocks(addr_shift);
  const int n = info->n;
  int rc = n;
  if (n > capacity_n) rc = -1;
  else {
    *m_out = info->m;
    for (int i = 0; i < n; ++i) { lin_res[i] = info->linearized_residuals(i); for (int j = 0; j < n; ++j) lin_jac[size_t(i) * n + j] = info->linearized_jacobians(i, j); }
    *n_blocks_out = int(blocks.size());
    int xo = 0;
    for (size_t k = 0; k < blocks.size(); ++k) {
      int kind = 2, index = 0;
      for (int i = 0; i <= Wo; ++i) { if (blocks[k] == para_pose[i].data()) { kind = 0; index = i; } if (blocks[k] == para_sb[i].data()) { kind = 1; index = i; } }
      kind_out[k] = kind; index_out[k] = index; offset_out[k] = info->keep_block_idx[k] - info->m; size_out[k] = info->keep_block_size[k];
      std::memcpy(x0_out + xo, info->keep_block_data[k], sizeof(double) * info->keep_block_size[k]);
      xo += info->keep_block_size[k];
    }
  }
  delete info;      // frees its factors (and with them the MarginalizationFactor object, not `last`)
  if (last) { last->keep_block_data.clear(); delete last; }
  delete loss;
  return rc;
}

}  // extern "C"
