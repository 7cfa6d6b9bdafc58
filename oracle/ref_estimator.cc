// C entry points over the REFERENCE's own Estimator (src/imu_processor/Estimator.cc, with MeasurementManager.cc, PointMapping.cc,
// ImuInitializer.cc, FeatureManager.cc and the factor sources), compiled from the sources where they lie against the stand-in headers
// of oracle/ref_shim.  TEST INFRASTRUCTURE (`make -C oracle ref` -> _ref/libref_estimator.so).
// What runs is the reference's ProcessImu (mid-point propagation + pre-integration), ProcessCompactData, ProcessLaserOdom (the stage
// machine, window filling, extrinsic rotation + RunInitialization, the de-skew branch), BuildLocalMap / CalculateFeatures /
// CalculateLaserOdom, SolveOptimization (problem assembly, the turn-off / convergence logic, marginalization), VectorToDouble /
// DoubleToVector and SlideWindow.  Stood in: Ceres' Problem / Solve (oracle/ref_shim/ceres/problem.h — the minimizer is the oracle's
// restatement of Ceres 1.14, so its step sequence is NOT independently pinned; every residual and Jacobian comes from the
// reference's factor classes), pcl::VoxelGrid / KdTreeFLANN / ExtractIndices, Eigen's dense API and decompositions, Sophus::SO3,
// the visualizers and the ROS plumbing.
#include <cstring>

#define private public
#define protected public
#include "imu_processor/Estimator.h"
#undef private
#undef protected

namespace {
struct Handle {
  lio::Estimator *est = nullptr;
  int last_event = 0;   // 0 skipped, 1 filling, 2 init_failed, 3 initialised, 4 solved
};
Eigen::Vector3d v3(const double *p) { return Eigen::Vector3d(p[0], p[1], p[2]); }
void put(const lio::Transform &t, float *out) {
  out[0] = t.rot.x(); out[1] = t.rot.y(); out[2] = t.rot.z(); out[3] = t.rot.w();
  out[4] = t.pos.x(); out[5] = t.pos.y(); out[6] = t.pos.z();
}
}  // namespace

extern "C" {

// ip: window_size, opt_window_size, init_window_factor, estimate_extrinsic, opt_extrinsic, imu_factor, point_distance_factor,
//     prior_factor, marginalization_factor, enable_deskew, cutoff_deskew, keep_features
// fp: corner_filter_size, surf_filter_size, min_match_sq_dis, min_plane_dis, q_lb (x y z w), p_lb
// dp: acc_n, gyr_n, acc_w, gyr_w, g_norm
void *ref_est_create(const int *ip, const float *fp, const double *dp) {
  lio::EstimatorConfig c;
  c.window_size = size_t(ip[0]); c.opt_window_size = size_t(ip[1]); c.init_window_factor = ip[2]; c.estimate_extrinsic = ip[3];
  c.opt_extrinsic = ip[4]; c.imu_factor = ip[5]; c.point_distance_factor = ip[6]; c.prior_factor = ip[7]; c.marginalization_factor = ip[8];
  c.enable_deskew = ip[9]; c.cutoff_deskew = ip[10]; c.keep_features = ip[11];
  c.corner_filter_size = fp[0]; c.surf_filter_size = fp[1]; c.min_match_sq_dis = fp[2]; c.min_plane_dis = fp[3];
  c.transform_lb = lio::Transform(Eigen::Quaternionf(fp[7], fp[4], fp[5], fp[6]), Eigen::Vector3f(fp[8], fp[9], fp[10]));
  c.pim_config.acc_n = dp[0]; c.pim_config.gyr_n = dp[1]; c.pim_config.acc_w = dp[2]; c.pim_config.gyr_w = dp[3]; c.pim_config.g_norm = dp[4];
  Handle *h = new Handle;
  h->est = new lio::Estimator(c, lio::MeasurementManagerConfig());
  return h;
}
void ref_est_destroy(void *hv) { Handle *h = static_cast<Handle *>(hv); delete h->est; delete h; }

void ref_est_process_imu(void *hv, double dt, const double *acc, const double *gyr, double stamp) {
  std_msgs::Header hd;
  hd.stamp = ros::Time(stamp);
  static_cast<Handle *>(hv)->est->ProcessImu(dt, v3(acc), v3(gyr), hd);
}

// One /compact_data message through Estimator::ProcessCompactData.  Returns the event (see Handle); T_out7 = transform_aft_mapped_
// (what ProcessLaserOdom was handed); rep = iterations, successful steps, termination, lidar residual blocks, blocks in all,
// initial cost, final cost, number of Problem::Evaluate calls, then their costs (up to 8), then the cost trace (up to 32).
int ref_est_process_compact(void *hv, const float *xyzi, size_t n, double stamp, float *T_out7, double *rep) {
  Handle *h = static_cast<Handle *>(hv);
  lio::Estimator &e = *h->est;
  std::shared_ptr<sensor_msgs::PointCloud2> m(new sensor_msgs::PointCloud2());
  m->xyzi.assign(xyzi, xyzi + 4 * n);
  m->header.stamp = ros::Time(stamp);
  const bool was_inited = e.stage_flag_ == lio::INITED;
  const size_t count_before = e.cir_buf_count_;
  ceres::evaluate_log().clear();
  ceres::last_blocks().clear();
  ceres::last_summary() = ceres::Solver::Summary();
  e.ProcessCompactData(m, m->header);
  const bool inited = e.stage_flag_ == lio::INITED;
  int ev;
  if (was_inited) ev = 4;
  else if (inited) ev = 3;
  else if (e.laser_odom_recv_count_ % size_t(e.estimator_config_.init_window_factor) != 0) ev = 0;
  else if (count_before < e.estimator_config_.window_size) ev = 1;
  else ev = 2;
  h->last_event = ev;
  if (T_out7) put(e.transform_aft_mapped_, T_out7);
  if (rep) {
    const ceres::Solver::Summary &s = ceres::last_summary();
    int n_lidar = 0;
    for (const ceres::internal::ResidualBlock &b : ceres::last_blocks()) if (dynamic_cast<lio::PivotPointPlaneFactor *>(b.cost)) ++n_lidar;
    rep[0] = s.iterations; rep[1] = s.successful; rep[2] = s.termination; rep[3] = n_lidar; rep[4] = double(ceres::last_blocks().size());
    rep[5] = s.initial_cost; rep[6] = s.final_cost;
    const std::vector<double> &ev_log = ceres::evaluate_log();
    rep[7] = double(ev_log.size());
    for (size_t k = 0; k < 8; ++k) rep[8 + k] = k < ev_log.size() ? ev_log[k] : 0.0;
    for (size_t k = 0; k < 32; ++k) rep[16 + k] = k < s.cost_trace.size() ? s.cost_trace[k] : 0.0;
  }
  return ev;
}

void ref_est_get_stage(void *hv, int *stage, int *cir_buf_count, int *extrinsic_stage, int *convergence, double *R_WI, double *g_vec) {
  lio::Estimator &e = *static_cast<Handle *>(hv)->est;
  *stage = e.stage_flag_ == lio::INITED ? 1 : 0;
  *cir_buf_count = int(e.cir_buf_count_);
  *extrinsic_stage = e.extrinsic_stage_;
  *convergence = e.convergence_flag_ ? 1 : 0;
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R_WI[r * 3 + c] = e.R_WI_(r, c);
  for (int d = 0; d < 3; ++d) g_vec[d] = e.g_vec_(d);
}

void ref_est_get_window(void *hv, double *Ps, double *Rs, double *Vs, double *Bas, double *Bgs, float *lb7) {
  lio::Estimator &e = *static_cast<Handle *>(hv)->est;
  const int n = int(e.estimator_config_.window_size) + 1;
  for (int i = 0; i < n && i < int(e.Ps_.size()); ++i) {
    for (int k = 0; k < 3; ++k) { Ps[3 * i + k] = e.Ps_[i](k); Vs[3 * i + k] = e.Vs_[i](k); Bas[3 * i + k] = e.Bas_[i](k); Bgs[3 * i + k] = e.Bgs_[i](k); }
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rs[9 * i + 3 * r + c] = e.Rs_[i](r, c);
  }
  put(e.transform_lb_, lb7);
}

// The PivotPointPlaneFactor blocks of the last solve that belong to opt-window frame `frame` (1..Wo), in the order they were added.
size_t ref_est_get_features(void *hv, int frame, double *pt, double *co) {
  lio::Estimator &e = *static_cast<Handle *>(hv)->est;
  size_t k = 0;
  for (const ceres::internal::ResidualBlock &b : ceres::last_blocks()) {
    lio::PivotPointPlaneFactor *f = dynamic_cast<lio::PivotPointPlaneFactor *>(b.cost);
    if (!f || b.params.size() != 3 || b.params[1] != e.para_pose_[frame]) continue;
    if (pt) for (int j = 0; j < 3; ++j) pt[3 * k + j] = f->point_(j);
    if (co) for (int j = 0; j < 4; ++j) co[4 * k + j] = f->coeff_(j);
    ++k;
  }
  return k;
}

size_t ref_est_get_local_map(void *hv, float *out) {
  lio::Estimator &e = *static_cast<Handle *>(hv)->est;
  if (!e.local_surf_points_filtered_ptr_) return 0;
  const lio::PointCloud &c = *e.local_surf_points_filtered_ptr_;
  if (out) for (size_t i = 0; i < c.size(); ++i) { out[4 * i] = c[i].x; out[4 * i + 1] = c[i].y; out[4 * i + 2] = c[i].z; out[4 * i + 3] = c[i].intensity; }
  return c.size();
}
size_t ref_est_get_surf_stack(void *hv, int frame, float *out) {
  lio::Estimator &e = *static_cast<Handle *>(hv)->est;
  if (frame < 0 || size_t(frame) >= e.surf_stack_.size() || !e.surf_stack_[frame]) return 0;
  const lio::PointCloud &c = *e.surf_stack_[frame];
  if (out) for (size_t i = 0; i < c.size(); ++i) { out[4 * i] = c[i].x; out[4 * i + 1] = c[i].y; out[4 * i + 2] = c[i].z; out[4 * i + 3] = c[i].intensity; }
  return c.size();
}

// The marginalization prior the last solve left behind: returns n (0: none); the kept blocks in the reference's (hash-map) order as
// kind (0 pose, 1 speed-bias, 2 extrinsic) / opt-window index in the NEXT window / column offset / ambient size; x0 concatenated.
int ref_est_get_prior(void *hv, double *lin_jac, double *lin_res, int *n_blocks, int *kind, int *index, int *offset, int *size, double *x0, int capacity_n) {
  lio::Estimator &e = *static_cast<Handle *>(hv)->est;
  lio::MarginalizationInfo *info = e.last_marginalization_info;
  if (!info) return 0;
  const int n = info->n;
  if (n > capacity_n) return -1;
  for (int i = 0; i < n; ++i) { lin_res[i] = info->linearized_residuals(i); for (int j = 0; j < n; ++j) lin_jac[size_t(i) * n + j] = info->linearized_jacobians(i, j); }
  const std::vector<double *> &blocks = e.last_marginalization_parameter_blocks;
  const int Wo = int(e.estimator_config_.opt_window_size);
  int xo = 0;
  *n_blocks = int(blocks.size());
  for (size_t k = 0; k < blocks.size(); ++k) {
    int kd = -1, ix = -1;
    for (int i = 0; i <= Wo; ++i) {
      if (blocks[k] == e.para_pose_[i]) { kd = 0; ix = i; }
      if (blocks[k] == e.para_speed_bias_[i]) { kd = 1; ix = i; }
    }
    if (blocks[k] == e.para_ex_pose_) { kd = 2; ix = 0; }
    kind[k] = kd; index[k] = ix; size[k] = info->keep_block_size[k]; offset[k] = info->keep_block_idx[k] - info->m;
    for (int j = 0; j < info->keep_block_size[k]; ++j) x0[xo++] = info->keep_block_data[k][j];
  }
  return n;
}

// ---- the problem of the last solve, as the reference handed it to ceres::Solve (for tests/host/ref_solve_check.hip: the PRODUCT's
// host solver on the same problem)
// params: for opt frame i = 0..Wo pose (7) then speed-bias (9), then the extrinsic (7): values at the start and at the end of the
// solve; flags[0] = extrinsic constant, flags[1] = a MarginalizationFactor block is present, flags[2] = a PriorFactor block is present
void ref_est_get_solve_params(void *hv, double *initial, double *final_, int *flags, double *prior_pos_rot7) {
  lio::Estimator &e = *static_cast<Handle *>(hv)->est;
  const int Wo = int(e.estimator_config_.opt_window_size);
  auto find = [&](double *p) -> const ceres::ParamRecord * { for (const ceres::ParamRecord &r : ceres::last_params()) if (r.ptr == p) return &r; return nullptr; };
  int o = 0;
  auto put_block = [&](double *p) { const ceres::ParamRecord *r = find(p); for (size_t k = 0; k < r->initial.size(); ++k) { initial[o] = r->initial[k]; final_[o] = r->final_[k]; ++o; } };
  for (int i = 0; i <= Wo; ++i) { put_block(e.para_pose_[i]); put_block(e.para_speed_bias_[i]); }
  put_block(e.para_ex_pose_);
  flags[0] = find(e.para_ex_pose_)->constant ? 1 : 0; flags[1] = 0; flags[2] = 0;
  for (const ceres::internal::ResidualBlock &b : ceres::last_blocks()) {
    if (dynamic_cast<lio::MarginalizationFactor *>(b.cost)) flags[1] = 1;
    if (lio::PriorFactor *f = dynamic_cast<lio::PriorFactor *>(b.cost)) {
      flags[2] = 1;
      for (int k = 0; k < 3; ++k) prior_pos_rot7[k] = f->pos_(k);
      prior_pos_rot7[3] = f->rot_.x(); prior_pos_rot7[4] = f->rot_.y(); prior_pos_rot7[5] = f->rot_.z(); prior_pos_rot7[6] = f->rot_.w();
    }
  }
}
// the parameter arrays as they stand (after ProcessCompactData: what VectorToDouble wrote for the marginalization, Estimator.cc:2152)
void ref_est_get_para(void *hv, double *out) {
  lio::Estimator &e = *static_cast<Handle *>(hv)->est;
  const int Wo = int(e.estimator_config_.opt_window_size);
  int o = 0;
  for (int i = 0; i <= Wo; ++i) { for (int k = 0; k < 7; ++k) out[o++] = e.para_pose_[i][k]; for (int k = 0; k < 9; ++k) out[o++] = e.para_speed_bias_[i][k]; }
  for (int k = 0; k < 7; ++k) out[o++] = e.para_ex_pose_[k];
}
// the ImuFactor between opt frames i and i + 1 of the last solve: returns the number of samples (-1: no such factor); head = acc0 (3),
// gyr0 (3), ba (3), bg (3) the integration was (re)started from; samples = dt, acc (3), gyr (3) each
int ref_est_get_imu_factor(void *hv, int i, double *head12, double *samples, int capacity) {
  lio::Estimator &e = *static_cast<Handle *>(hv)->est;
  for (const ceres::internal::ResidualBlock &b : ceres::last_blocks()) {
    lio::ImuFactor *f = dynamic_cast<lio::ImuFactor *>(b.cost);
    if (!f || b.params[0] != e.para_pose_[i]) continue;
    const lio::IntegrationBase &p = *f->pre_integration_;
    const int n = int(p.dt_buf_.size());
    if (n > capacity) return -2;
    for (int k = 0; k < 3; ++k) { head12[k] = p.linearized_acc_(k); head12[3 + k] = p.linearized_gyr_(k); head12[6 + k] = p.linearized_ba_(k); head12[9 + k] = p.linearized_bg_(k); }
    for (int s = 0; s < n; ++s) {
      samples[7 * s] = p.dt_buf_[s];
      for (int k = 0; k < 3; ++k) { samples[7 * s + 1 + k] = p.acc_buf_[s](k); samples[7 * s + 4 + k] = p.gyr_buf_[s](k); }
    }
    return n;
  }
  return -1;
}

// ---- the estimator's buffers as they stand (after ProcessCompactData: post-slide), by logical index of the reference's CircularBuffers,
// for tests that inject this state into another implementation and take ONE step from it
// the pre-integration in window slot `slot` (0..W): returns the number of samples (-1: none); head / samples as ref_est_get_imu_factor
int ref_est_get_preintegration(void *hv, int slot, double *head12, double *samples, int capacity) {
  lio::Estimator &e = *static_cast<Handle *>(hv)->est;
  if (slot < 0 || size_t(slot) >= e.pre_integrations_.size() || !e.pre_integrations_[slot]) return -1;
  const lio::IntegrationBase &p = *e.pre_integrations_[slot];
  const int n = int(p.dt_buf_.size());
  if (n > capacity) return -2;
  for (int k = 0; k < 3; ++k) { head12[k] = p.linearized_acc_(k); head12[3 + k] = p.linearized_gyr_(k); head12[6 + k] = p.linearized_ba_(k); head12[9 + k] = p.linearized_bg_(k); }
  for (int s = 0; s < n; ++s) {
    samples[7 * s] = p.dt_buf_[s];
    for (int k = 0; k < 3; ++k) { samples[7 * s + 1 + k] = p.acc_buf_[s](k); samples[7 * s + 4 + k] = p.gyr_buf_[s](k); }
  }
  return n;
}
// the pre-integration in flight (tmp_pre_integration_): acc, gyr, ba, bg it was started from
void ref_est_get_tmp_preintegration(void *hv, double *head12) {
  lio::Estimator &e = *static_cast<Handle *>(hv)->est;
  const lio::IntegrationBase &p = *e.tmp_pre_integration_;
  for (int k = 0; k < 3; ++k) { head12[k] = p.linearized_acc_(k); head12[3 + k] = p.linearized_gyr_(k); head12[6 + k] = p.linearized_ba_(k); head12[9 + k] = p.linearized_bg_(k); }
}

// ---- MeasurementManager::GetMeasurements alone (MeasurementManager.cc:54-108): messages in, pairings out
void *ref_mm_create(double msg_time_delay) {
  lio::MeasurementManager *m = new lio::MeasurementManager();
  m->mm_config_.msg_time_delay = msg_time_delay;
  return m;
}
void ref_mm_destroy(void *h) { delete static_cast<lio::MeasurementManager *>(h); }
void ref_mm_push_imu(void *h, double stamp) {
  std::shared_ptr<sensor_msgs::Imu> m(new sensor_msgs::Imu());
  m->header.stamp = ros::Time(stamp);
  static_cast<lio::MeasurementManager *>(h)->ImuHandler(m);
}
void ref_mm_push_compact(void *h, double stamp) {
  std::shared_ptr<sensor_msgs::PointCloud2> m(new sensor_msgs::PointCloud2());
  m->header.stamp = ros::Time(stamp);
  static_cast<lio::MeasurementManager *>(h)->CompactDataHandler(m);
}
// every pairing that is ready: out[4k..] = compact stamp, number of IMU messages, first and last IMU stamp; returns the count
int ref_mm_get_measurements(void *h, double *out, int capacity) {
  lio::PairMeasurements ms = static_cast<lio::MeasurementManager *>(h)->GetMeasurements();
  int k = 0;
  for (const lio::PairMeasurement &m : ms) {
    if (k >= capacity) break;
    out[4 * k] = m.second->header.stamp.toSec();
    out[4 * k + 1] = double(m.first.size());
    out[4 * k + 2] = m.first.empty() ? 0.0 : m.first.front()->header.stamp.toSec();
    out[4 * k + 3] = m.first.empty() ? 0.0 : m.first.back()->header.stamp.toSec();
    ++k;
  }
  return k;
}

}  // extern "C"
