// C entry points over the PRODUCT's drop-in estimator class (lio-mapping_amd/dropin/EstimatorHip.{h,cc}) as estimator_node would run it:
// the reference's own MeasurementManager.cc (compiled where it lies: ImuHandler / CompactDataHandler / GetMeasurements) queues and pairs
// the messages, EstimatorHip::ProcessEstimation runs on its own thread (estimator_node.cc:153) and calls liblio_hip.so.
// TEST INFRASTRUCTURE: `make -C oracle ref` -> _ref/libdropin_estimator.so (links ../lio-mapping_amd/csrc/liblio_hip.so; ROS / PCL /
// Eigen come from the stand-in headers of oracle/ref_shim, like every other _ref library).  Driven by tests/test_gpu_dropin.py.
#include <chrono>
#include <cstring>
#include <thread>

#include "EstimatorHip.h"

namespace {
struct Handle {
  lio::EstimatorHip *est = nullptr;
  std::thread loop;
  ros::NodeHandle nh;
};
}  // namespace

extern "C" {

// ip / fp / dp as ref_est_create (oracle/ref_estimator.cc): the fields of the reference's EstimatorConfig; max_solver_time: the solver's
// wall-clock cap (0.10 s in the reference, Estimator.cc:1921; <= 0 lifts it, as the parity tests do on every side)
void *dropin_create(const int *ip, const float *fp, const double *dp, double msg_time_delay, double max_solver_time) {
  lio::EstimatorConfig c;
  c.window_size = size_t(ip[0]); c.opt_window_size = size_t(ip[1]); c.init_window_factor = ip[2]; c.estimate_extrinsic = ip[3];
  c.opt_extrinsic = ip[4]; c.imu_factor = ip[5]; c.point_distance_factor = ip[6]; c.prior_factor = ip[7]; c.marginalization_factor = ip[8];
  c.enable_deskew = ip[9]; c.cutoff_deskew = ip[10]; c.keep_features = ip[11];
  c.corner_filter_size = fp[0]; c.surf_filter_size = fp[1]; c.min_match_sq_dis = fp[2]; c.min_plane_dis = fp[3];
  c.transform_lb = lio::Transform(Eigen::Quaternionf(fp[7], fp[4], fp[5], fp[6]), Eigen::Vector3f(fp[8], fp[9], fp[10]));
  c.pim_config.acc_n = dp[0]; c.pim_config.gyr_n = dp[1]; c.pim_config.acc_w = dp[2]; c.pim_config.gyr_w = dp[3]; c.pim_config.g_norm = dp[4];
  lio::MeasurementManagerConfig mm;
  mm.msg_time_delay = msg_time_delay;
  Handle *h = new Handle;
  h->est = new lio::EstimatorHip(c, mm);                 // estimator_node.cc:142
  if (h->est->handle() && max_solver_time != 0.10) {
    h->est->max_solver_time_in_seconds_ = max_solver_time;
    h->est->ClearState();
  }
  if (!h->est->handle()) { delete h->est; delete h; return nullptr; }
  h->est->SetupRos(h->nh);                               // :143
  h->loop = std::thread(&lio::EstimatorHip::ProcessEstimation, h->est);   // :153
  return h;
}
void dropin_destroy(void *hv) {
  Handle *h = static_cast<Handle *>(hv);
  h->est->RequestStop();
  if (h->loop.joinable()) h->loop.join();
  delete h->est;
  delete h;
}
// the two subscriber callbacks of MeasurementManager::SetupRos (MeasurementManager.cc:35-52), called as the ROS spinner would
void dropin_push_imu(void *hv, double stamp, const double *acc, const double *gyr) {
  std::shared_ptr<sensor_msgs::Imu> m(new sensor_msgs::Imu());
  m->header.stamp = ros::Time(stamp);
  m->linear_acceleration.x = acc[0]; m->linear_acceleration.y = acc[1]; m->linear_acceleration.z = acc[2];
  m->angular_velocity.x = gyr[0]; m->angular_velocity.y = gyr[1]; m->angular_velocity.z = gyr[2];
  static_cast<Handle *>(hv)->est->ImuHandler(m);
}
void dropin_push_compact(void *hv, double stamp, const float *xyzi, size_t n) {
  std::shared_ptr<sensor_msgs::PointCloud2> m(new sensor_msgs::PointCloud2());
  m->header.stamp = ros::Time(stamp);
  m->xyzi.assign(xyzi, xyzi + 4 * n);
  static_cast<Handle *>(hv)->est->CompactDataHandler(m);
}
// blocks until thread B has finished `count` /compact_data messages; 0 = ok, -1 = timed out
int dropin_wait_processed(void *hv, size_t count, double timeout_s) {
  Handle *h = static_cast<Handle *>(hv);
  const auto t0 = std::chrono::steady_clock::now();
  while (h->est->processed_count() < count) {
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return -1;
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
  return 0;
}
size_t dropin_processed(void *hv) { return static_cast<Handle *>(hv)->est->processed_count(); }

// what the class mirrors under the reference's member names: stage_flag_, cir_buf_count_, extrinsic_stage_, the last event, the error
// code of the last library call; R_WI_, g_vec_
void dropin_get_stage(void *hv, int *out5, double *R_WI, double *g_vec) {
  lio::EstimatorHip &e = *static_cast<Handle *>(hv)->est;
  out5[0] = e.stage_flag_ == lio::INITED ? 1 : 0; out5[1] = int(e.cir_buf_count_); out5[2] = e.extrinsic_stage_; out5[3] = e.last_event_;
  out5[4] = e.last_error();
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R_WI[3 * r + c] = e.R_WI_(r, c);
  for (int d = 0; d < 3; ++d) g_vec[d] = e.g_vec_(d);
}
// Ps_ / Rs_ / Vs_ / Bas_ / Bgs_ by logical index of the CircularBuffers (returns how many entries they hold), transform_lb_,
// transform_aft_mapped_
int dropin_get_window(void *hv, double *Ps, double *Rs, double *Vs, double *Bas, double *Bgs, float *lb7, float *aft7) {
  lio::EstimatorHip &e = *static_cast<Handle *>(hv)->est;
  const int n = int(e.Ps_.size());
  for (int i = 0; i < n; ++i) {
    for (int k = 0; k < 3; ++k) { Ps[3 * i + k] = e.Ps_[i](k); Vs[3 * i + k] = e.Vs_[i](k); Bas[3 * i + k] = e.Bas_[i](k); Bgs[3 * i + k] = e.Bgs_[i](k); }
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rs[9 * i + 3 * r + c] = e.Rs_[i](r, c);
  }
  auto put = [](const lio::Transform &t, float *o) { o[0] = t.rot.x(); o[1] = t.rot.y(); o[2] = t.rot.z(); o[3] = t.rot.w(); o[4] = t.pos.x(); o[5] = t.pos.y(); o[6] = t.pos.z(); };
  put(e.transform_lb_, lb7);
  put(e.transform_aft_mapped_, aft7);
  return n;
}
// the last message on /predict_laser_odom and /local_laser_odom (stamp, seq, orientation xyzw, position) and the solve report's
// iterations / lidar residuals / final cost
void dropin_get_published(void *hv, double *laser9, double *local9, double *rep3) {
  lio::EstimatorHip &e = *static_cast<Handle *>(hv)->est;
  auto put = [](const nav_msgs::Odometry &m, double *o) {
    o[0] = m.header.stamp.toSec(); o[1] = double(m.header.seq);
    o[2] = m.pose.pose.orientation.x; o[3] = m.pose.pose.orientation.y; o[4] = m.pose.pose.orientation.z; o[5] = m.pose.pose.orientation.w;
    o[6] = m.pose.pose.position.x; o[7] = m.pose.pose.position.y; o[8] = m.pose.pose.position.z;
  };
  put(e.laser_odom_, laser9);
  put(e.local_odom_, local9);
  rep3[0] = e.last_report_.iterations; rep3[1] = e.last_report_.n_lidar_residuals; rep3[2] = e.last_report_.final_cost;
}

}  // extern "C"
