// EstimatorHip.cc — see EstimatorHip.h.  Host C++ of the product: the reference's estimator surface over the C-ABI of liblio_hip.so.
#include "EstimatorHip.h"

#include <cstdio>

#include <pcl_conversions/pcl_conversions.h>

namespace lio {

namespace {
void ToPod(const Transform &t, lio_transform_f *out) {
  out->q[0] = t.rot.x(); out->q[1] = t.rot.y(); out->q[2] = t.rot.z(); out->q[3] = t.rot.w();
  out->p[0] = t.pos.x(); out->p[1] = t.pos.y(); out->p[2] = t.pos.z();
}
Transform FromPod(const lio_transform_f &t) {
  return Transform(Eigen::Quaternionf(t.q[3], t.q[0], t.q[1], t.q[2]), Eigen::Vector3f(t.p[0], t.p[1], t.p[2]));
}
void PutPose(const Eigen::Quaterniond &q, const Eigen::Vector3d &p, nav_msgs::Odometry &m) {
  m.pose.pose.orientation.x = q.x(); m.pose.pose.orientation.y = q.y(); m.pose.pose.orientation.z = q.z(); m.pose.pose.orientation.w = q.w();
  m.pose.pose.position.x = p.x(); m.pose.pose.position.y = p.y(); m.pose.pose.position.z = p.z();
}
}  // namespace

EstimatorHip::EstimatorHip() : EstimatorHip(EstimatorConfig()) {}

EstimatorHip::EstimatorHip(EstimatorConfig config, MeasurementManagerConfig mm_config) {
  R_WI_.setIdentity();
  g_vec_ = Vector3d(0, 0, -config.pim_config.g_norm);
  laser_cloud_surf_last_.reset(new PointCloud());
  laser_cloud_corner_last_.reset(new PointCloud());
  SetupAllEstimatorConfig(config, mm_config);
  ClearState();
}

EstimatorHip::~EstimatorHip() {
  if (est_) lio_est_destroy(est_);
}

bool EstimatorHip::Check(int rc, const char *what) {
  last_error_ = rc;
  if (rc == LIO_OK) return true;
  // the reference's methods are void and log their failures (Estimator.cc:558,1650-1653); so does this class
  LOG(ERROR) << what << " failed with code " << rc;
  std::fprintf(stderr, "EstimatorHip: %s failed with code %d\n", what, rc);
  return false;
}

void EstimatorHip::SetupAllEstimatorConfig(const EstimatorConfig &config, const MeasurementManagerConfig &mm_config) {
  this->mm_config_ = mm_config;
  if (!config.imu_factor) this->mm_config_.enable_imu = false;   // Estimator.cc:196-198
  estimator_config_ = config;
  transform_lb_ = config.transform_lb;
  extrinsic_stage_ = config.estimate_extrinsic;
  const size_t n = config.window_size + 1;
  Ps_.Reset(n); Rs_.Reset(n); Vs_.Reset(n); Bas_.Reset(n); Bgs_.Reset(n);
  stamps_.clear();
}

void EstimatorHip::CreateHandle() {
  const EstimatorConfig &c = estimator_config_;
  lio_est_config k;
  lio_est_default_config(&k);
  k.window_size = int(c.window_size); k.opt_window_size = int(c.opt_window_size);          // Estimator.h:78-79
  k.init_window_factor = c.init_window_factor; k.extrinsic_stage = c.estimate_extrinsic;     // :80-81
  k.corner_filter_size = c.corner_filter_size; k.surf_filter_size = c.surf_filter_size;      // :83-84
  k.min_match_sq_dis = c.min_match_sq_dis; k.min_plane_dis = c.min_plane_dis;                // :87-88
  ToPod(c.transform_lb, &k.transform_lb);                                                    // :89
  k.opt_extrinsic = c.opt_extrinsic; k.imu_factor = c.imu_factor; k.point_distance_factor = c.point_distance_factor;
  k.prior_factor = c.prior_factor; k.marginalization_factor = c.marginalization_factor;
  k.enable_deskew = c.enable_deskew; k.cutoff_deskew = c.cutoff_deskew; k.keep_features = c.keep_features;
  k.acc_n = c.pim_config.acc_n; k.gyr_n = c.pim_config.gyr_n; k.acc_w = c.pim_config.acc_w; k.gyr_w = c.pim_config.gyr_w;
  k.g_norm = c.pim_config.g_norm;
  k.max_num_iterations = max_num_iterations_; k.max_solver_time = max_solver_time_in_seconds_;   // Estimator.cc:1916,1921
  if (est_) lio_est_destroy(est_);
  est_ = lio_est_create(&k);
  if (!est_) {
    // no GPU, or a configuration the library refuses: fatal, like the reference's LOG(FATAL) at Estimator.cc:558
    LOG(FATAL) << "lio_est_create failed (no MI355X visible, or bad EstimatorConfig)";
    std::fprintf(stderr, "EstimatorHip: lio_est_create failed (no GPU visible, or bad EstimatorConfig)\n");
    last_error_ = LIO_ERR_DEVICE;
  }
}

void EstimatorHip::ClearState() {
  CreateHandle();
  stage_flag_ = NOT_INITED;
  cir_buf_count_ = 0;
  convergence_flag_ = false;
  R_WI_.setIdentity();
  last_event_ = 0;
  stamps_.clear();
  const size_t n = estimator_config_.window_size + 1;
  Ps_.Reset(n); Rs_.Reset(n); Vs_.Reset(n); Bas_.Reset(n); Bgs_.Reset(n);
  last_report_ = lio_solve_report();
}

void EstimatorHip::SetupRos(ros::NodeHandle &nh) {
  MeasurementManager::SetupRos(nh);   // /imu/data and /compact_data subscriptions: the reference's own
  predict_odom_.header.frame_id = "/world"; predict_odom_.child_frame_id = "/imu_predict";
  pub_predict_odom_ = nh.advertise<nav_msgs::Odometry>("/predict_odom", 100);
  laser_odom_.header.frame_id = "/world"; laser_odom_.child_frame_id = "/laser_predict";
  pub_laser_odom_ = nh.advertise<nav_msgs::Odometry>("/predict_laser_odom", 100);
  local_odom_.header.frame_id = "/world"; local_odom_.child_frame_id = "/laser_predict";
  pub_local_odom_ = nh.advertise<nav_msgs::Odometry>("/local_laser_odom", 100);
  pub_extrinsic_ = nh.advertise<geometry_msgs::PoseStamped>("/extrinsic_lb", 10);
}

void EstimatorHip::Refresh() {
  if (!est_) return;
  int stage = 0, count = 0, ex_stage = 0, event = 0;
  double R[9], g[3];
  if (!Check(lio_est_get_stage(est_, &stage, &count, &ex_stage, &event, R, g), "lio_est_get_stage")) return;
  stage_flag_ = stage ? INITED : NOT_INITED;
  cir_buf_count_ = size_t(count);
  extrinsic_stage_ = ex_stage;
  last_event_ = event;
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R_WI_(r, c) = R[3 * r + c];
  g_vec_ = Vector3d(g[0], g[1], g[2]);
  const int n = int(estimator_config_.window_size) + 1;
  std::vector<double> P(3 * n), Rm(9 * n), V(3 * n), Ba(3 * n), Bg(3 * n);
  lio_transform_f lb;
  if (!Check(lio_est_get_window(est_, n, P.data(), Rm.data(), V.data(), Ba.data(), Bg.data(), &lb), "lio_est_get_window")) return;
  transform_lb_ = FromPod(lb);
  // the reference's CircularBuffers hold cir_buf_count_ + 1 entries while the window fills and W + 1 afterwards; logical index i
  // of the library's window is logical index i of those buffers
  const int held = stage ? n : std::min(n, count + 1);
  Ps_.Reset(n); Rs_.Reset(n); Vs_.Reset(n); Bas_.Reset(n); Bgs_.Reset(n);
  for (int i = 0; i < held; ++i) {
    Ps_.push(Vector3d(P[3 * i], P[3 * i + 1], P[3 * i + 2]));
    Vs_.push(Vector3d(V[3 * i], V[3 * i + 1], V[3 * i + 2]));
    Bas_.push(Vector3d(Ba[3 * i], Ba[3 * i + 1], Ba[3 * i + 2]));
    Bgs_.push(Vector3d(Bg[3 * i], Bg[3 * i + 1], Bg[3 * i + 2]));
    Matrix3d Ri;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Ri(r, c) = Rm[9 * i + 3 * r + c];
    Rs_.push(Ri);
  }
}

// Estimator::ProcessImu (Estimator.cc:338-427): propagation and pre-integration run in the library; /predict_odom is published from
// the newest state as at :395-424
void EstimatorHip::ProcessImu(double dt, const Vector3d &linear_acceleration, const Vector3d &angular_velocity, const std_msgs::Header &header) {
  if (!est_) return;
  const double a[3] = {linear_acceleration.x(), linear_acceleration.y(), linear_acceleration.z()};
  const double w[3] = {angular_velocity.x(), angular_velocity.y(), angular_velocity.z()};
  if (!Check(lio_est_process_imu(est_, dt, a, w, header.stamp.toSec()), "lio_est_process_imu")) return;
  if (stage_flag_ != INITED) { Refresh(); return; }   // the reference propagates Ps_ / Rs_ / Vs_[cir_buf_count_] per sample (:387-394): keep the mirrors current
  {
    const int n = int(estimator_config_.window_size) + 1;
    std::vector<double> P(3 * n), Rm(9 * n), V(3 * n), Ba(3 * n);
    if (!Check(lio_est_get_window(est_, n, P.data(), Rm.data(), V.data(), Ba.data(), nullptr, nullptr), "lio_est_get_window")) return;
    Matrix3d Rl;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rl(r, c) = Rm[9 * (n - 1) + 3 * r + c];
    Ps_.last() = Vector3d(P[3 * n - 3], P[3 * n - 2], P[3 * n - 1]);
    Vs_.last() = Vector3d(V[3 * n - 3], V[3 * n - 2], V[3 * n - 1]);
    Rs_.last() = Rl;
    predict_odom_.header.stamp = header.stamp;
    predict_odom_.header.seq += 1;
    PutPose(Eigen::Quaterniond(Rl), Ps_.last(), predict_odom_);
    predict_odom_.twist.twist.linear.x = Vs_.last().x(); predict_odom_.twist.twist.linear.y = Vs_.last().y(); predict_odom_.twist.twist.linear.z = Vs_.last().z();
    predict_odom_.twist.twist.angular.x = Ba[3 * n - 3]; predict_odom_.twist.twist.angular.y = Ba[3 * n - 2]; predict_odom_.twist.twist.angular.z = Ba[3 * n - 1];
    pub_predict_odom_.publish(predict_odom_);
  }
}

void EstimatorHip::Pack(const PointCloud &cloud, std::vector<float> &xyzi) {
  xyzi.resize(4 * cloud.size());   // pcl::PointXYZI is 32 B with padding; the library takes x, y, z, intensity
  for (size_t i = 0; i < cloud.size(); ++i) {
    xyzi[4 * i] = cloud[i].x; xyzi[4 * i + 1] = cloud[i].y; xyzi[4 * i + 2] = cloud[i].z; xyzi[4 * i + 3] = cloud[i].intensity;
  }
}

// /extrinsic_lb (published by SolveOptimization itself, Estimator.cc:2343-2353: on every solve, the initialising one included) and —
// in the INITED branch only (:728-758; the initialisation branch :541-600 runs SolveOptimization + SlideWindow and nothing else) —
// /local_laser_odom and /predict_laser_odom from the window after the slide
void EstimatorHip::PublishAfterSolve(const std_msgs::Header &header, bool odometry_topics) {
  const int pivot_idx = int(estimator_config_.window_size) - int(estimator_config_.opt_window_size);
  const Twist<double> transform_lb = transform_lb_.cast<double>();
  if (odometry_topics) {
  if (size_t(pivot_idx + 1) < stamps_.size()) local_odom_.header.stamp = ros::Time(stamps_[pivot_idx + 1]);
  local_odom_.header.seq += 1;
  {
    const Eigen::Quaterniond rot(Rs_[pivot_idx] * transform_lb.rot.inverse());
    PutPose(rot, Ps_[pivot_idx] - rot * transform_lb.pos, local_odom_);
    pub_local_odom_.publish(local_odom_);
  }
  laser_odom_.header.stamp = header.stamp;
  laser_odom_.header.seq += 1;
  {
    const Eigen::Quaterniond rot(Rs_.last() * transform_lb.rot.inverse());
    PutPose(rot, Ps_.last() - rot * transform_lb.pos, laser_odom_);
    pub_laser_odom_.publish(laser_odom_);
  }
  }
  geometry_msgs::PoseStamped ex;
  ex.header = header;
  ex.pose.position.x = transform_lb.pos.x(); ex.pose.position.y = transform_lb.pos.y(); ex.pose.position.z = transform_lb.pos.z();
  ex.pose.orientation.w = transform_lb.rot.w(); ex.pose.orientation.x = transform_lb.rot.x();
  ex.pose.orientation.y = transform_lb.rot.y(); ex.pose.orientation.z = transform_lb.rot.z();
  pub_extrinsic_.publish(ex);
}

// Estimator::ProcessCompactData (Estimator.cc:776-856): the /compact_data decode, the PointMapping base until the IMU is initialised
// (or the IMU-predicted transform afterwards) and ProcessLaserOdom are ONE library call
void EstimatorHip::ProcessCompactData(const sensor_msgs::PointCloud2ConstPtr &compact_data, const std_msgs::Header &header) {
  if (!est_) return;
  PointCloud cloud;
  pcl::fromROSMsg(*compact_data, cloud);
  Pack(cloud, scratch_);
  lio_transform_f T;
  const bool was_inited = stage_flag_ == INITED;
  if (!Check(lio_est_process_compact(est_, scratch_.data(), cloud.size(), header.stamp.toSec(), &T, &last_report_), "lio_est_process_compact")) return;
  transform_aft_mapped_ = FromPod(T);
  Refresh();
  convergence_flag_ = last_report_.convergence_flag != 0;
  // Headers_: a frame enters the window on events 1 (filling), 2, 3, 4; the oldest leaves once the window is full
  if (last_event_ != 0) {
    stamps_.push_back(header.stamp.toSec());
    if (stamps_.size() > estimator_config_.window_size + 1) stamps_.erase(stamps_.begin());
  }
  if (last_event_ == 3 && (estimator_config_.enable_deskew || estimator_config_.cutoff_deskew)) {
    // Estimator.cc:549-558: the scan-to-scan odometry stops de-skewing once the IMU does it
    ros::ServiceClient client = nh_.serviceClient<std_srvs::SetBool>("/enable_odom");
    std_srvs::SetBool srv;
    srv.request.data = 0;
    if (!client.call(srv)) LOG(FATAL) << "FAILED TO CALL TURNING OFF THE ORIGINAL LASER ODOM";
  }
  if (stage_flag_ == INITED && (last_event_ == 4 || (last_event_ == 3 && !was_inited))) {
    if (last_event_ == 3) stamps_.push_back(header.stamp.toSec());   // after the first slide Headers_ holds the newest stamp twice (:2646-2655)
    if (stamps_.size() > estimator_config_.window_size + 1) stamps_.erase(stamps_.begin());
    PublishAfterSolve(header, last_event_ == 4);
  }
}

// Estimator::ProcessLaserOdom (Estimator.cc:430-774) for a caller that fills laser_cloud_{surf,corner}_last_ itself
void EstimatorHip::ProcessLaserOdom(const Transform &transform_in, const std_msgs::Header &header) {
  if (!est_) return;
  lio_transform_f t;
  ToPod(transform_in, &t);
  Pack(*laser_cloud_surf_last_, scratch_);
  Pack(*laser_cloud_corner_last_, scratch2_);
  if (!Check(lio_est_process_laser_odom(est_, &t, scratch_.data(), laser_cloud_surf_last_->size(), scratch2_.data(), laser_cloud_corner_last_->size(),
                                        header.stamp.toSec(), &last_report_), "lio_est_process_laser_odom")) return;
  Refresh();
  if (stage_flag_ == INITED && last_event_ >= 3) PublishAfterSolve(header, last_event_ == 4);
}

void EstimatorHip::SolveOptimization() {
  if (!est_) return;
  if (Check(lio_est_solve_optimization(est_, &last_report_), "lio_est_solve_optimization")) Refresh();
}

void EstimatorHip::SlideWindow() {
  if (!est_) return;
  if (Check(lio_est_slide_window(est_), "lio_est_slide_window")) Refresh();
}

void EstimatorHip::RequestStop() {
  stop_.store(true);
  con_.notify_all();
}

// Estimator::ProcessEstimation (Estimator.cc:2668-2770): wait for paired measurements (the reference's GetMeasurements), feed the IMU
// samples of the interval — the one after the laser stamp interpolated onto it (:2708-2726) — then the /compact_data message
void EstimatorHip::ProcessEstimation() {
  while (true) {
    PairMeasurements measurements;
    {
      std::unique_lock<std::mutex> buf_lk(buf_mutex_);
      con_.wait(buf_lk, [&] { return (measurements = GetMeasurements()).size() != 0 || stop_.load(); });
    }
    if (measurements.empty()) return;   // stop requested and nothing left to pair
    std::lock_guard<std::mutex> batch(thread_mutex_);
    for (auto &measurement : measurements) {
      const CompactDataConstPtr &compact_data_msg = measurement.second;
      const double laser_time = compact_data_msg->header.stamp.toSec() + mm_config_.msg_time_delay;
      double acc[3] = {0, 0, 0}, gyr[3] = {0, 0, 0};
      for (auto &imu_msg : measurement.first) {
        const double imu_time = imu_msg->header.stamp.toSec();
        const double m_acc[3] = {imu_msg->linear_acceleration.x, imu_msg->linear_acceleration.y, imu_msg->linear_acceleration.z};
        const double m_gyr[3] = {imu_msg->angular_velocity.x, imu_msg->angular_velocity.y, imu_msg->angular_velocity.z};
        double dt;
        if (imu_time <= laser_time) {
          if (curr_time_ < 0) curr_time_ = imu_time;
          dt = imu_time - curr_time_;
          ROS_ASSERT(dt >= 0);
          curr_time_ = imu_time;
          for (int k = 0; k < 3; ++k) { acc[k] = m_acc[k]; gyr[k] = m_gyr[k]; }
        } else {
          const double dt_1 = laser_time - curr_time_, dt_2 = imu_time - laser_time;
          ROS_ASSERT(dt_1 >= 0);
          ROS_ASSERT(dt_2 >= 0);
          ROS_ASSERT(dt_1 + dt_2 > 0);
          curr_time_ = laser_time;
          const double w1 = dt_2 / (dt_1 + dt_2), w2 = dt_1 / (dt_1 + dt_2);
          for (int k = 0; k < 3; ++k) { acc[k] = w1 * acc[k] + w2 * m_acc[k]; gyr[k] = w1 * gyr[k] + w2 * m_gyr[k]; }
          dt = dt_1;
        }
        ProcessImu(dt, Vector3d(acc[0], acc[1], acc[2]), Vector3d(gyr[0], gyr[1], gyr[2]), imu_msg->header);
      }
      ProcessCompactData(compact_data_msg, compact_data_msg->header);
      processed_.fetch_add(1);
    }
  }
}

}  // namespace lio
