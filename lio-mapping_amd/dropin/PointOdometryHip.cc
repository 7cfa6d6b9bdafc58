// PointOdometryHip.cc — see PointOdometryHip.h.  Host glue only (inboxes, pairing, publishing); the step itself is lio_odom_process.
#include "PointOdometryHip.h"

#include <cmath>
#include <cstdio>

namespace lio {

namespace {
const char *const kTopics[5] = {"/laser_cloud_sharp", "/laser_cloud_less_sharp", "/laser_cloud_flat", "/laser_cloud_less_flat", "/full_cloud"};

Transform FromC(const lio_transform_f &t) {
  Transform out;
  out.rot = Eigen::Quaternionf(t.q[3], t.q[0], t.q[1], t.q[2]);   // lio_c.h: q = x y z w
  out.pos = Eigen::Vector3f(t.p[0], t.p[1], t.p[2]);
  return out;
}
void FillPose(const Transform &t, geometry_msgs::Pose &pose) {
  pose.orientation.x = t.rot.x(); pose.orientation.y = t.rot.y(); pose.orientation.z = t.rot.z(); pose.orientation.w = t.rot.w();
  pose.position.x = t.pos.x(); pose.position.y = t.pos.y(); pose.position.z = t.pos.z();
}
PointT HeaderPoint(float x, float y, float z, float i) {
  PointT p;
  p.x = x; p.y = y; p.z = z; p.intensity = i;
  return p;
}
}  // namespace

PointOdometryHip::PointOdometryHip(float scan_period, int io_ratio, size_t num_max_iterations)
    : period_(scan_period), io_ratio_(io_ratio), max_iterations_(num_max_iterations), kept_corner_(new PointCloud()), kept_surf_(new PointCloud()) {
  for (Inbox &b : in_) b.cloud.reset(new PointCloud());
  odom_msg_.header.frame_id = "/camera_init";      // PointOdometry.cc:86-90
  odom_msg_.child_frame_id = "/laser_odom";
  odom_tf_.frame_id_ = "/camera_init";
  odom_tf_.child_frame_id_ = "/camera";
  OpenHandle();
}

PointOdometryHip::~PointOdometryHip() {
  if (odom_) lio_odom_destroy(odom_);
}

bool PointOdometryHip::Ok(int rc, const char *what) {
  last_error_ = rc;
  if (rc != LIO_OK) std::fprintf(stderr, "PointOdometryHip: %s failed with code %d\n", what, rc);
  return rc == LIO_OK;
}

void PointOdometryHip::OpenHandle() {
  if (odom_) lio_odom_destroy(odom_);
  odom_ = lio_odom_create(period_, io_ratio_, int(max_iterations_), no_deskew_ ? 1 : 0);   // (no_deskew is fixed at creation)
  if (!odom_) { Ok(LIO_ERR_DEVICE, "lio_odom_create"); return; }
  if (!running_) Ok(lio_odom_enable(odom_, 0), "lio_odom_enable");
}

void PointOdometryHip::set_no_deskew(bool on) {
  if (on == no_deskew_) return;
  no_deskew_ = on;
  OpenHandle();
}

void PointOdometryHip::SetupRos(ros::NodeHandle &nh) {   // PointOdometry.cc:105-150: same parameters, topics and queue sizes
  ros_ready_ = true;
  bool no_deskew = false;
  nh.param("compact_data", compact_, true);
  nh.param("no_deskew", no_deskew, false);
  set_no_deskew(no_deskew);
  enable_service_ = nh.advertiseService("/enable_odom", &PointOdometryHip::EnableOdom, this);
  if (compact_) {
    pub_compact_ = nh.advertise<sensor_msgs::PointCloud2>("/compact_data", 2);
  } else {
    pub_corner_ = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_corner_last", 2);
    pub_surf_ = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_surf_last", 2);
    pub_full_ = nh.advertise<sensor_msgs::PointCloud2>("/full_odom_cloud", 2);
  }
  pub_to_init_ = nh.advertise<nav_msgs::Odometry>("/laser_odom_to_init", 5);
  pub_to_last_ = nh.advertise<nav_msgs::Odometry>("/laser_odom_to_last", 5);
  typedef void (PointOdometryHip::*Handler)(const sensor_msgs::PointCloud2ConstPtr &);
  const Handler handlers[kChannels] = {&PointOdometryHip::LaserCloudSharpHandler, &PointOdometryHip::LaserCloudLessSharpHandler,
                                       &PointOdometryHip::LaserCloudFlatHandler, &PointOdometryHip::LaserCloudLessFlatHandler,
                                       &PointOdometryHip::LaserFullCloudHandler};
  for (int c = 0; c < kChannels; ++c) sub_[c] = nh.subscribe<sensor_msgs::PointCloud2>(kTopics[c], 2, handlers[c], this);
}

bool PointOdometryHip::EnableOdom(std_srvs::SetBoolRequest &req, std_srvs::SetBoolResponse &res) {   // PointOdometry.h:126-131
  running_ = req.data > 0;
  if (odom_) Ok(lio_odom_enable(odom_, running_ ? 1 : 0), "lio_odom_enable");
  res.success = true;
  return true;
}

// PointOdometry.cc:152-206: the five handlers differ in the member they fill
void PointOdometryHip::Receive(Channel c, const sensor_msgs::PointCloud2ConstPtr &msg) {
  Inbox &box = in_[c];
  box.stamp = msg->header.stamp;
  box.cloud->clear();
  pcl::fromROSMsg(*msg, *box.cloud);
  std::vector<int> kept;
  pcl::removeNaNFromPointCloud(*box.cloud, *box.cloud, kept);
  box.fresh = true;
}

void PointOdometryHip::Reset() {
  for (Inbox &b : in_) b.fresh = false;
}

bool PointOdometryHip::HasNewData() {   // every topic fresh and within 5 ms of the sharp corners' stamp
  for (const Inbox &b : in_) {
    if (!b.fresh) return false;
    if (std::fabs((b.stamp - in_[kSharp].stamp).toSec()) >= 0.005) return false;
  }
  return true;
}

// PointOdometry.cc:260-292 for one cloud: a point measured at fraction s of the sweep is taken back to the sweep's start by the
// partial motion (translation s p, rotation slerp(identity, q, s)) and forward to its end by the whole motion; the intensity loses
// its time part
size_t PointOdometryHip::TransformToEnd(PointCloudPtr &cloud) {
  const Eigen::Quaternionf whole = sweep_motion_.rot;
  const Eigen::Vector3f shift = sweep_motion_.pos;
  const float per_second = 1.0f / period_;
  Eigen::Quaternionf identity;
  identity.setIdentity();
  for (PointT &pt : cloud->points) {
    const int ring = int(pt.intensity);
    const float s = no_deskew_ ? 0.0f : per_second * (pt.intensity - ring);
    pt.x -= s * shift.x(); pt.y -= s * shift.y(); pt.z -= s * shift.z();
    pt.intensity = ring;
    const Eigen::Quaternionf partial = identity.slerp(s, whole);
    RotatePoint(partial.conjugate(), pt);
    RotatePoint(whole, pt);
    pt.x += shift.x(); pt.y += shift.y(); pt.z += shift.z();
  }
  return cloud->points.size();
}

void PointOdometryHip::Download(int which, PointCloud &into) {
  const size_t n = lio_odom_get_last_cloud(odom_, which, nullptr);
  std::vector<float> xyzi(4 * n);
  if (n) lio_odom_get_last_cloud(odom_, which, xyzi.data());
  into.clear();
  for (size_t k = 0; k < n; ++k) into.push_back(HeaderPoint(xyzi[4 * k], xyzi[4 * k + 1], xyzi[4 * k + 2], xyzi[4 * k + 3]));
}

void PointOdometryHip::Process() {
  if (!odom_ || !HasNewData()) return;
  Reset();
  size_t counts[4];
  for (int c = 0; c < 4; ++c) {
    const PointCloud &src = *in_[c].cloud;
    counts[c] = src.size();
    stage_[c].resize(4 * counts[c]);
    for (size_t i = 0; i < counts[c]; ++i) {
      stage_[c][4 * i] = src[i].x; stage_[c][4 * i + 1] = src[i].y; stage_[c][4 * i + 2] = src[i].z; stage_[c][4 * i + 3] = src[i].intensity;
    }
  }
  lio_transform_f in_init, over_sweep;
  const int rc = lio_odom_process(odom_, stage_[kSharp].data(), counts[kSharp], stage_[kLessSharp].data(), counts[kLessSharp], stage_[kFlat].data(), counts[kFlat],
                                  stage_[kLessFlat].data(), counts[kLessFlat], &in_init, &over_sweep, nullptr, nullptr);
  if (!Ok(rc, "lio_odom_process")) return;
  Download(0, *kept_corner_);
  Download(1, *kept_surf_);
  if (!have_previous_) {   // PointOdometry.cc:302-310: the first sweep only becomes "last"; nothing is published
    have_previous_ = true;
    return;
  }
  ++sweeps_done_;
  pose_in_init_ = FromC(in_init);
  sweep_motion_ = FromC(over_sweep);
  PublishResults();
}

void PointOdometryHip::PublishResults() {   // PointOdometry.cc:685-790
  if (!ros_ready_) return;
  const ros::Time stamp = in_[kSharp].stamp;
  odom_msg_.header.stamp = stamp;
  FillPose(pose_in_init_, odom_msg_.pose.pose);
  pub_to_init_.publish(odom_msg_);
  odom_tf_.stamp_ = stamp;
  odom_tf_.setRotation(tf::Quaternion(pose_in_init_.rot.x(), pose_in_init_.rot.y(), pose_in_init_.rot.z(), pose_in_init_.rot.w()));
  odom_tf_.setOrigin(tf::Vector3(pose_in_init_.pos.x(), pose_in_init_.pos.y(), pose_in_init_.pos.z()));
  tf_out_.sendTransform(odom_tf_);
  FillPose(sweep_motion_, odom_msg_.pose.pose);
  pub_to_last_.publish(odom_msg_);

  const bool clouds_due = io_ratio_ < 2 || sweeps_done_ % io_ratio_ == 1;   // the input / output ratio of the cloud topics
  if (!clouds_due) return;
  PointCloudPtr &full = in_[kFull].cloud;
  if (running_) TransformToEnd(full);
  if (!compact_) {
    PublishCloudMsg(pub_corner_, *kept_corner_, stamp, "/camera");
    PublishCloudMsg(pub_surf_, *kept_surf_, stamp, "/camera");
    PublishCloudMsg(pub_full_, *full, stamp, "/camera");
    return;
  }
  // /compact_data: position | rotation (x y z, w in the intensity) | the three sizes | corner, surf, full clouds (:732-764)
  PointCloud packed;
  packed.push_back(HeaderPoint(pose_in_init_.pos.x(), pose_in_init_.pos.y(), pose_in_init_.pos.z(), 0.f));
  packed.push_back(HeaderPoint(pose_in_init_.rot.x(), pose_in_init_.rot.y(), pose_in_init_.rot.z(), pose_in_init_.rot.w()));
  packed.push_back(HeaderPoint(float(kept_corner_->size()), float(kept_surf_->size()), float(full->size()), pose_in_init_.rot.w()));   // (the reference reuses one point: w stays in the intensity)
  packed += *kept_corner_;
  packed += *kept_surf_;
  packed += *full;
  PublishCloudMsg(pub_compact_, packed, stamp, "/camera");
}

void PointOdometryHip::Spin() {   // PointOdometry.h:133-147
  ros::Rate rate(200);
  for (bool alive = ros::ok(); alive; alive = ros::ok()) {
    ros::spinOnce();
    Process();
    rate.sleep();
  }
}

}  // namespace lio
