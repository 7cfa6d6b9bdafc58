// PointOdometryHip.cc — see PointOdometryHip.h.  Host glue: message handling, pairing, publishing; the step itself is lio_odom_process.
#include "PointOdometryHip.h"

#include <cmath>
#include <cstdio>

namespace lio {

PointOdometryHip::PointOdometryHip(float scan_period, int io_ratio, size_t num_max_iterations)
    : scan_period_(scan_period), time_factor_(1 / scan_period), io_ratio_(io_ratio), num_max_iterations_(num_max_iterations),
      corner_points_sharp_(new PointCloud()), corner_points_less_sharp_(new PointCloud()), surf_points_flat_(new PointCloud()),
      surf_points_less_flat_(new PointCloud()), full_cloud_(new PointCloud()), last_corner_cloud_(new PointCloud()), last_surf_cloud_(new PointCloud()) {
  laser_odometry_msg_.header.frame_id = "/camera_init";
  laser_odometry_msg_.child_frame_id = "/laser_odom";
  laser_odometry_trans_.frame_id_ = "/camera_init";
  laser_odometry_trans_.child_frame_id_ = "/camera";
  Recreate();
}

PointOdometryHip::~PointOdometryHip() { if (odom_) lio_odom_destroy(odom_); }

bool PointOdometryHip::Check(int rc, const char *what) {
  last_error_ = rc;
  if (rc == LIO_OK) return true;
  std::fprintf(stderr, "PointOdometryHip: %s failed with code %d\n", what, rc);
  return false;
}

void PointOdometryHip::Recreate() {
  if (odom_) lio_odom_destroy(odom_);
  odom_ = lio_odom_create(scan_period_, io_ratio_, int(num_max_iterations_), no_deskew_ ? 1 : 0);
  if (!odom_) Check(LIO_ERR_DEVICE, "lio_odom_create");
  else if (!enable_odom_) Check(lio_odom_enable(odom_, 0), "lio_odom_enable");
}

void PointOdometryHip::SetupRos(ros::NodeHandle &nh) {
  is_ros_setup_ = true;
  const bool had = no_deskew_;
  nh.param("compact_data", compact_data_, true);
  nh.param("no_deskew", no_deskew_, false);
  if (no_deskew_ != had) Recreate();   // the library takes no_deskew at creation (PointOdometry.cc:109)
  enable_odom_service_ = nh.advertiseService("/enable_odom", &PointOdometryHip::EnableOdom, this);
  if (compact_data_) {
    pub_compact_data_ = nh.advertise<sensor_msgs::PointCloud2>("/compact_data", 2);
  } else {
    pub_laser_cloud_corner_last_ = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_corner_last", 2);
    pub_laser_cloud_surf_last_ = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_surf_last", 2);
    pub_full_cloud_ = nh.advertise<sensor_msgs::PointCloud2>("/full_odom_cloud", 2);
  }
  pub_laser_odometry_ = nh.advertise<nav_msgs::Odometry>("/laser_odom_to_init", 5);
  pub_diff_odometry_ = nh.advertise<nav_msgs::Odometry>("/laser_odom_to_last", 5);
  sub_corner_points_sharp_ = nh.subscribe<sensor_msgs::PointCloud2>("/laser_cloud_sharp", 2, &PointOdometryHip::LaserCloudSharpHandler, this);
  sub_corner_points_less_sharp_ = nh.subscribe<sensor_msgs::PointCloud2>("/laser_cloud_less_sharp", 2, &PointOdometryHip::LaserCloudLessSharpHandler, this);
  sub_surf_points_flat_ = nh.subscribe<sensor_msgs::PointCloud2>("/laser_cloud_flat", 2, &PointOdometryHip::LaserCloudFlatHandler, this);
  sub_surf_points_less_flat_ = nh.subscribe<sensor_msgs::PointCloud2>("/laser_cloud_less_flat", 2, &PointOdometryHip::LaserCloudLessFlatHandler, this);
  sub_full_cloud_ = nh.subscribe<sensor_msgs::PointCloud2>("/full_cloud", 2, &PointOdometryHip::LaserFullCloudHandler, this);
}

bool PointOdometryHip::EnableOdom(std_srvs::SetBoolRequest &req, std_srvs::SetBoolResponse &res) {
  enable_odom_ = req.data > 0;
  if (odom_) Check(lio_odom_enable(odom_, enable_odom_ ? 1 : 0), "lio_odom_enable");
  res.success = true;
  return true;
}

static void take(const sensor_msgs::PointCloud2ConstPtr &msg, PointCloudPtr &cloud, ros::Time &stamp, bool &flag) {
  stamp = msg->header.stamp;
  cloud->clear();
  pcl::fromROSMsg(*msg, *cloud);
  std::vector<int> indices;
  pcl::removeNaNFromPointCloud(*cloud, *cloud, indices);
  flag = true;
}
void PointOdometryHip::LaserCloudSharpHandler(const sensor_msgs::PointCloud2ConstPtr &m) { take(m, corner_points_sharp_, time_corner_points_sharp_, new_corner_points_sharp_); }
void PointOdometryHip::LaserCloudLessSharpHandler(const sensor_msgs::PointCloud2ConstPtr &m) { take(m, corner_points_less_sharp_, time_corner_points_less_sharp_, new_corner_points_less_sharp_); }
void PointOdometryHip::LaserCloudFlatHandler(const sensor_msgs::PointCloud2ConstPtr &m) { take(m, surf_points_flat_, time_surf_points_flat_, new_surf_points_flat_); }
void PointOdometryHip::LaserCloudLessFlatHandler(const sensor_msgs::PointCloud2ConstPtr &m) { take(m, surf_points_less_flat_, time_surf_points_less_flat_, new_surf_points_less_flat_); }
void PointOdometryHip::LaserFullCloudHandler(const sensor_msgs::PointCloud2ConstPtr &m) { take(m, full_cloud_, time_full_cloud_, new_full_cloud_); }

void PointOdometryHip::Reset() {
  new_corner_points_sharp_ = new_corner_points_less_sharp_ = new_surf_points_flat_ = new_surf_points_less_flat_ = new_full_cloud_ = false;
}

bool PointOdometryHip::HasNewData() {
  return new_corner_points_sharp_ && new_corner_points_less_sharp_ && new_surf_points_flat_ && new_surf_points_less_flat_ && new_full_cloud_ &&
         std::fabs((time_corner_points_less_sharp_ - time_corner_points_sharp_).toSec()) < 0.005 &&
         std::fabs((time_surf_points_flat_ - time_corner_points_sharp_).toSec()) < 0.005 &&
         std::fabs((time_surf_points_less_flat_ - time_corner_points_sharp_).toSec()) < 0.005 &&
         std::fabs((time_full_cloud_ - time_corner_points_sharp_).toSec()) < 0.005;
}

// PointOdometry.cc:260-292, for the pass-through full-resolution cloud
size_t PointOdometryHip::TransformToEnd(PointCloudPtr &cloud) {
  const size_t cloud_size = cloud->points.size();
  for (size_t i = 0; i < cloud_size; i++) {
    PointT &point = cloud->points[i];
    float s = time_factor_ * (point.intensity - int(point.intensity));
    if (no_deskew_) s = 0;
    point.x -= s * transform_es_.pos.x();
    point.y -= s * transform_es_.pos.y();
    point.z -= s * transform_es_.pos.z();
    point.intensity = int(point.intensity);
    Eigen::Quaternionf q_id, q_s, q_e;
    q_e = transform_es_.rot;
    q_id.setIdentity();
    q_s = q_id.slerp(s, q_e);
    RotatePoint(q_s.conjugate(), point);
    RotatePoint(q_e, point);
    point.x += transform_es_.pos.x();
    point.y += transform_es_.pos.y();
    point.z += transform_es_.pos.z();
  }
  return cloud_size;
}

void PointOdometryHip::Pack(const PointCloud &c, std::vector<float> &xyzi) {
  xyzi.resize(4 * c.size());
  for (size_t i = 0; i < c.size(); ++i) { xyzi[4 * i] = c[i].x; xyzi[4 * i + 1] = c[i].y; xyzi[4 * i + 2] = c[i].z; xyzi[4 * i + 3] = c[i].intensity; }
}
void PointOdometryHip::Fetch(int which, PointCloud &dst) {
  const size_t n = lio_odom_get_last_cloud(odom_, which, nullptr);
  std::vector<float> buf(4 * n);
  if (n) lio_odom_get_last_cloud(odom_, which, buf.data());
  dst.clear();
  for (size_t k = 0; k < n; ++k) {
    PointT p;
    p.x = buf[4 * k]; p.y = buf[4 * k + 1]; p.z = buf[4 * k + 2]; p.intensity = buf[4 * k + 3];
    dst.push_back(p);
  }
}

void PointOdometryHip::Process() {
  if (!HasNewData() || !odom_) return;
  Reset();
  const bool first = !system_inited_;
  Pack(*corner_points_sharp_, b_sharp_); Pack(*corner_points_less_sharp_, b_less_sharp_);
  Pack(*surf_points_flat_, b_flat_); Pack(*surf_points_less_flat_, b_less_flat_);
  lio_transform_f t_sum, t_es;
  if (!Check(lio_odom_process(odom_, b_sharp_.data(), corner_points_sharp_->size(), b_less_sharp_.data(), corner_points_less_sharp_->size(), b_flat_.data(),
                              surf_points_flat_->size(), b_less_flat_.data(), surf_points_less_flat_->size(), &t_sum, &t_es, nullptr, nullptr), "lio_odom_process")) return;
  Fetch(0, *last_corner_cloud_);
  Fetch(1, *last_surf_cloud_);
  system_inited_ = true;
  if (first) return;   // :302-310: the first sweep only becomes "last"
  ++frame_count_;
  transform_sum_.rot = Eigen::Quaternionf(t_sum.q[3], t_sum.q[0], t_sum.q[1], t_sum.q[2]);
  transform_sum_.pos = Eigen::Vector3f(t_sum.p[0], t_sum.p[1], t_sum.p[2]);
  transform_es_.rot = Eigen::Quaternionf(t_es.q[3], t_es.q[0], t_es.q[1], t_es.q[2]);
  transform_es_.pos = Eigen::Vector3f(t_es.p[0], t_es.p[1], t_es.p[2]);
  PublishResults();
}

void PointOdometryHip::PublishResults() {
  if (!is_ros_setup_) return;
  geometry_msgs::Quaternion geo_quat;
  geo_quat.x = transform_sum_.rot.x(); geo_quat.y = transform_sum_.rot.y(); geo_quat.z = transform_sum_.rot.z(); geo_quat.w = transform_sum_.rot.w();
  laser_odometry_msg_.header.stamp = time_corner_points_sharp_;
  laser_odometry_msg_.pose.pose.orientation = geo_quat;
  laser_odometry_msg_.pose.pose.position.x = transform_sum_.pos.x();
  laser_odometry_msg_.pose.pose.position.y = transform_sum_.pos.y();
  laser_odometry_msg_.pose.pose.position.z = transform_sum_.pos.z();
  pub_laser_odometry_.publish(laser_odometry_msg_);
  laser_odometry_trans_.stamp_ = time_corner_points_sharp_;
  laser_odometry_trans_.setRotation(tf::Quaternion(geo_quat.x, geo_quat.y, geo_quat.z, geo_quat.w));
  laser_odometry_trans_.setOrigin(tf::Vector3(transform_sum_.pos.x(), transform_sum_.pos.y(), transform_sum_.pos.z()));
  tf_broadcaster_.sendTransform(laser_odometry_trans_);
  geo_quat.x = transform_es_.rot.x(); geo_quat.y = transform_es_.rot.y(); geo_quat.z = transform_es_.rot.z(); geo_quat.w = transform_es_.rot.w();
  laser_odometry_msg_.pose.pose.orientation = geo_quat;
  laser_odometry_msg_.pose.pose.position.x = transform_es_.pos.x();
  laser_odometry_msg_.pose.pose.position.y = transform_es_.pos.y();
  laser_odometry_msg_.pose.pose.position.z = transform_es_.pos.z();
  pub_diff_odometry_.publish(laser_odometry_msg_);
  if (io_ratio_ < 2 || frame_count_ % io_ratio_ == 1) {
    const ros::Time sweepTime = time_corner_points_sharp_;
    if (enable_odom_) TransformToEnd(full_cloud_);
    if (compact_data_) {
      PointCloud compact_data;
      PointT compact_point;
      compact_point.x = transform_sum_.pos.x(); compact_point.y = transform_sum_.pos.y(); compact_point.z = transform_sum_.pos.z();
      compact_data.push_back(compact_point);
      compact_point.x = transform_sum_.rot.x(); compact_point.y = transform_sum_.rot.y(); compact_point.z = transform_sum_.rot.z();
      compact_point.intensity = transform_sum_.rot.w();
      compact_data.push_back(compact_point);
      compact_point.x = last_corner_cloud_->size(); compact_point.y = last_surf_cloud_->size(); compact_point.z = full_cloud_->size();
      compact_data.push_back(compact_point);
      compact_data += (*last_corner_cloud_);
      compact_data += (*last_surf_cloud_);
      compact_data += (*full_cloud_);
      PublishCloudMsg(pub_compact_data_, compact_data, sweepTime, "/camera");
    } else {
      PublishCloudMsg(pub_laser_cloud_corner_last_, *last_corner_cloud_, sweepTime, "/camera");
      PublishCloudMsg(pub_laser_cloud_surf_last_, *last_surf_cloud_, sweepTime, "/camera");
      PublishCloudMsg(pub_full_cloud_, *full_cloud_, sweepTime, "/camera");
    }
  }
}

void PointOdometryHip::Spin() {
  ros::Rate rate(200);
  bool status = ros::ok();
  while (status) {
    ros::spinOnce();
    Process();
    status = ros::ok();
    rate.sleep();
  }
}

}  // namespace lio
