// PointOdometryHip.h — drop-in for lio::PointOdometry (include/point_processor/PointOdometry.h:99-223 of hyye/lio-mapping) with the
// MI355X library behind it.
//
// estimator_node.cc:147-151 constructs this class in place of lio::PointOdometry, calls SetupRos() and Spin() on its thread exactly
// as it does today: the five scan-registration topics arrive through the same five handlers, a sweep is complete when all five
// carry the same stamp (HasNewData, PointOdometry.cc:220-229), and Process() hands the four feature clouds of a complete sweep to
// ONE library call (lio_odom_process: correspondence search, point-to-line / point-to-plane rows, the 6 x 6 step with its
// degeneracy handling, <= 25 iterations, the accumulation into transform_sum_ and TransformToEnd of the two clouds kept for the
// next sweep — PointOdometry.cc:294-683).  /laser_odom_to_init, /laser_odom_to_last, the tf and /compact_data (or the three cloud
// topics) go out as at :685-790; the full-resolution cloud — a pass-through for viewers and the map builder, on nobody's solve
// path — is carried to the sweep's end on the host (TransformToEnd below).
//
// Builds inside the reference's catkin tree or against the stand-in headers of oracle/ref_shim (`make -C oracle ref` ->
// oracle/_ref/libdropin_frontend.so).
#ifndef LIO_POINT_ODOMETRY_HIP_H_
#define LIO_POINT_ODOMETRY_HIP_H_

#include <vector>

#include "point_processor/PointOdometry.h"   // Transform, PointT, PointCloudPtr, the message types (types only: PointOdometry.cc is not linked)
#include "lio_c.h"

namespace lio {

class PointOdometryHip {
 public:
  PointOdometryHip(float scan_period = 0.1, int io_ratio = 2, size_t num_max_iterations = 25);
  ~PointOdometryHip();
  PointOdometryHip(const PointOdometryHip &) = delete;
  PointOdometryHip &operator=(const PointOdometryHip &) = delete;

  // ---- the reference's public surface
  void SetupRos(ros::NodeHandle &nh);
  void Reset();
  void LaserCloudSharpHandler(const sensor_msgs::PointCloud2ConstPtr &msg) { Receive(kSharp, msg); }
  void LaserCloudLessSharpHandler(const sensor_msgs::PointCloud2ConstPtr &msg) { Receive(kLessSharp, msg); }
  void LaserCloudFlatHandler(const sensor_msgs::PointCloud2ConstPtr &msg) { Receive(kFlat, msg); }
  void LaserCloudLessFlatHandler(const sensor_msgs::PointCloud2ConstPtr &msg) { Receive(kLessFlat, msg); }
  void LaserFullCloudHandler(const sensor_msgs::PointCloud2ConstPtr &msg) { Receive(kFull, msg); }
  bool HasNewData();
  size_t TransformToEnd(PointCloudPtr &cloud);   // host: only the pass-through cloud comes here
  void Process();
  void PublishResults();
  bool EnableOdom(std_srvs::SetBoolRequest &req, std_srvs::SetBoolResponse &res);
  void Spin();

  // ---- beyond it
  int last_error() const { return last_error_; }   // LIO_OK or the code of the last library call (also logged)
  lio_odom *handle() { return odom_; }
  void set_no_deskew(bool on);                      // the ~no_deskew parameter, for callers without a parameter server
  // what the reference keeps private, readable here (tests and viewers)
  const Transform &transform_es() const { return sweep_motion_; }
  const Transform &transform_sum() const { return pose_in_init_; }
  long frame_count() const { return sweeps_done_; }
  const PointCloud &last_corner_cloud() const { return *kept_corner_; }
  const PointCloud &last_surf_cloud() const { return *kept_surf_; }

 private:
  enum Channel { kSharp = 0, kLessSharp, kFlat, kLessFlat, kFull, kChannels };
  struct Inbox { PointCloudPtr cloud; ros::Time stamp; bool fresh = false; };   // the latest message of a topic
  void Receive(Channel c, const sensor_msgs::PointCloud2ConstPtr &msg);
  bool Ok(int rc, const char *what);
  void OpenHandle();
  void Download(int which, PointCloud &into);

  Inbox in_[kChannels];
  float period_;
  int io_ratio_;
  size_t max_iterations_;
  bool ros_ready_ = false, compact_ = false, running_ = true, no_deskew_ = false, have_previous_ = false;
  long sweeps_done_ = 0;
  Transform sweep_motion_, pose_in_init_;        // transform_es_, transform_sum_ of the reference
  PointCloudPtr kept_corner_, kept_surf_;        // last_corner_cloud_, last_surf_cloud_
  lio_odom *odom_ = nullptr;
  int last_error_ = LIO_OK;
  std::vector<float> stage_[4];                  // xyzi of the four feature clouds on their way to the library

  nav_msgs::Odometry odom_msg_;
  tf::StampedTransform odom_tf_;
  tf::TransformBroadcaster tf_out_;
  ros::Publisher pub_corner_, pub_surf_, pub_full_, pub_to_last_, pub_to_init_, pub_compact_;
  ros::Subscriber sub_[kChannels];
  ros::ServiceServer enable_service_;
};

}  // namespace lio

#endif  // LIO_POINT_ODOMETRY_HIP_H_
