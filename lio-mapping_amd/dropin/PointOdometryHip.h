// PointOdometryHip.h — drop-in for lio::PointOdometry (include/point_processor/PointOdometry.h:99-223 of hyye/lio-mapping) with the
// MI355X library behind it.
//
// estimator_node.cc:147-151 constructs this class in place of lio::PointOdometry, calls SetupRos() and Spin() on its thread exactly
// as it does today: the five scan-registration topics arrive through the same handlers, HasNewData() pairs them by stamp (:220-229),
// and Process() hands the four feature clouds of a paired sweep to ONE library call (lio_odom_process: correspondence search, the
// point-to-line / point-to-plane rows, the 6 x 6 step with its degeneracy handling, <= 25 iterations, the accumulation into
// transform_sum_ and TransformToEnd of the two clouds kept for the next sweep — PointOdometry.cc:294-683).  /laser_odom_to_init,
// /laser_odom_to_last, the tf, and /compact_data (or the three cloud topics) are published as at :685-790; the full-resolution
// cloud — a pass-through for viewers and the map builder, on nobody's solve path — is carried to the sweep's end on the host
// with the reference's own per-point formula (:260-292).
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
  PointOdometryHip(float scan_period = 0.1, int io_ratio = 2, size_t num_max_iterations = 25);   // PointOdometry.cc:66-103
  ~PointOdometryHip();
  PointOdometryHip(const PointOdometryHip &) = delete;
  PointOdometryHip &operator=(const PointOdometryHip &) = delete;

  void SetupRos(ros::NodeHandle &nh);                                                             // :105-150
  void Reset();                                                                                   // :208-218
  void LaserCloudSharpHandler(const sensor_msgs::PointCloud2ConstPtr &corner_points_sharp_msg);   // :152-206
  void LaserCloudLessSharpHandler(const sensor_msgs::PointCloud2ConstPtr &corner_points_less_sharp_msg);
  void LaserCloudFlatHandler(const sensor_msgs::PointCloud2ConstPtr &surf_points_flat_msg);
  void LaserCloudLessFlatHandler(const sensor_msgs::PointCloud2ConstPtr &surf_points_less_flat_msg);
  void LaserFullCloudHandler(const sensor_msgs::PointCloud2ConstPtr &full_cloud_msg);
  bool HasNewData();                                                                              // :220-229
  size_t TransformToEnd(PointCloudPtr &cloud);                                                    // :260-292 (host: the pass-through cloud only)
  void Process();                                                                                 // :294-683
  void PublishResults();                                                                          // :685-790
  bool EnableOdom(std_srvs::SetBoolRequest &req, std_srvs::SetBoolResponse &res);                 // PointOdometry.h:126-131
  void Spin();                                                                                    // PointOdometry.h:133-147

  int last_error() const { return last_error_; }
  lio_odom *handle() { return odom_; }
  void set_no_deskew(bool on) { if (on != no_deskew_) { no_deskew_ = on; Recreate(); } }   // the ~no_deskew parameter, for callers without a parameter server
  // what the reference keeps private, readable here (the estimator node's tests and viewers look at them through accessors)
  const Transform &transform_es() const { return transform_es_; }
  const Transform &transform_sum() const { return transform_sum_; }
  long frame_count() const { return frame_count_; }
  const PointCloud &last_corner_cloud() const { return *last_corner_cloud_; }
  const PointCloud &last_surf_cloud() const { return *last_surf_cloud_; }

 private:
  bool Check(int rc, const char *what);
  void Recreate();
  static void Pack(const PointCloud &c, std::vector<float> &xyzi);
  void Fetch(int which, PointCloud &dst);

  float scan_period_, time_factor_;
  int io_ratio_;
  long frame_count_ = 0;
  bool system_inited_ = false;
  size_t num_max_iterations_;
  Transform transform_es_, transform_sum_;
  PointCloudPtr corner_points_sharp_, corner_points_less_sharp_, surf_points_flat_, surf_points_less_flat_, full_cloud_, last_corner_cloud_, last_surf_cloud_;
  ros::Time time_corner_points_sharp_, time_corner_points_less_sharp_, time_surf_points_flat_, time_surf_points_less_flat_, time_full_cloud_;
  bool new_corner_points_sharp_ = false, new_corner_points_less_sharp_ = false, new_surf_points_flat_ = false, new_surf_points_less_flat_ = false,
       new_full_cloud_ = false;
  nav_msgs::Odometry laser_odometry_msg_;
  tf::StampedTransform laser_odometry_trans_;
  ros::Publisher pub_laser_cloud_corner_last_, pub_laser_cloud_surf_last_, pub_full_cloud_, pub_diff_odometry_, pub_laser_odometry_, pub_compact_data_;
  tf::TransformBroadcaster tf_broadcaster_;
  ros::Subscriber sub_corner_points_sharp_, sub_corner_points_less_sharp_, sub_surf_points_flat_, sub_surf_points_less_flat_, sub_full_cloud_;
  ros::ServiceServer enable_odom_service_;
  bool is_ros_setup_ = false, compact_data_ = false, enable_odom_ = true, no_deskew_ = false;
  lio_odom *odom_ = nullptr;
  int last_error_ = LIO_OK;
  std::vector<float> b_sharp_, b_less_sharp_, b_flat_, b_less_flat_;
};

}  // namespace lio

#endif  // LIO_POINT_ODOMETRY_HIP_H_
