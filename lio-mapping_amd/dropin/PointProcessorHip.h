// PointProcessorHip.h — drop-in for lio::PointProcessor (include/point_processor/PointProcessor.h:127-230 of hyye/lio-mapping) with the
// MI355X library behind it.
//
// processor_node.cc:66-83 constructs this class in place of lio::PointProcessor (`processor = PointProcessorHip(-24.9f, 2, 64);` — the
// class is move-assignable for that line), calls SetupConfig / SetupRos and lets the /velodyne_points callback drive it; the ROS-free
// sequence of test/test_point_processor/test_point_processor.cc:103-106 (SetInputCloud -> PointToRing -> ExtractFeaturePoints) works
// too.  What changes: ring binning, masks, curvature, the pick loops and the per-ring voxel filter (PointProcessor.cc:207-783) run
// inside liblio_hip.so through include/lio_c.h — ONE library call per sweep, issued by PointToRing().  After it the public members
// hold what the same members of lio::PointProcessor would hold: laser_scans (intensity = ring + rel. time), intensity_scans
// (int(input intensity) + rel. time), scan_ranges (inclusive first / last index per ring, :193-201); ExtractFeaturePoints() fills the
// four feature clouds (protected, published as the same five topics, :783-797).
//
// Builds inside the reference's catkin tree or, as the repo's tests do, against the stand-in headers of oracle/ref_shim
// (`make -C oracle ref` -> oracle/_ref/libdropin_frontend.so).
#ifndef LIO_POINT_PROCESSOR_HIP_H_
#define LIO_POINT_PROCESSOR_HIP_H_

#include <vector>

#include "point_processor/PointProcessor.h"   // PointProcessorConfig, PointT, PointIR, PointCloud, IndexRange (types only: PointProcessor.cc is not linked)
#include "lio_c.h"

namespace lio {

class PointProcessorHip {
 public:
  PointProcessorHip();
  PointProcessorHip(float lower_bound, float upper_bound, int num_rings, bool uneven = false);   // PointProcessor.cc:73-95
  ~PointProcessorHip();
  PointProcessorHip(PointProcessorHip &&o) noexcept;
  PointProcessorHip &operator=(PointProcessorHip &&o) noexcept;
  PointProcessorHip(const PointProcessorHip &) = delete;
  PointProcessorHip &operator=(const PointProcessorHip &) = delete;

  void Process();                                                                                  // :96-100
  void PointCloudHandler(const sensor_msgs::PointCloud2ConstPtr &raw_points_msg);                  // :102-121
  void SetupConfig(PointProcessorConfig config);                                                   // PointProcessor.h:141-143 (a new library handle)
  void SetupRos(ros::NodeHandle &nh);                                                              // :123-140
  void SetInputCloud(const PointCloudConstPtr &cloud_in, ros::Time time_in = ros::Time::now());    // :174-177
  void SetInputCloud(const pcl::PointCloud<PointIR>::Ptr &cloud_in, ros::Time time_in = ros::Time::now());   // :180-183
  void PointToRing();                                                                              // :185-205 (+ :207-536 in the library)
  void ExtractFeaturePoints();                                                                     // :647-783 (computed by the same library call)
  void PublishResults();                                                                           // :783-797

  int last_error() const { return last_error_; }   // LIO_OK or the code of the last library call (also logged)
  lio_pp *handle() { return pp_; }

  std::vector<PointCloudPtr> laser_scans;
  std::vector<PointCloudPtr> intensity_scans;
  std::vector<IndexRange> scan_ranges;

 protected:
  void Reset(const ros::Time &scan_time, const bool &is_new_sweep = true);   // :142-172
  bool Check(int rc, const char *what);
  void Recreate();

  ros::Time sweep_start_, scan_time_;
  float lower_bound_, upper_bound_;
  int num_rings_;
  PointProcessorConfig config_;
  PointCloudConstPtr cloud_ptr_;
  pcl::PointCloud<PointIR>::Ptr cloud_ir_ptr_;
  PointCloud cloud_in_rings_, corner_points_sharp_, corner_points_less_sharp_, surface_points_flat_, surface_points_less_flat_;
  ros::Subscriber sub_raw_points_;
  ros::Publisher pub_full_cloud_, pub_corner_points_sharp_, pub_corner_points_less_sharp_, pub_surf_points_flat_, pub_surf_points_less_flat_;
  bool is_ros_setup_ = false, uneven_ = false, processed_ = false;
  lio_pp *pp_ = nullptr;
  int last_error_ = LIO_OK;
  std::vector<float> scratch_;
  std::vector<uint16_t> rings_;
};

}  // namespace lio

#endif  // LIO_POINT_PROCESSOR_HIP_H_
