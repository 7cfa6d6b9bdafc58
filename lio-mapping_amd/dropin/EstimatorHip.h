// EstimatorHip.h — drop-in for lio::Estimator (include/imu_processor/Estimator.h:110-299 of hyye/lio-mapping) with the
// MI355X library behind it.
//
// What stays the reference's: MeasurementManager (queues, mutexes, GetMeasurements pairing: MeasurementManager.cc:54-140) is
// INHERITED, not rewritten — estimator_node.cc:142-153 constructs this class in place of lio::Estimator, calls SetupRos() and runs
// ProcessEstimation() on its own thread exactly as it does today; EstimatorConfig / MeasurementManagerConfig / Transform /
// CircularBuffer are the reference's types.  What changes: everything Estimator.cc computes (ProcessImu's propagation and
// pre-integration, the PointMapping base until the IMU is initialised, the initialisation, BuildLocalMap, the factors, the solve,
// the marginalization, SlideWindow) runs inside liblio_hip.so through include/lio_c.h.  After every call the public state members
// below hold what the same members of lio::Estimator would hold (same names, same CircularBuffer indexing), so PublishResults-style
// readers keep working, and the three odometry topics + /extrinsic_lb are published as at Estimator.cc:395-424,728-758,2343-2353.
//
// Builds inside the reference's catkin tree (ROS + PCL + Eigen) or, as the repo's tests do, against the stand-in headers of
// oracle/ref_shim (`make -C oracle ref` -> oracle/_ref/libdropin_estimator.so, which also compiles the reference's own
// MeasurementManager.cc where it lies).
#ifndef LIO_ESTIMATOR_HIP_H_
#define LIO_ESTIMATOR_HIP_H_

#include <atomic>
#include <vector>

#include <geometry_msgs/PoseStamped.h>
#include <nav_msgs/Odometry.h>
#include <std_srvs/SetBool.h>

#include "imu_processor/Estimator.h"   // EstimatorConfig, EstimatorStageFlag, Transform, PointCloudPtr (types only: Estimator.cc is not linked)
#include "lio_c.h"

namespace lio {

class EstimatorHip : public MeasurementManager {
 public:
  EstimatorHip();
  explicit EstimatorHip(EstimatorConfig config, MeasurementManagerConfig mm_config = MeasurementManagerConfig());
  ~EstimatorHip();
  EstimatorHip(const EstimatorHip &) = delete;
  EstimatorHip &operator=(const EstimatorHip &) = delete;

  void ClearState();                                             // Estimator.cc:231-291: a fresh library handle
  void SetupRos(ros::NodeHandle &nh) override;                   // Estimator.cc:293-336
  void SetupAllEstimatorConfig(const EstimatorConfig &config, const MeasurementManagerConfig &mm_config);   // Estimator.cc:145-229

  void ProcessEstimation();                                      // Estimator.cc:2668-2770 (thread B of estimator_node.cc:153)
  void ProcessImu(double dt, const Vector3d &linear_acceleration, const Vector3d &angular_velocity, const std_msgs::Header &header);
  void ProcessLaserOdom(const Transform &transform_in, const std_msgs::Header &header);   // implicit inputs: laser_cloud_{surf,corner}_last_
  void ProcessCompactData(const sensor_msgs::PointCloud2ConstPtr &compact_data, const std_msgs::Header &header);
  void SolveOptimization();                                      // Estimator.cc:1648
  void SlideWindow();                                            // Estimator.cc:2570

  // Not in the reference (its loop never returns): ProcessEstimation() leaves its loop once this has been called and the
  // queues hold no complete pairing.  processed_count() = /compact_data messages ProcessEstimation has finished.
  void RequestStop();
  size_t processed_count() const { return processed_.load(); }
  lio_est *handle() { return est_; }                             // for callers that want an entry point this class does not wrap
  int last_error() const { return last_error_; }                 // LIO_OK or the code the last library call returned (also logged)

  // ceres::Solver::Options the reference hard-codes in SolveOptimization (Estimator.cc:1916,1921); read by ClearState() (a new handle)
  int max_num_iterations_ = 10;
  double max_solver_time_in_seconds_ = 0.10;

  // ---- state, under the reference's names (refreshed from the library after every Process* / Solve / Slide call)
  EstimatorStageFlag stage_flag_ = NOT_INITED;
  EstimatorConfig estimator_config_;
  size_t cir_buf_count_ = 0;
  int extrinsic_stage_ = 2;
  CircularBuffer<Vector3d> Ps_{16}, Vs_{16}, Bas_{16}, Bgs_{16};
  CircularBuffer<Matrix3d> Rs_{16};
  Transform transform_lb_{Eigen::Quaternionf(1, 0, 0, 0), Eigen::Vector3f(0, 0, -0.1)};
  Matrix3d R_WI_;
  Vector3d g_vec_;
  Transform transform_aft_mapped_;                               // what ProcessCompactData handed to ProcessLaserOdom (Estimator.cc:846)
  bool convergence_flag_ = false;
  lio_solve_report last_report_;                                 // stage times under the reference's TicToc names, costs, iterations
  int last_event_ = 0;                                           // lio_est_get_stage: 0 skipped, 1 filling, 2 init failed, 3 initialised, 4 solved
  // implicit inputs of ProcessLaserOdom when a caller drives it directly (PointMapping's members in the reference)
  PointCloudPtr laser_cloud_surf_last_, laser_cloud_corner_last_;

  nav_msgs::Odometry predict_odom_, laser_odom_, local_odom_;    // last published (Estimator.h:254-262)

 private:
  void CreateHandle();
  void Refresh();                                                // library -> the state members above
  void PublishAfterSolve(const std_msgs::Header &header, bool odometry_topics);
  bool Check(int rc, const char *what);
  static void Pack(const PointCloud &cloud, std::vector<float> &xyzi);

  lio_est *est_ = nullptr;
  std::vector<float> scratch_, scratch2_;
  std::vector<double> stamps_;                                   // Headers_: stamp of every frame in the window (for /local_laser_odom)
  std::atomic<bool> stop_{false};
  std::atomic<size_t> processed_{0};
  int last_error_ = 0;
  ros::Publisher pub_predict_odom_, pub_laser_odom_, pub_local_odom_, pub_extrinsic_;
};

}  // namespace lio

#endif  // LIO_ESTIMATOR_HIP_H_
