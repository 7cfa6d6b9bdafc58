// C entry points over the REFERENCE's own PointOdometry (src/point_processor/PointOdometry.cc), compiled from the source where it
// lies against the stand-in headers of oracle/ref_shim.  TEST INFRASTRUCTURE (`make -C oracle ref` -> _ref/libref_odometry.so).
// What runs is the reference's message handlers, HasNewData, TransformToStart / TransformToEnd, the correspondence search rules,
// the point-to-line / point-to-plane coefficients, the 6 x 6 system, the degeneracy handling, the update and termination rules,
// the accumulation into transform_sum_ and the /compact_data packing.  What is stood in for: the kd-tree (exact search), Eigen's
// ColPivHouseholderQR / SelfAdjointEigenSolver (forwarded to the oracle's restatements, oracle/ref_shim/decomp_from_oracle.h),
// Eigen's small dense / quaternion API, Sophus::SO3, the ROS plumbing.
#include <chrono>
#include <cmath>
#include <cstring>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#define private public   // the class keeps its state private and has no accessors; the layout is unchanged
#include "point_processor/PointOdometry.h"
#undef private

namespace {
struct Odo {
  lio::PointOdometry o;
  ros::NodeHandle nh;
  Odo(float sp, int io, size_t it) : o(sp, io, it) {}
};
sensor_msgs::PointCloud2ConstPtr msg_of(const float *xyzi, size_t n, double stamp) {
  std::shared_ptr<sensor_msgs::PointCloud2> m(new sensor_msgs::PointCloud2());
  m->xyzi.assign(xyzi, xyzi + 4 * n);
  m->header.stamp = ros::Time(stamp);
  return m;
}
void put(const lio::Transform &t, float *out) {
  out[0] = t.rot.x(); out[1] = t.rot.y(); out[2] = t.rot.z(); out[3] = t.rot.w();
  out[4] = t.pos.x(); out[5] = t.pos.y(); out[6] = t.pos.z();
}
}  // namespace

extern "C" {

void *ref_odom_create(float scan_period, int io_ratio, int max_iterations, int no_deskew) {
  Odo *h = new Odo(scan_period, io_ratio, size_t(max_iterations));
  h->o.SetupRos(h->nh);            // compact_data = true, no_deskew = false: the launch files' defaults
  h->o.no_deskew_ = no_deskew != 0;
  h->o.Reset();
  return h;
}
void ref_odom_destroy(void *h) { delete static_cast<Odo *>(h); }
void ref_odom_enable(void *h, int on) { static_cast<Odo *>(h)->o.enable_odom_ = on != 0; }
// one sweep: the five messages of the processor node, then Process()
void ref_odom_process(void *h, const float *sharp, size_t n1, const float *less_sharp, size_t n2, const float *flat, size_t n3, const float *less_flat,
                      size_t n4, const float *full, size_t n5, double stamp) {
  lio::PointOdometry &o = static_cast<Odo *>(h)->o;
  ros::PublishedLog::last_cloud().xyzi.clear();
  o.LaserCloudSharpHandler(msg_of(sharp, n1, stamp));
  o.LaserCloudLessSharpHandler(msg_of(less_sharp, n2, stamp));
  o.LaserCloudFlatHandler(msg_of(flat, n3, stamp));
  o.LaserCloudLessFlatHandler(msg_of(less_flat, n4, stamp));
  o.LaserFullCloudHandler(msg_of(full, n5, stamp));
  o.Process();
}
// q = x y z w, then p: transform_es_ and transform_sum_
void ref_odom_get(void *h, float *T_es, float *T_sum, long *frame_count) {
  lio::PointOdometry &o = static_cast<Odo *>(h)->o;
  put(o.transform_es_, T_es); put(o.transform_sum_, T_sum);
  *frame_count = o.frame_count_;
}
// which: 0 last_corner_cloud_, 1 last_surf_cloud_, 2 the /compact_data message published by the last Process() (empty if none)
size_t ref_odom_count(void *h, int which) {
  lio::PointOdometry &o = static_cast<Odo *>(h)->o;
  if (which == 0) return o.last_corner_cloud_->size();
  if (which == 1) return o.last_surf_cloud_->size();
  return ros::PublishedLog::last_cloud().xyzi.size() / 4;
}
void ref_odom_get_cloud(void *h, int which, float *out) {
  lio::PointOdometry &o = static_cast<Odo *>(h)->o;
  if (which == 2) { const std::vector<float> &v = ros::PublishedLog::last_cloud().xyzi; if (!v.empty()) std::memcpy(out, v.data(), v.size() * sizeof(float)); return; }
  const lio::PointCloud &c = which == 0 ? *o.last_corner_cloud_ : *o.last_surf_cloud_;
  for (size_t i = 0; i < c.size(); ++i) { out[4 * i] = c[i].x; out[4 * i + 1] = c[i].y; out[4 * i + 2] = c[i].z; out[4 * i + 3] = c[i].intensity; }
}

}  // extern "C"
