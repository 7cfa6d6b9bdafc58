// C entry points over the REFERENCE's own PointMapping (src/point_processor/PointMapping.cc), compiled from the source where it
// lies against the stand-in headers of oracle/ref_shim.  TEST INFRASTRUCTURE (`make -C oracle ref` -> _ref/libref_mapping.so).
// What runs is the reference's /compact_data decoder, TransformAssociateToMap, the cube-window shifting and FOV selection, the
// stack / map assembly, OptimizeTransformTobeMapped (corner line fit, surf plane fit, scores, 6 x 6 system, degeneracy branch, update,
// termination), TransformUpdate and UpdateMapDatabase.  Stood in: pcl::VoxelGrid (forwards to the oracle's restatement), the kd-tree
// (exact search), Eigen's ColPivHouseholderQR / SelfAdjointEigenSolver (forwarded to the oracle's), Eigen's small dense / quaternion
// API, Sophus::SO3, the ROS plumbing.
#include <cstring>

#include "point_processor/PointMapping.h"

namespace {
struct Probe : public lio::PointMapping {
  using lio::PointMapping::PointMapping;
  const lio::Transform &tf(int which) const {
    switch (which) { case 0: return transform_tobe_mapped_; case 1: return transform_aft_mapped_; case 2: return transform_bef_mapped_; default: return transform_sum_; }
  }
  const lio::PointCloud &cloud(int which) const {
    switch (which) {
      case 0: return *laser_cloud_corner_stack_downsampled_;
      case 1: return *laser_cloud_surf_stack_downsampled_;
      case 2: return *laser_cloud_corner_from_map_;
      default: return *laser_cloud_surf_from_map_;
    }
  }
  const lio::PointCloud &cube(int cls, size_t idx) const { return cls == 0 ? *laser_cloud_corner_array_[idx] : *laser_cloud_surf_array_[idx]; }
  size_t ncubes() const { return laser_cloud_num_; }
  const std::vector<size_t> &valid() const { return laser_cloud_valid_idx_; }
  void center(int *c) const { c[0] = laser_cloud_cen_length_; c[1] = laser_cloud_cen_width_; c[2] = laser_cloud_cen_height_; }
  void thresholds(float sq_dis, float plane_dis) { min_match_sq_dis_ = sq_dis; min_plane_dis_ = plane_dis; }
};
void put(const lio::Transform &t, float *out) {
  out[0] = t.rot.x(); out[1] = t.rot.y(); out[2] = t.rot.z(); out[3] = t.rot.w();
  out[4] = t.pos.x(); out[5] = t.pos.y(); out[6] = t.pos.z();
}
}  // namespace

extern "C" {

void *ref_map_create(float scan_period, int max_iterations) { return new Probe(scan_period, size_t(max_iterations)); }
void ref_map_destroy(void *h) { delete static_cast<Probe *>(h); }
void ref_map_set_init_flag(void *h, int on) { static_cast<Probe *>(h)->SetInitFlag(on != 0); }
// one /compact_data message (header rows + corner + surf + full) through CompactDataHandler, then Process()
void ref_map_process_compact(void *h, const float *xyzi, size_t n, double stamp) {
  std::shared_ptr<sensor_msgs::PointCloud2> m(new sensor_msgs::PointCloud2());
  m->xyzi.assign(xyzi, xyzi + 4 * n);
  m->header.stamp = ros::Time(stamp);
  Probe *p = static_cast<Probe *>(h);
  p->CompactDataHandler(m);
  p->Process();
}
// which: 0 transform_tobe_mapped_, 1 transform_aft_mapped_, 2 transform_bef_mapped_, 3 transform_sum_ (q = x y z w, then p)
void ref_map_get_transform(void *h, int which, float *out7) { put(static_cast<Probe *>(h)->tf(which), out7); }
// which: 0 corner stack (down-sampled), 1 surf stack (down-sampled), 2 corner from map, 3 surf from map
size_t ref_map_count(void *h, int which) { return static_cast<Probe *>(h)->cloud(which).size(); }
void ref_map_get_cloud(void *h, int which, float *out) {
  const lio::PointCloud &c = static_cast<Probe *>(h)->cloud(which);
  for (size_t i = 0; i < c.size(); ++i) { out[4 * i] = c[i].x; out[4 * i + 1] = c[i].y; out[4 * i + 2] = c[i].z; out[4 * i + 3] = c[i].intensity; }
}
// cube center (3 ints), number of valid cubes and their indices
int ref_map_cube_state(void *h, int *center3, long long *valid_idx, int capacity) {
  Probe *p = static_cast<Probe *>(h);
  p->center(center3);
  const std::vector<size_t> &v = p->valid();
  for (size_t i = 0; i < v.size() && int(i) < capacity; ++i) valid_idx[i] = (long long)v[i];
  return int(v.size());
}
size_t ref_map_cube_count(void *h, int cls, long long idx) { return static_cast<Probe *>(h)->cube(cls, size_t(idx)).size(); }
void ref_map_get_cube(void *h, int cls, long long idx, float *out) {
  const lio::PointCloud &c = static_cast<Probe *>(h)->cube(cls, size_t(idx));
  for (size_t i = 0; i < c.size(); ++i) { out[4 * i] = c[i].x; out[4 * i + 1] = c[i].y; out[4 * i + 2] = c[i].z; out[4 * i + 3] = c[i].intensity; }
}

}  // extern "C"
