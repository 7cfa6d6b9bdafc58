// C entry points over the REFERENCE's own MapBuilder (src/map_builder/MapBuilder.cc over src/point_processor/PointMapping.cc), compiled
// from the sources where they lie against the stand-in headers of oracle/ref_shim.  TEST INFRASTRUCTURE (`make -C oracle ref` ->
// _ref/libref_mapbuilder.so).  What runs is the reference's four message handlers and HasNewData, ProcessMap (first-frame adoption of
// the odometry, Transform4DAssociateToMap or TransformAssociateToMap, the cube window, the stack / map assembly, the skip_count gate),
// OptimizeMap (the 4-DoF Gauss-Newton: correspondences and fits as in PointMapping, ConstrainedRotAxis, the constrained update, the
// degeneracy branch) or OptimizeTransformTobeMapped, Transform4DUpdate / TransformUpdate and UpdateMapDatabase.  Stood in: as for
// oracle/ref_mapping.cc, plus Eigen::AngleAxis (oracle/ref_shim/Eigen/Eigen, from Eigen's formula).
#include <cstring>

#define private public
#define protected public
#include "map_builder/MapBuilder.h"
#undef private
#undef protected

namespace {
void put(const lio::Transform &t, float *out) {
  out[0] = t.rot.x(); out[1] = t.rot.y(); out[2] = t.rot.z(); out[3] = t.rot.w();
  out[4] = t.pos.x(); out[5] = t.pos.y(); out[6] = t.pos.z();
}
std::shared_ptr<sensor_msgs::PointCloud2> cloud_msg(const float *xyzi, size_t n, double stamp) {
  std::shared_ptr<sensor_msgs::PointCloud2> m(new sensor_msgs::PointCloud2());
  if (n) m->xyzi.assign(xyzi, xyzi + 4 * n);
  m->header.stamp = ros::Time(stamp);
  return m;
}
const lio::PointCloud &cloud_of(lio::MapBuilder *p, int which) {
  switch (which) {
    case 0: return *p->laser_cloud_corner_stack_downsampled_;
    case 1: return *p->laser_cloud_surf_stack_downsampled_;
    case 2: return *p->laser_cloud_corner_from_map_;
    default: return *p->laser_cloud_surf_from_map_;
  }
}
void copy_out(const lio::PointCloud &c, float *out) {
  for (size_t i = 0; i < c.size(); ++i) { out[4 * i] = c[i].x; out[4 * i + 1] = c[i].y; out[4 * i + 2] = c[i].z; out[4 * i + 3] = c[i].intensity; }
}
}  // namespace

extern "C" {

// fp: corner_filter_size, surf_filter_size, map_filter_size, min_match_sq_dis, min_plane_dis
void *ref_mb_create(const float *fp, int enable_4d, int skip_count) {
  lio::MapBuilderConfig c;
  c.corner_filter_size = fp[0]; c.surf_filter_size = fp[1]; c.map_filter_size = fp[2]; c.min_match_sq_dis = fp[3]; c.min_plane_dis = fp[4];
  lio::MapBuilder *m = new lio::MapBuilder(c);
  m->enable_4d_ = enable_4d != 0;   // (what SetupRos reads from the parameter server, MapBuilder.cc:109-110)
  m->skip_count_ = skip_count;
  return m;
}
void ref_mb_destroy(void *h) { delete static_cast<lio::MapBuilder *>(h); }
// one frame as the odometry node publishes it: corner, surf, full cloud and /laser_odom_to_init, all with the same stamp; then ProcessMap()
void ref_mb_process(void *h, const float *corner, size_t nc, const float *surf, size_t ns, const float *full, size_t nf, const float *T7, double stamp) {
  lio::MapBuilder *p = static_cast<lio::MapBuilder *>(h);
  p->LaserCloudCornerLastHandler(cloud_msg(corner, nc, stamp));
  p->LaserCloudSurfLastHandler(cloud_msg(surf, ns, stamp));
  p->LaserFullCloudHandler(cloud_msg(full, nf, stamp));
  std::shared_ptr<nav_msgs::Odometry> od(new nav_msgs::Odometry());
  od->header.stamp = ros::Time(stamp);
  od->pose.pose.orientation.x = T7[0]; od->pose.pose.orientation.y = T7[1]; od->pose.pose.orientation.z = T7[2]; od->pose.pose.orientation.w = T7[3];
  od->pose.pose.position.x = T7[4]; od->pose.pose.position.y = T7[5]; od->pose.pose.position.z = T7[6];
  p->LaserOdometryHandler(od);
  p->ProcessMap();
}
// which: 0 transform_tobe_mapped_, 1 transform_aft_mapped_, 2 transform_bef_mapped_, 3 transform_sum_ (q = x y z w, then p)
void ref_mb_get_transform(void *h, int which, float *out7) {
  lio::MapBuilder *p = static_cast<lio::MapBuilder *>(h);
  put(which == 0 ? p->transform_tobe_mapped_ : which == 1 ? p->transform_aft_mapped_ : which == 2 ? p->transform_bef_mapped_ : p->transform_sum_.transform(), out7);
}
size_t ref_mb_count(void *h, int which) { return cloud_of(static_cast<lio::MapBuilder *>(h), which).size(); }
void ref_mb_get_cloud(void *h, int which, float *out) { copy_out(cloud_of(static_cast<lio::MapBuilder *>(h), which), out); }
int ref_mb_cube_state(void *h, int *center3, long long *valid_idx, int capacity) {
  lio::MapBuilder *p = static_cast<lio::MapBuilder *>(h);
  center3[0] = p->laser_cloud_cen_length_; center3[1] = p->laser_cloud_cen_width_; center3[2] = p->laser_cloud_cen_height_;
  const std::vector<size_t> &v = p->laser_cloud_valid_idx_;
  for (size_t i = 0; i < v.size() && int(i) < capacity; ++i) valid_idx[i] = (long long)v[i];
  return int(v.size());
}
size_t ref_mb_cube_count(void *h, int cls, long long idx) {
  lio::MapBuilder *p = static_cast<lio::MapBuilder *>(h);
  return (cls == 0 ? *p->laser_cloud_corner_array_[size_t(idx)] : *p->laser_cloud_surf_array_[size_t(idx)]).size();
}
void ref_mb_get_cube(void *h, int cls, long long idx, float *out) {
  lio::MapBuilder *p = static_cast<lio::MapBuilder *>(h);
  copy_out(cls == 0 ? *p->laser_cloud_corner_array_[size_t(idx)] : *p->laser_cloud_surf_array_[size_t(idx)], out);
}

// One keyframe of the batched refinement (BASELINE.json configs[4], lio_kf_batch_*): the reference's own Gauss-Newton loop —
// PointMapping::OptimizeTransformTobeMapped, or MapBuilder::OptimizeMap with four_dof — run on caller-supplied from-map clouds and
// down-sampled stacks from the pose T7_in (q = x y z w, then p).  point_on_z_axis_ is fixed before the loop as Process() / ProcessMap()
// do (PointMapping.cc:803-806).  T7_out = transform_tobe_mapped_ after the loop.
void ref_kf_refine(const float *fp, int four_dof, const float *corner_map, size_t ncm, const float *surf_map, size_t nsm, const float *corner_stack,
                   size_t ncs, const float *surf_stack, size_t nss, const float *T7_in, float *T7_out) {
  lio::MapBuilderConfig c;
  c.corner_filter_size = fp[0]; c.surf_filter_size = fp[1]; c.map_filter_size = fp[2]; c.min_match_sq_dis = fp[3]; c.min_plane_dis = fp[4];
  lio::MapBuilder m(c);
  auto fill = [](lio::PointCloudPtr &dst, const float *src, size_t n) {
    dst->clear();
    for (size_t i = 0; i < n; ++i) { lio::PointT p; p.x = src[4 * i]; p.y = src[4 * i + 1]; p.z = src[4 * i + 2]; p.intensity = src[4 * i + 3]; dst->push_back(p); }
  };
  fill(m.laser_cloud_corner_from_map_, corner_map, ncm);
  fill(m.laser_cloud_surf_from_map_, surf_map, nsm);
  fill(m.laser_cloud_corner_stack_downsampled_, corner_stack, ncs);
  fill(m.laser_cloud_surf_stack_downsampled_, surf_stack, nss);
  m.transform_tobe_mapped_ = lio::Transform(Eigen::Quaternionf(T7_in[3], T7_in[0], T7_in[1], T7_in[2]), Eigen::Vector3f(T7_in[4], T7_in[5], T7_in[6]));
  m.transform_sum_ = m.transform_tobe_mapped_;      // (OptimizeMap derives its — unused — constrained axis from it)
  m.transform_bef_mapped_ = m.transform_tobe_mapped_;
  m.point_on_z_axis_.x = 0.0; m.point_on_z_axis_.y = 0.0; m.point_on_z_axis_.z = 10.0;
  m.PointAssociateToMap(m.point_on_z_axis_, m.point_on_z_axis_, m.transform_tobe_mapped_);
  if (four_dof) m.OptimizeMap(); else m.OptimizeTransformTobeMapped();
  put(m.transform_tobe_mapped_, T7_out);
}

}  // extern "C"
