// C entry points over the REFERENCE's own PointProcessor (src/point_processor/PointProcessor.cc), compiled from the source where it
// lies against the stand-in headers of oracle/ref_shim (PCL's point / cloud containers, ROS message and node types that do nothing,
// glog macros; pcl::VoxelGrid forwards to the oracle's restatement).  TEST INFRASTRUCTURE, built by `make -C oracle ref` into
// oracle/_ref/libref_pointproc.so; tests/golden/make_ref_pointproc_vectors.py turns its outputs into committed vectors.
#include <cstring>

#include "point_processor/PointProcessor.h"

namespace {
struct Probe : public lio::PointProcessor {
  using lio::PointProcessor::PointProcessor;
  const lio::PointCloud &cloud(int which) const {
    switch (which) {
      case 0: return cloud_in_rings_;
      case 1: return corner_points_sharp_;
      case 2: return corner_points_less_sharp_;
      case 3: return surface_points_flat_;
      default: return surface_points_less_flat_;
    }
  }
};
}  // namespace

extern "C" {

// cfg = num_scan_subregions, num_curvature_regions, max_corner_sharp, max_corner_less_sharp, max_surf_flat, infer_start_ori;
// fcfg = surf_curv_th, less_flat_filter_size, scan_period, rad_diff
void *ref_pp_create(float lower, float upper, int rings, int uneven, const int *cfg, const double *fcfg) {
  Probe *p = new Probe(lower, upper, rings, uneven != 0);
  lio::PointProcessorConfig c;
  c.num_scan_subregions = cfg[0]; c.num_curvature_regions = cfg[1]; c.max_corner_sharp = cfg[2]; c.max_corner_less_sharp = cfg[3];
  c.max_surf_flat = cfg[4]; c.infer_start_ori_ = cfg[5] != 0;
  c.surf_curv_th = float(fcfg[0]); c.less_flat_filter_size = float(fcfg[1]); c.scan_period = fcfg[2]; c.rad_diff = fcfg[3];
  p->SetupConfig(c);
  return p;
}
void ref_pp_destroy(void *h) { delete static_cast<Probe *>(h); }
// one sweep: SetInputCloud + Process (PointToRing, ExtractFeaturePoints; PublishResults returns at once: ROS is not set up).
// ring = null: the elevation overload; else the PointXYZIR overload (the handle must have been created with uneven = 1)
void ref_pp_process(void *h, const float *xyzi, size_t n, const uint16_t *ring) {
  Probe *p = static_cast<Probe *>(h);
  if (!ring) {
    lio::PointCloudPtr c(new lio::PointCloud());
    for (size_t i = 0; i < n; ++i) { lio::PointT q; q.x = xyzi[4 * i]; q.y = xyzi[4 * i + 1]; q.z = xyzi[4 * i + 2]; q.intensity = xyzi[4 * i + 3]; c->push_back(q); }
    p->SetInputCloud(lio::PointCloudConstPtr(c));
  } else {
    pcl::PointCloud<lio::PointIR>::Ptr c(new pcl::PointCloud<lio::PointIR>());
    for (size_t i = 0; i < n; ++i) { lio::PointIR q; q.x = xyzi[4 * i]; q.y = xyzi[4 * i + 1]; q.z = xyzi[4 * i + 2]; q.intensity = xyzi[4 * i + 3]; q.ring = ring[i]; c->push_back(q); }
    p->SetInputCloud(c);
  }
  p->Process();
}
// which: 0 cloud_in_rings_ (the intensity scans in ring order: int(input intensity) + rel. time), 1 sharp, 2 less sharp, 3 flat,
// 4 less flat (after the voxel filter), 5 laser_scans in ring order (ring + rel. time: what the feature extraction works on)
size_t ref_pp_count(void *h, int which) {
  Probe *p = static_cast<Probe *>(h);
  if (which != 5) return p->cloud(which).size();
  size_t n = 0;
  for (const lio::PointCloudPtr &c : p->laser_scans) n += c->size();
  return n;
}
void ref_pp_get(void *h, int which, float *out) {
  Probe *p = static_cast<Probe *>(h);
  if (which == 5) {
    size_t k = 0;
    for (const lio::PointCloudPtr &c : p->laser_scans)
      for (size_t i = 0; i < c->size(); ++i, ++k) { out[4 * k] = (*c)[i].x; out[4 * k + 1] = (*c)[i].y; out[4 * k + 2] = (*c)[i].z; out[4 * k + 3] = (*c)[i].intensity; }
    return;
  }
  const lio::PointCloud &c = p->cloud(which);
  for (size_t i = 0; i < c.size(); ++i) { out[4 * i] = c[i].x; out[4 * i + 1] = c[i].y; out[4 * i + 2] = c[i].z; out[4 * i + 3] = c[i].intensity; }
}
// scan_ranges (first, last) per ring
void ref_pp_ranges(void *h, int rings, long long *out) {
  Probe *p = static_cast<Probe *>(h);
  for (int r = 0; r < rings && r < int(p->scan_ranges.size()); ++r) { out[2 * r] = (long long)p->scan_ranges[r].first; out[2 * r + 1] = (long long)p->scan_ranges[r].second; }
}

}  // extern "C"
