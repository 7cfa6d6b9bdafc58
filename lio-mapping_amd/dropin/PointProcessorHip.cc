// PointProcessorHip.cc — see PointProcessorHip.h.  Host glue only: every number comes out of liblio_hip.so.
#include "PointProcessorHip.h"

#include <cstdio>
#include <cstring>
#include <utility>

namespace lio {

PointProcessorHip::PointProcessorHip() : PointProcessorHip(-15.0f, 15.0f, 16) {}

PointProcessorHip::PointProcessorHip(float lower_bound, float upper_bound, int num_rings, bool uneven)
    : lower_bound_(lower_bound), upper_bound_(upper_bound), num_rings_(num_rings), uneven_(uneven) {
  for (int i = 0; i < num_rings_; ++i) {
    laser_scans.push_back(PointCloudPtr(new PointCloud()));
    intensity_scans.push_back(PointCloudPtr(new PointCloud()));
  }
  Recreate();
}

PointProcessorHip::~PointProcessorHip() { if (pp_) lio_pp_destroy(pp_); }

PointProcessorHip::PointProcessorHip(PointProcessorHip &&o) noexcept { *this = std::move(o); }
PointProcessorHip &PointProcessorHip::operator=(PointProcessorHip &&o) noexcept {
  if (this == &o) return *this;
  if (pp_) lio_pp_destroy(pp_);
  laser_scans = std::move(o.laser_scans); intensity_scans = std::move(o.intensity_scans); scan_ranges = std::move(o.scan_ranges);
  sweep_start_ = o.sweep_start_; scan_time_ = o.scan_time_; lower_bound_ = o.lower_bound_; upper_bound_ = o.upper_bound_; num_rings_ = o.num_rings_;
  config_ = o.config_; cloud_ptr_ = o.cloud_ptr_; cloud_ir_ptr_ = o.cloud_ir_ptr_;
  cloud_in_rings_ = o.cloud_in_rings_; corner_points_sharp_ = o.corner_points_sharp_; corner_points_less_sharp_ = o.corner_points_less_sharp_;
  surface_points_flat_ = o.surface_points_flat_; surface_points_less_flat_ = o.surface_points_less_flat_;
  is_ros_setup_ = o.is_ros_setup_; uneven_ = o.uneven_; processed_ = o.processed_; last_error_ = o.last_error_;
  pp_ = o.pp_; o.pp_ = nullptr;
  return *this;
}

bool PointProcessorHip::Check(int rc, const char *what) {
  last_error_ = rc;
  if (rc == LIO_OK) return true;
  std::fprintf(stderr, "PointProcessorHip: %s failed with code %d\n", what, rc);
  return false;
}

// a library handle for the current bounds and configuration (PointProcessorConfig -> lio_pp_config, field by field)
void PointProcessorHip::Recreate() {
  if (pp_) { lio_pp_destroy(pp_); pp_ = nullptr; }
  lio_pp_config c;
  lio_pp_default_config(&c);
  c.scan_period = float(config_.scan_period); c.num_scan_subregions = config_.num_scan_subregions; c.num_curvature_regions = config_.num_curvature_regions;
  c.surf_curv_th = config_.surf_curv_th; c.max_corner_sharp = config_.max_corner_sharp; c.max_corner_less_sharp = config_.max_corner_less_sharp;
  c.max_surf_flat = config_.max_surf_flat; c.less_flat_filter_size = config_.less_flat_filter_size;
  c.infer_start_ori = config_.infer_start_ori_ ? 1 : 0; c.rad_diff = config_.rad_diff;
  const int chk = lio_pp_check_config(lower_bound_, upper_bound_, num_rings_, &c);
  if (!Check(chk, "lio_pp_check_config")) return;
  pp_ = lio_pp_create(lower_bound_, upper_bound_, num_rings_, &c);
  if (!pp_) Check(LIO_ERR_DEVICE, "lio_pp_create");
}

void PointProcessorHip::SetupConfig(PointProcessorConfig config) {
  config_ = config;
  Recreate();
}

void PointProcessorHip::SetupRos(ros::NodeHandle &nh) {
  is_ros_setup_ = true;
  sub_raw_points_ = nh.subscribe<sensor_msgs::PointCloud2>("/velodyne_points", 2, &PointProcessorHip::PointCloudHandler, this);
  pub_full_cloud_ = nh.advertise<sensor_msgs::PointCloud2>("/full_cloud", 2);
  pub_corner_points_sharp_ = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_sharp", 2);
  pub_corner_points_less_sharp_ = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_less_sharp", 2);
  pub_surf_points_flat_ = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_flat", 2);
  pub_surf_points_less_flat_ = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_less_flat", 2);
}

void PointProcessorHip::PointCloudHandler(const sensor_msgs::PointCloud2ConstPtr &raw_points_msg) {
  if (!uneven_) {
    PointCloud laser_cloud_in;
    pcl::fromROSMsg(*raw_points_msg, laser_cloud_in);
    SetInputCloud(PointCloudConstPtr(new PointCloud(laser_cloud_in)), raw_points_msg->header.stamp);
  } else {
    pcl::PointCloud<PointIR> laser_cloud_in;
    pcl::fromROSMsg(*raw_points_msg, laser_cloud_in);
    SetInputCloud(pcl::PointCloud<PointIR>::Ptr(new pcl::PointCloud<PointIR>(laser_cloud_in)), raw_points_msg->header.stamp);
  }
  Process();
}

void PointProcessorHip::Reset(const ros::Time &scan_time, const bool &is_new_sweep) {
  scan_time_ = scan_time;
  if (!is_new_sweep) return;
  sweep_start_ = scan_time_;
  cloud_in_rings_.clear(); corner_points_sharp_.clear(); corner_points_less_sharp_.clear(); surface_points_flat_.clear(); surface_points_less_flat_.clear();
  scan_ranges.clear();
  for (PointCloudPtr &c : laser_scans) c->clear();
  for (PointCloudPtr &c : intensity_scans) c->clear();
  processed_ = false;
}

void PointProcessorHip::SetInputCloud(const PointCloudConstPtr &cloud_in, ros::Time time_in) {
  Reset(time_in);
  cloud_ptr_ = cloud_in;
}
void PointProcessorHip::SetInputCloud(const pcl::PointCloud<PointIR>::Ptr &cloud_in, ros::Time time_in) {
  Reset(time_in);
  cloud_ir_ptr_ = cloud_in;
}

void PointProcessorHip::Process() {
  PointToRing();
  ExtractFeaturePoints();
  PublishResults();
}

// The sweep goes through the library here: lio_pp_process / lio_pp_process_rings = PointToRing + ExtractFeaturePoints in one call.
void PointProcessorHip::PointToRing() {
  if (!pp_) return;
  size_t n = 0;
  if (!uneven_) {
    if (!cloud_ptr_) return;
    n = cloud_ptr_->size();
    scratch_.resize(4 * n);
    for (size_t i = 0; i < n; ++i) {
      const PointT &p = (*cloud_ptr_)[i];
      scratch_[4 * i] = p.x; scratch_[4 * i + 1] = p.y; scratch_[4 * i + 2] = p.z; scratch_[4 * i + 3] = p.intensity;
    }
    if (!Check(lio_pp_process(pp_, scratch_.data(), n), "lio_pp_process")) return;
  } else {
    if (!cloud_ir_ptr_) return;
    n = cloud_ir_ptr_->size();
    scratch_.resize(4 * n); rings_.resize(n);
    for (size_t i = 0; i < n; ++i) {
      const PointIR &p = (*cloud_ir_ptr_)[i];
      scratch_[4 * i] = p.x; scratch_[4 * i + 1] = p.y; scratch_[4 * i + 2] = p.z; scratch_[4 * i + 3] = p.intensity; rings_[i] = p.ring;
    }
    if (!Check(lio_pp_process_rings(pp_, scratch_.data(), rings_.data(), n), "lio_pp_process_rings")) return;
  }
  processed_ = true;
  // laser_scans / intensity_scans / scan_ranges / cloud_in_rings_ from the ring-ordered cloud (:191-201)
  const size_t nr = lio_pp_count(pp_, LIO_PP_RINGS);
  std::vector<float> ring_cloud(4 * nr), ring_int(nr);
  std::vector<int32_t> off(num_rings_ + 1, 0);
  if (nr && (!Check(lio_pp_get_cloud(pp_, LIO_PP_RINGS, ring_cloud.data()), "lio_pp_get_cloud") ||
             !Check(lio_pp_get_ring_intensity(pp_, ring_int.data()), "lio_pp_get_ring_intensity"))) return;
  if (!Check(lio_pp_get_ring_offsets(pp_, off.data()), "lio_pp_get_ring_offsets")) return;
  size_t cloud_size = 0;
  for (int r = 0; r < num_rings_; ++r) {
    PointCloud &ls = *laser_scans[r], &is = *intensity_scans[r];
    for (int32_t k = off[r]; k < off[r + 1]; ++k) {
      PointT p;
      p.x = ring_cloud[4 * k]; p.y = ring_cloud[4 * k + 1]; p.z = ring_cloud[4 * k + 2]; p.intensity = ring_cloud[4 * k + 3];
      ls.push_back(p);
      p.intensity = ring_int[k];
      is.push_back(p);
    }
    cloud_in_rings_ += is;
    IndexRange range(cloud_size, 0);
    cloud_size += ls.size();
    range.second = (cloud_size > 0 ? cloud_size - 1 : 0);
    scan_ranges.push_back(range);
  }
}

void PointProcessorHip::ExtractFeaturePoints() {
  if (!pp_ || !processed_) return;
  const struct { int which; PointCloud *dst; } outs[4] = {{LIO_PP_SHARP, &corner_points_sharp_}, {LIO_PP_LESS_SHARP, &corner_points_less_sharp_},
                                                          {LIO_PP_FLAT, &surface_points_flat_}, {LIO_PP_LESS_FLAT, &surface_points_less_flat_}};
  for (const auto &o : outs) {
    const size_t n = lio_pp_count(pp_, o.which);
    std::vector<float> buf(4 * n);
    if (n && !Check(lio_pp_get_cloud(pp_, o.which, buf.data()), "lio_pp_get_cloud")) return;
    o.dst->clear();
    for (size_t k = 0; k < n; ++k) {
      PointT p;
      p.x = buf[4 * k]; p.y = buf[4 * k + 1]; p.z = buf[4 * k + 2]; p.intensity = buf[4 * k + 3];
      o.dst->push_back(p);
    }
  }
}

void PointProcessorHip::PublishResults() {
  if (!is_ros_setup_) return;
  PublishCloudMsg(pub_full_cloud_, cloud_in_rings_, sweep_start_, config_.capture_frame_id);
  PublishCloudMsg(pub_corner_points_sharp_, corner_points_sharp_, sweep_start_, config_.capture_frame_id);
  PublishCloudMsg(pub_corner_points_less_sharp_, corner_points_less_sharp_, sweep_start_, config_.capture_frame_id);
  PublishCloudMsg(pub_surf_points_flat_, surface_points_flat_, sweep_start_, config_.capture_frame_id);
  PublishCloudMsg(pub_surf_points_less_flat_, surface_points_less_flat_, sweep_start_, config_.capture_frame_id);
}

}  // namespace lio
