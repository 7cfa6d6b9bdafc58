// pcl::VoxelGrid stand-in: the interface the reference calls (setInputCloud / setLeafSize / filter), forwarding to the ORACLE'S
// restatement of PCL's filter (oracle/cloud.h).  So the less-flat cloud that comes out of the reference's PointProcessor here is not
// an independent check of the voxel filter — everything in front of it (ring split, rel-time, curvature, masks, picks, labels) is.
#pragma once
#include "../../../cloud.h"
#include "../point_types.h"
namespace pcl {
template <typename PointT>
class VoxelGrid {
  typename PointCloud<PointT>::ConstPtr in_;
  float leaf_ = 0.f;

 public:
  void setInputCloud(const typename PointCloud<PointT>::ConstPtr &c) { in_ = c; }
  void setInputCloud(const typename PointCloud<PointT>::Ptr &c) { in_ = c; }
  void setLeafSize(float lx, float, float) { leaf_ = lx; }
  void filter(PointCloud<PointT> &out) {
    orc::Cloud a, b;
    for (const PointT &p : in_->points) a.push_back({p.x, p.y, p.z, p.intensity});
    orc::VoxelGrid(a, leaf_, b);
    out.clear();
    for (const orc::P4 &q : b) { PointT p; p.x = q.x; p.y = q.y; p.z = q.z; p.intensity = q.i; out.push_back(p); }
  }
};
}  // namespace pcl
