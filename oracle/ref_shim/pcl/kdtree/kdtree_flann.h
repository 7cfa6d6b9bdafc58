// pcl::KdTreeFLANN stand-in: EXACT k nearest neighbours by exhaustive search, ordered by (squared distance, index) — what FLANN's
// kd-tree returns with eps = 0 up to the order of exact ties.  The search structure is third-party; its results are defined by
// the metric, which is what this reproduces.  oracle/ref_shim: test infrastructure.
#pragma once
#include <algorithm>
#include <memory>
#include <vector>

#include "../point_types.h"
namespace pcl {
template <typename PointT>
class KdTreeFLANN {
  typename PointCloud<PointT>::ConstPtr cloud_;

 public:
  typedef std::shared_ptr<KdTreeFLANN<PointT> > Ptr;
  void setInputCloud(const typename PointCloud<PointT>::ConstPtr &c) { cloud_ = c; }
  void setInputCloud(const typename PointCloud<PointT>::Ptr &c) { cloud_ = c; }
  int nearestKSearch(const PointT &p, int k, std::vector<int> &idx, std::vector<float> &sqd) const {
    const size_t n = cloud_ ? cloud_->size() : 0;
    std::vector<std::pair<float, int> > d(n);
    for (size_t i = 0; i < n; ++i) {
      const PointT &q = (*cloud_)[i];
      const float dx = q.x - p.x, dy = q.y - p.y, dz = q.z - p.z;
      d[i] = std::make_pair(dx * dx + dy * dy + dz * dz, int(i));
    }
    const size_t kk = std::min<size_t>(size_t(k), n);
    std::partial_sort(d.begin(), d.begin() + kk, d.end());
    idx.resize(kk); sqd.resize(kk);
    for (size_t i = 0; i < kk; ++i) { sqd[i] = d[i].first; idx[i] = d[i].second; }
    return int(kk);
  }
};
}  // namespace pcl
