// Stand-in for pcl::ExtractIndices (oracle/ref_shim: test infrastructure): keeps, or with setNegative(true) drops, the listed
// points, in cloud order — as PCL's filter does for an unorganised cloud.
#pragma once
#include <memory>
#include <vector>

#include "../../utils/common_ros.h"
namespace pcl {
struct PointIndices { typedef std::shared_ptr<PointIndices> Ptr; std::vector<int> indices; };
template <typename PointT>
class ExtractIndices {
  typename PointCloud<PointT>::ConstPtr in_;
  PointIndices::Ptr idx_;
  bool negative_ = false;

 public:
  void setInputCloud(const typename PointCloud<PointT>::ConstPtr &c) { in_ = c; }
  void setIndices(const PointIndices::Ptr &i) { idx_ = i; }
  void setNegative(bool n) { negative_ = n; }
  void filter(PointCloud<PointT> &out) {
    std::vector<char> listed(in_->size(), 0);
    for (int i : idx_->indices) if (i >= 0 && size_t(i) < listed.size()) listed[i] = 1;
    PointCloud<PointT> res;
    if (negative_) { for (size_t i = 0; i < in_->size(); ++i) if (!listed[i]) res.push_back((*in_)[i]); }
    else { for (int i : idx_->indices) if (i >= 0 && size_t(i) < in_->size()) res.push_back((*in_)[i]); }
    res.header = in_->header;
    out = res;
  }
};
}  // namespace pcl
