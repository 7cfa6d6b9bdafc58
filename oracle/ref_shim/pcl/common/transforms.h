// pcl::transformPointCloud stand-in (oracle/ref_shim: test infrastructure): x' = R x + t per point with a 4 x 4 or affine transform,
// the arithmetic order of PCL's own loop (row by row, three products summed left to right, then the translation).
#pragma once
#include "../point_types.h"
namespace pcl {
template <typename PointT, typename T>
void transformPointCloud(const PointCloud<PointT> &in, PointCloud<PointT> &out, const Eigen::Transform<T, 3, Eigen::Affine> &tf) {
  const PointCloud<PointT> src(in);
  out = src;
  const Eigen::Matrix<T, 3, 3> &R = tf.linear();
  const Eigen::Matrix<T, 3, 1> &t = tf.translation();
  for (size_t i = 0; i < src.size(); ++i) {
    const PointT &p = src[i];
    out[i].x = float(R(0, 0) * p.x + R(0, 1) * p.y + R(0, 2) * p.z + t(0));
    out[i].y = float(R(1, 0) * p.x + R(1, 1) * p.y + R(1, 2) * p.z + t(1));
    out[i].z = float(R(2, 0) * p.x + R(2, 1) * p.y + R(2, 2) * p.z + t(2));
  }
}
}  // namespace pcl
