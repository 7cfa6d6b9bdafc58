// Stand-in for the few PCL types the reference's PointProcessor touches (oracle/ref_shim: test infrastructure): a point with
// x, y, z, intensity, a vector-like cloud with shared-pointer typedefs.  Containers only — no algorithm of PCL is imitated here
// except VoxelGrid (pcl/filters/voxel_grid.h), which forwards to the oracle's restatement and is therefore NOT independently pinned.
#pragma once
#include <cmath>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../Eigen/Eigen"   // (the real PCL headers bring Eigen in; the reference's utils/math_utils.h relies on that)
#ifndef EIGEN_ALIGN16
#define EIGEN_ALIGN16 __attribute__((aligned(16)))
#endif
#ifndef EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#endif
#define PCL_ADD_POINT4D union { float data[4]; struct { float x; float y; float z; }; }
#define POINT_CLOUD_REGISTER_POINT_STRUCT(name, fields)
#define pcl_isfinite(x) std::isfinite(x)
namespace pcl {
struct PointXYZI {
  PCL_ADD_POINT4D;
  float intensity;
  PointXYZI() : intensity(0.f) { x = y = z = 0.f; data[3] = 1.f; }
} EIGEN_ALIGN16;
struct PCLHeader { uint32_t seq = 0; uint64_t stamp = 0; std::string frame_id; };
template <typename PointT>
class PointCloud {
 public:
  typedef std::shared_ptr<PointCloud<PointT> > Ptr;
  typedef std::shared_ptr<const PointCloud<PointT> > ConstPtr;
  std::vector<PointT> points;
  PCLHeader header;
  uint32_t width = 0, height = 0;
  bool is_dense = true;
  size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  void clear() { points.clear(); width = height = 0; }
  void push_back(const PointT &p) { points.push_back(p); width = uint32_t(points.size()); height = 1; }
  void resize(size_t n) { points.resize(n); width = uint32_t(n); height = 1; }
  void reserve(size_t n) { points.reserve(n); }
  PointT &operator[](size_t i) { return points[i]; }
  const PointT &operator[](size_t i) const { return points[i]; }
  PointT &front() { return points.front(); }
  const PointT &front() const { return points.front(); }
  PointT &back() { return points.back(); }
  const PointT &back() const { return points.back(); }
  typename std::vector<PointT>::iterator begin() { return points.begin(); }
  typename std::vector<PointT>::iterator end() { return points.end(); }
  typename std::vector<PointT>::const_iterator begin() const { return points.begin(); }
  typename std::vector<PointT>::const_iterator end() const { return points.end(); }
  PointCloud &operator+=(const PointCloud &o) { points.insert(points.end(), o.points.begin(), o.points.end()); width = uint32_t(points.size()); height = 1; return *this; }
  Ptr makeShared() const { return Ptr(new PointCloud<PointT>(*this)); }
};
}  // namespace pcl
