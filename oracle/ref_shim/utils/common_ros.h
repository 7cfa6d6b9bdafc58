// Stand-in for the reference's utils/common_ros.h (ROS + PCL headers) and for the ROS / glog names its sources mention: logging
// macros that swallow their arguments, message / publisher / node-handle types that do nothing.  oracle/ref_shim: test infrastructure.
#pragma once
#include <iostream>
#include <memory>
#include <string>

#include "../pcl/point_types.h"

struct RefShimNullStream {
  template <typename T> RefShimNullStream &operator<<(const T &) { return *this; }
  RefShimNullStream &operator<<(std::ostream &(*)(std::ostream &)) { return *this; }
};
#ifndef DLOG
#define DLOG(severity) RefShimNullStream()
#endif
#ifndef LOG
#define LOG(severity) RefShimNullStream()
#endif
#ifndef ROS_DEBUG
#define ROS_DEBUG(...) do { } while (0)
#endif
#ifndef ROS_DEBUG_STREAM
#define ROS_DEBUG_STREAM(args) do { } while (0)
#endif

namespace ros {
struct Time {
  double sec = 0;
  static Time now() { return Time(); }
  double toSec() const { return sec; }
};
struct Publisher { template <typename M> void publish(const M &) const {} };
struct Subscriber {};
struct NodeHandle {
  template <typename M> Publisher advertise(const std::string &, int) { return Publisher(); }
  template <typename M, typename C> Subscriber subscribe(const std::string &, int, void (C::*)(const std::shared_ptr<const M> &), C *) { return Subscriber(); }
};
}  // namespace ros
namespace std_msgs {
struct Header { ros::Time stamp; std::string frame_id; };
struct Float32 { float data = 0.f; };
}  // namespace std_msgs
namespace sensor_msgs {
struct PointCloud2 { std_msgs::Header header; };
typedef std::shared_ptr<const PointCloud2> PointCloud2ConstPtr;
}  // namespace sensor_msgs
namespace pcl {
template <typename PointT> void fromROSMsg(const sensor_msgs::PointCloud2 &, PointCloud<PointT> &) {}
template <typename PointT> void toROSMsg(const PointCloud<PointT> &, sensor_msgs::PointCloud2 &) {}
template <typename PointT> void removeNaNFromPointCloud(const PointCloud<PointT> &in, PointCloud<PointT> &out, std::vector<int> &idx) {
  out.clear(); idx.clear();
  for (size_t i = 0; i < in.size(); ++i)
    if (std::isfinite(in[i].x) && std::isfinite(in[i].y) && std::isfinite(in[i].z)) { out.push_back(in[i]); idx.push_back(int(i)); }
}
}  // namespace pcl
namespace lio {
template <typename PointT>
inline void PublishCloudMsg(ros::Publisher &, const pcl::PointCloud<PointT> &, const ros::Time &, std::string) {}
}  // namespace lio
