// Stand-in for the reference's utils/common_ros.h (ROS + PCL headers) and for the ROS / glog names its sources mention: logging
// macros that swallow their arguments, message / publisher / node-handle types that do nothing.  oracle/ref_shim: test infrastructure.
#pragma once
#include <cstdlib>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

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
inline RefShimNullStream RefShimCheck(bool ok) { if (!ok) std::abort(); return RefShimNullStream(); }
#ifndef LOG_ASSERT
#define LOG_ASSERT(c) RefShimCheck(bool(c))
#endif
#ifndef CHECK
#define CHECK(c) RefShimCheck(bool(c))
#endif
#ifndef ROS_BREAK
#define ROS_BREAK() std::abort()
#endif
#ifndef ROS_ASSERT
#define ROS_ASSERT(c) do { if (!(c)) std::abort(); } while (0)
#endif
#ifndef ROS_WARN
#define ROS_WARN(...) do { } while (0)
#define ROS_INFO(...) do { } while (0)
#define ROS_ERROR(...) do { } while (0)
#define ROS_INFO_STREAM(a) do { } while (0)
#define ROS_WARN_STREAM(a) do { } while (0)
#define ROS_ERROR_STREAM(a) do { } while (0)
#endif
#ifndef ROS_DEBUG_STREAM
#define ROS_DEBUG_STREAM(args) do { } while (0)
#endif

namespace boost {
using std::shared_ptr;
using std::make_shared;
struct mutex { void lock() {} void unlock() {} };
}  // namespace boost
namespace ros {
struct Duration { double sec = 0; double toSec() const { return sec; } };
struct Time {
  double sec = 0;
  Time() {}
  explicit Time(double s) : sec(s) {}
  static Time now() { return Time(); }
  double toSec() const { return sec; }
  Time &fromSec(double s) { sec = s; return *this; }
  Duration operator-(const Time &o) const { Duration d; d.sec = sec - o.sec; return d; }
  bool operator==(const Time &o) const { return sec == o.sec; }
};
// a publisher that keeps the last message of each kind it was given (the stand-in messages are plain structs)
struct PublishedLog;
struct Publisher { template <typename M> void publish(const M &m) const; };
struct Subscriber {};
struct ServiceServer {};
struct TransportHints { TransportHints &tcpNoDelay() { return *this; } };
struct ServiceClient { template <typename S> bool call(S &) { return true; } };
struct Rate { explicit Rate(double) {} void sleep() {} };
inline bool ok() { return false; }
inline void spinOnce() {}
struct NodeHandle {
  template <typename M> Publisher advertise(const std::string &, int) { return Publisher(); }
  template <typename M, typename C> Subscriber subscribe(const std::string &, int, void (C::*)(const std::shared_ptr<const M> &), C *) { return Subscriber(); }
  template <typename M, typename C> Subscriber subscribe(const std::string &, int, void (C::*)(const std::shared_ptr<const M> &), C *, const TransportHints &) { return Subscriber(); }
  template <typename C, typename Rq, typename Rs> ServiceServer advertiseService(const std::string &, bool (C::*)(Rq &, Rs &), C *) { return ServiceServer(); }
  template <typename T> bool param(const std::string &, T &v, const T &dflt) const { v = dflt; return false; }
  template <typename S> ServiceClient serviceClient(const std::string &) { return ServiceClient(); }
};
}  // namespace ros
namespace std_msgs {
struct Header { ros::Time stamp; std::string frame_id; unsigned seq = 0; };
struct Float32 { float data = 0.f; };
}  // namespace std_msgs
struct geometry_msgs_Quaternion_fwd { double x = 0, y = 0, z = 0, w = 1; };
namespace sensor_msgs {
// the payload of a cloud message here: x, y, z, intensity per point (what the reference's nodes exchange)
struct PointCloud2 { std_msgs::Header header; std::vector<float> xyzi; };
typedef std::shared_ptr<const PointCloud2> PointCloud2ConstPtr;
struct Vec3 { double x = 0, y = 0, z = 0; };
struct Imu { std_msgs::Header header; Vec3 linear_acceleration, angular_velocity; geometry_msgs_Quaternion_fwd orientation; };
typedef std::shared_ptr<const Imu> ImuConstPtr;
}  // namespace sensor_msgs
namespace geometry_msgs {
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Point { double x = 0, y = 0, z = 0; };
struct Pose { Point position; Quaternion orientation; };
struct PoseWithCovariance { Pose pose; };
struct PoseStamped { std_msgs::Header header; Pose pose; };
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Twist { Vector3 linear, angular; };
struct TwistWithCovariance { Twist twist; };
}  // namespace geometry_msgs
namespace nav_msgs {
struct Odometry { typedef std::shared_ptr<const Odometry> ConstPtr; std_msgs::Header header; std::string child_frame_id; geometry_msgs::PoseWithCovariance pose; geometry_msgs::TwistWithCovariance twist; };
typedef std::shared_ptr<const Odometry> OdometryConstPtr;
struct Path { std_msgs::Header header; std::vector<geometry_msgs::PoseStamped> poses; };
}  // namespace nav_msgs
namespace visualization_msgs {
struct Marker { std_msgs::Header header; };
struct MarkerArray { std::vector<Marker> markers; };
}  // namespace visualization_msgs
namespace std_srvs {
struct SetBoolRequest { bool data = false; };
struct SetBoolResponse { bool success = false; std::string message; };
struct SetBool { SetBoolRequest request; SetBoolResponse response; };
}  // namespace std_srvs
namespace tf {
struct Quaternion { double x_, y_, z_, w_; Quaternion(double x = 0, double y = 0, double z = 0, double w = 1) : x_(x), y_(y), z_(z), w_(w) {} };
struct Vector3 { double x_, y_, z_; Vector3(double x = 0, double y = 0, double z = 0) : x_(x), y_(y), z_(z) {} };
struct StampedTransform {
  ros::Time stamp_; std::string frame_id_, child_frame_id_;
  Quaternion q_; Vector3 o_;
  void setRotation(const Quaternion &q) { q_ = q; }
  void setOrigin(const Vector3 &o) { o_ = o; }
  void setIdentity() { q_ = Quaternion(); o_ = Vector3(); }
};
struct TransformBroadcaster { void sendTransform(const StampedTransform &) {} };
}  // namespace tf
namespace ros {
struct PublishedLog { static sensor_msgs::PointCloud2 &last_cloud() { static sensor_msgs::PointCloud2 m; return m; } static nav_msgs::Odometry &last_odom() { static nav_msgs::Odometry m; return m; } };
template <typename M> inline void Publisher::publish(const M &) const {}
template <> inline void Publisher::publish<sensor_msgs::PointCloud2>(const sensor_msgs::PointCloud2 &m) const { PublishedLog::last_cloud() = m; }
template <> inline void Publisher::publish<nav_msgs::Odometry>(const nav_msgs::Odometry &m) const { PublishedLog::last_odom() = m; }
}  // namespace ros
namespace pcl {
struct PointXYZ { PCL_ADD_POINT4D; PointXYZ() { x = y = z = 0.f; data[3] = 1.f; } PointXYZ(float a, float b, float c) { x = a; y = b; z = c; data[3] = 1.f; } };
struct Normal { float normal_x, normal_y, normal_z, curvature; Normal(float a = 0, float b = 0, float c = 0) : normal_x(a), normal_y(b), normal_z(c), curvature(0) {} };
template <typename A, typename B> void copyPointCloud(const PointCloud<A> &in, PointCloud<B> &out) {
  out.clear();
  for (size_t i = 0; i < in.size(); ++i) { B p; p.x = in[i].x; p.y = in[i].y; p.z = in[i].z; out.push_back(p); }
}
template <typename PointT> void shim_set_intensity(PointT &, float) {}
inline void shim_set_intensity(PointXYZI &p, float v) { p.intensity = v; }
template <typename PointT> float shim_get_intensity(const PointT &) { return 0.f; }
inline float shim_get_intensity(const PointXYZI &p) { return p.intensity; }
template <typename PointT> void fromROSMsg(const sensor_msgs::PointCloud2 &m, PointCloud<PointT> &c) {
  c.clear();
  for (size_t i = 0; i + 3 < m.xyzi.size(); i += 4) { PointT p; p.x = m.xyzi[i]; p.y = m.xyzi[i + 1]; p.z = m.xyzi[i + 2]; shim_set_intensity(p, m.xyzi[i + 3]); c.push_back(p); }
}
template <typename PointT> void toROSMsg(const PointCloud<PointT> &c, sensor_msgs::PointCloud2 &m) {
  m.xyzi.clear();
  for (size_t i = 0; i < c.size(); ++i) { m.xyzi.push_back(c[i].x); m.xyzi.push_back(c[i].y); m.xyzi.push_back(c[i].z); m.xyzi.push_back(shim_get_intensity(c[i])); }
}
template <typename PointT> void removeNaNFromPointCloud(const PointCloud<PointT> &in, PointCloud<PointT> &out, std::vector<int> &idx) {
  const std::vector<PointT> src(in.points);   // (the reference filters clouds in place: in and out may be the same object)
  out.clear(); idx.clear();
  for (size_t i = 0; i < src.size(); ++i)
    if (std::isfinite(src[i].x) && std::isfinite(src[i].y) && std::isfinite(src[i].z)) { out.push_back(src[i]); idx.push_back(int(i)); }
}
}  // namespace pcl
namespace lio {
// as in the reference's utils/common_ros.h: convert, stamp, publish (the stand-in publisher keeps the last cloud message)
template <typename PointT>
inline void PublishCloudMsg(ros::Publisher &publisher, const pcl::PointCloud<PointT> &cloud, const ros::Time &stamp, std::string frame_id) {
  sensor_msgs::PointCloud2 out;
  out.header.frame_id = frame_id;
  out.header.stamp = stamp;
  pcl::toROSMsg(cloud, out);      // (the stand-in conversion fills the xyzi payload only, the header set above stays)
  publisher.publish(out);
}
}  // namespace lio
