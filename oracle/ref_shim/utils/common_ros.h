// Stand-in for the reference's utils/common_ros.h (ROS + PCL headers): the factor sources only need its logging macros to exist.
#pragma once
#include <iostream>
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
