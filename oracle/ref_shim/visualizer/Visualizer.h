// Stand-in for the reference's visualizer/Visualizer.h (RViz markers and a PCL viewer: out of scope) — found ahead
// of the reference's own header on the include path, so that Estimator.h sees classes of the same names that do nothing.
// oracle/ref_shim: test infrastructure.
#pragma once
#include <string>
#include <vector>

#include "point_processor/PointMapping.h"

namespace lio {
class Visualizer {
 public:
  Visualizer(std::string = "visualizer", std::vector<double> = {0.0, 1.0, 0.0}, std::vector<double> = {1.0, 1.0, 0.0}) {}
  void UpdateMarkers(std::vector<Transform>, std::vector<Transform>) {}
  void UpdateVelocity(double) {}
  void PublishMarkers() {}
};
class PlaneNormalVisualizer {
 public:
  void Spin() {}
  void UpdateCloud(pcl::PointCloud<pcl::PointXYZ>::ConstPtr, std::string = "cloud", std::vector<double> = {1.0, 0.0, 1.0}) {}
  void UpdateCloudAndNormals(pcl::PointCloud<pcl::PointXYZ>::ConstPtr, pcl::PointCloud<pcl::Normal>::ConstPtr, int = 10, std::string = "cloud",
                             std::string = "normals", std::vector<double> = {1.0, 1.0, 1.0}, std::vector<double> = {1.0, 1.0, 0.0}) {}
  bool init = false;
  bool first = false;
};
}  // namespace lio
