// The two Ceres interfaces the reference's factor classes derive from (ceres/sized_cost_function.h, ceres/local_parameterization.h):
// just enough of the class shape for the reference sources to compile where they lie.  Test infrastructure (oracle/ref_shim).
#pragma once
#include <memory>
#include <vector>

#include "../utils/common_ros.h"   // ROS_DEBUG / DLOG: the real headers reach the factor sources through ROS includes
namespace ceres {
class CostFunction {
 public:
  virtual ~CostFunction() {}
  virtual bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const = 0;
  const std::vector<int> &parameter_block_sizes() const { return sizes_; }
  int num_residuals() const { return nres_; }

 protected:
  std::vector<int> sizes_;
  int nres_ = 0;
};
template <int kNumResiduals, int... Ns>
class SizedCostFunction : public CostFunction {
 public:
  SizedCostFunction() { nres_ = kNumResiduals; sizes_ = std::vector<int>{Ns...}; }
};
class LocalParameterization {
 public:
  virtual ~LocalParameterization() {}
  virtual bool Plus(const double *x, const double *delta, double *x_plus_delta) const = 0;
  virtual bool ComputeJacobian(const double *x, double *jacobian) const = 0;
  virtual int GlobalSize() const = 0;
  virtual int LocalSize() const = 0;
};
}  // namespace ceres
