// The two Ceres interfaces the reference's factor classes derive from (ceres/sized_cost_function.h, ceres/local_parameterization.h):
// just enough of the class shape for the reference sources to compile where they lie.  Test infrastructure (oracle/ref_shim).
#pragma once
#include <algorithm>
#include <cmath>
#include <limits>
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
  std::vector<int> *mutable_parameter_block_sizes() { return &sizes_; }
  void set_num_residuals(int n) { nres_ = n; }
  std::vector<int> sizes_;
  int nres_ = 0;
};
template <int kNumResiduals, int... Ns>
class SizedCostFunction : public CostFunction {
 public:
  SizedCostFunction() { nres_ = kNumResiduals; sizes_ = std::vector<int>{Ns...}; }
};
class LossFunction {
 public:
  virtual ~LossFunction() {}
  virtual void Evaluate(double sq_norm, double out[3]) const = 0;
};
// rho(s) = b log(1 + s / b), b = a^2 (ceres/loss_function.h)
class CauchyLoss : public LossFunction {
  const double b_, c_;

 public:
  explicit CauchyLoss(double a) : b_(a * a), c_(1.0 / b_) {}
  void Evaluate(double s, double rho[3]) const override {
    const double sum = 1.0 + s * c_, inv = 1.0 / sum;
    rho[0] = b_ * std::log(sum);
    rho[1] = std::max(std::numeric_limits<double>::min(), inv);
    rho[2] = -c_ * (inv * inv);
  }
};
class HuberLoss : public LossFunction {
  const double a_, b_;

 public:
  explicit HuberLoss(double a) : a_(a), b_(a * a) {}
  void Evaluate(double s, double rho[3]) const override {
    if (s > b_) { const double r = std::sqrt(s); rho[0] = 2.0 * a_ * r - b_; rho[1] = std::max(std::numeric_limits<double>::min(), a_ / r); rho[2] = -rho[1] / (2.0 * s); }
    else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
  }
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

#include "problem.h"   // ceres::Problem / ceres::Solve stand-ins (used by the reference's Estimator.cc only)
