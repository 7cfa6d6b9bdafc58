// Sophus::SO3 stand-in (the vendored Sophus needs Eigen's internals): a unit quaternion with hat / exp / log / matrix, enough for
// utils/geometry_utils.h and the one SO3 the reference's PointOdometry constructs.  oracle/ref_shim: test infrastructure.
#pragma once
#include <cmath>

#include "../../Eigen/Eigen"
namespace Sophus {
template <typename T> struct Constants {
  static T epsilon() { return T(1e-10); }
  static T pi() { return T(3.141592653589793238462643383279502884); }
};
template <> struct Constants<float> {
  static float epsilon() { return 1e-5f; }
  static float pi() { return 3.141592653589793238462643383279502884f; }
};
template <typename T>
class SO3 {
  Eigen::Quaternion<T> q_;

 public:
  typedef Eigen::Matrix<T, 3, 1> Tangent;
  typedef Eigen::Matrix<T, 3, 3> Transformation;
  SO3() {}
  template <typename D> explicit SO3(const Eigen::QuaternionBase<D> &q) : q_(q) { q_.normalize(); }   // (Sophus normalises what it is given)
  const Eigen::Quaternion<T> &unit_quaternion() const { return q_; }
  Transformation matrix() const { return q_.toRotationMatrix(); }
  static Transformation hat(const Tangent &v) {
    Transformation m;
    m << T(0), -v.z(), v.y(), v.z(), T(0), -v.x(), -v.y(), v.x(), T(0);
    return m;
  }
  static SO3 exp(const Tangent &omega) {
    const T theta_sq = omega.squaredNorm(), theta = std::sqrt(theta_sq), half = T(0.5) * theta;
    T imag, real;
    if (theta < Constants<T>::epsilon()) { const T t4 = theta_sq * theta_sq; imag = T(0.5) - theta_sq / T(48) + t4 / T(3840); real = T(1) - theta_sq / T(8) + t4 / T(384); }
    else { imag = std::sin(half) / theta; real = std::cos(half); }
    SO3 r;
    r.q_ = Eigen::Quaternion<T>(real, imag * omega.x(), imag * omega.y(), imag * omega.z());
    return r;
  }
  Tangent log() const {
    const T sq = q_.vec().squaredNorm(), w = q_.w();
    T two_atan_nbyw_by_n;
    if (sq < Constants<T>::epsilon() * Constants<T>::epsilon()) two_atan_nbyw_by_n = T(2) / w - T(2) * sq / (T(3) * w * w * w);
    else {
      const T n = std::sqrt(sq);
      if (std::fabs(w) < Constants<T>::epsilon()) two_atan_nbyw_by_n = (w > T(0) ? Constants<T>::pi() : -Constants<T>::pi()) / n;
      else two_atan_nbyw_by_n = T(2) * std::atan(n / w) / n;
    }
    return two_atan_nbyw_by_n * q_.vec();
  }
};
typedef SO3<float> SO3f;
typedef SO3<double> SO3d;
}  // namespace Sophus
