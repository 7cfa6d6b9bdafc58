// oracle/liomath.h — TEST INFRASTRUCTURE ONLY (CPU oracle).  Not part of the product.
//
// Dependency-free restatement of the handful of Eigen 3.3 / utils semantics the reference's hot
// path relies on.  Eigen itself is absent from /root/reference and
// from this image, so every routine states the documented algorithm; parity with Eigen's rounding
// is UNPINNED (cross-checked against numpy/scipy in tests/).  Reference call sites:
//   include/utils/math_utils.h:39-234, include/utils/geometry_utils.h:288-317,
//   include/utils/Twist.h:39-97.
// Build with -ffp-contract=off (the reference is Release without -march: no FMA contraction).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace orc {

template <typename T>
struct V3 {
  T x{}, y{}, z{};
  V3() = default;
  V3(T a, T b, T c) : x(a), y(b), z(c) {}
  T &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
  const T &operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
  V3 operator+(const V3 &o) const { return {x + o.x, y + o.y, z + o.z}; }
  V3 operator-(const V3 &o) const { return {x - o.x, y - o.y, z - o.z}; }
  V3 operator-() const { return {-x, -y, -z}; }
  V3 operator*(T s) const { return {x * s, y * s, z * s}; }
  V3 operator/(T s) const { return {x / s, y / s, z / s}; }
  V3 &operator+=(const V3 &o) { x += o.x; y += o.y; z += o.z; return *this; }
  T dot(const V3 &o) const { return x * o.x + y * o.y + z * o.z; }
  V3 cross(const V3 &o) const { return {y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x}; }
  T squaredNorm() const { return x * x + y * y + z * z; }
  T norm() const { return std::sqrt(squaredNorm()); }
  V3 normalized() const { T n2 = squaredNorm(); return n2 > T(0) ? *this / std::sqrt(n2) : *this; }
  template <typename U> V3<U> cast() const { return {U(x), U(y), U(z)}; }
};
template <typename T> inline V3<T> operator*(T s, const V3<T> &v) { return v * s; }

template <typename T>
struct M3 {
  T m[3][3]{};
  static M3 Identity() { M3 r; r.m[0][0] = r.m[1][1] = r.m[2][2] = T(1); return r; }
  static M3 Zero() { return M3(); }
  T &operator()(int i, int j) { return m[i][j]; }
  const T &operator()(int i, int j) const { return m[i][j]; }
  M3 operator*(const M3 &o) const {
    M3 r;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        T s = m[i][0] * o.m[0][j];
        s += m[i][1] * o.m[1][j];
        s += m[i][2] * o.m[2][j];
        r.m[i][j] = s;
      }
    return r;
  }
  V3<T> operator*(const V3<T> &v) const {
    return {m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z, m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z,
            m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z};
  }
  M3 operator*(T s) const { M3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = m[i][j] * s; return r; }
  M3 operator+(const M3 &o) const { M3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = m[i][j] + o.m[i][j]; return r; }
  M3 operator-(const M3 &o) const { M3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = m[i][j] - o.m[i][j]; return r; }
  M3 operator-() const { M3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = -m[i][j]; return r; }
  M3 transpose() const { M3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = m[j][i]; return r; }
  T trace() const { return m[0][0] + m[1][1] + m[2][2]; }
  // general 3x3 inverse via cofactors (Eigen's fixed-size 3x3 inverse is cofactor based)
  M3 inverse() const {
    M3 c;
    c.m[0][0] = m[1][1] * m[2][2] - m[1][2] * m[2][1];
    c.m[0][1] = m[0][2] * m[2][1] - m[0][1] * m[2][2];
    c.m[0][2] = m[0][1] * m[1][2] - m[0][2] * m[1][1];
    c.m[1][0] = m[1][2] * m[2][0] - m[1][0] * m[2][2];
    c.m[1][1] = m[0][0] * m[2][2] - m[0][2] * m[2][0];
    c.m[1][2] = m[0][2] * m[1][0] - m[0][0] * m[1][2];
    c.m[2][0] = m[1][0] * m[2][1] - m[1][1] * m[2][0];
    c.m[2][1] = m[0][1] * m[2][0] - m[0][0] * m[2][1];
    c.m[2][2] = m[0][0] * m[1][1] - m[0][1] * m[1][0];
    T det = m[0][0] * c.m[0][0] + m[0][1] * c.m[1][0] + m[0][2] * c.m[2][0];
    T inv = T(1) / det;
    return c * inv;
  }
  template <typename U> M3<U> cast() const { M3<U> r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = U(m[i][j]); return r; }
};

// math_utils.h:134-141 SkewSymmetric
template <typename T>
inline M3<T> Skew(const V3<T> &v) {
  M3<T> s;
  s.m[0][1] = -v.z; s.m[0][2] = v.y;
  s.m[1][0] = v.z;  s.m[1][2] = -v.x;
  s.m[2][0] = -v.y; s.m[2][1] = v.x;
  return s;
}

// Eigen::Quaternion semantics (coeff order x,y,z,w)
template <typename T>
struct Q {
  T x{}, y{}, z{}, w{T(1)};
  Q() = default;
  Q(T w_, T x_, T y_, T z_) : x(x_), y(y_), z(z_), w(w_) {}  // Eigen ctor order (w,x,y,z)
  static Q Identity() { return Q(); }
  V3<T> vec() const { return {x, y, z}; }
  T squaredNorm() const { return x * x + y * y + z * z + w * w; }
  T norm() const { return std::sqrt(squaredNorm()); }
  Q conjugate() const { return Q(w, -x, -y, -z); }
  Q normalized() const {
    T n2 = squaredNorm();
    if (n2 > T(0)) { T n = std::sqrt(n2); return Q(w / n, x / n, y / n, z / n); }
    return *this;
  }
  void normalize() { *this = normalized(); }
  Q inverse() const {
    T n2 = squaredNorm();
    if (n2 > T(0)) return Q(w / n2, -x / n2, -y / n2, -z / n2);
    return Q(T(0), T(0), T(0), T(0));
  }
  // Eigen quat_product (generic path)
  Q operator*(const Q &b) const {
    return Q(w * b.w - x * b.x - y * b.y - z * b.z, w * b.x + x * b.w + y * b.z - z * b.y,
             w * b.y + y * b.w + z * b.x - x * b.z, w * b.z + z * b.w + x * b.y - y * b.x);
  }
  // Eigen _transformVector: v + w*2(u x v) + u x 2(u x v)
  V3<T> operator*(const V3<T> &v) const {
    V3<T> u = vec();
    V3<T> uv = u.cross(v);
    uv += uv;
    V3<T> wuv = uv * w;
    V3<T> r = v + wuv;
    return r + u.cross(uv);
  }
  // Eigen toRotationMatrix (assumes unit norm; reproduced verbatim for non-unit inputs too)
  M3<T> toRotationMatrix() const {
    M3<T> r;
    const T tx = T(2) * x, ty = T(2) * y, tz = T(2) * z;
    const T twx = tx * w, twy = ty * w, twz = tz * w;
    const T txx = tx * x, txy = ty * x, txz = tz * x;
    const T tyy = ty * y, tyz = tz * y, tzz = tz * z;
    r.m[0][0] = T(1) - (tyy + tzz); r.m[0][1] = txy - twz;          r.m[0][2] = txz + twy;
    r.m[1][0] = txy + twz;          r.m[1][1] = T(1) - (txx + tzz); r.m[1][2] = tyz - twx;
    r.m[2][0] = txz - twy;          r.m[2][1] = tyz + twx;          r.m[2][2] = T(1) - (txx + tyy);
    return r;
  }
  // Eigen Quaternion(Matrix3) — Shepperd branches
  static Q FromMatrix(const M3<T> &mat) {
    Q q;
    T t = mat.trace();
    if (t > T(0)) {
      t = std::sqrt(t + T(1));
      q.w = T(0.5) * t;
      t = T(0.5) / t;
      q.x = (mat(2, 1) - mat(1, 2)) * t;
      q.y = (mat(0, 2) - mat(2, 0)) * t;
      q.z = (mat(1, 0) - mat(0, 1)) * t;
    } else {
      int i = 0;
      if (mat(1, 1) > mat(0, 0)) i = 1;
      if (mat(2, 2) > mat(i, i)) i = 2;
      int j = (i + 1) % 3, k = (j + 1) % 3;
      t = std::sqrt(mat(i, i) - mat(j, j) - mat(k, k) + T(1));
      T c[3];
      c[i] = T(0.5) * t;
      t = T(0.5) / t;
      q.w = (mat(k, j) - mat(j, k)) * t;
      c[j] = (mat(j, i) + mat(i, j)) * t;
      c[k] = (mat(k, i) + mat(i, k)) * t;
      q.x = c[0]; q.y = c[1]; q.z = c[2];
    }
    return q;
  }
  T dot(const Q &o) const { return x * o.x + y * o.y + z * o.z + w * o.w; }
  // Eigen slerp (B.4)
  Q slerp(T t, const Q &other) const {
    const T one = T(1) - std::numeric_limits<T>::epsilon();
    T d = dot(other);
    T absD = std::fabs(d);
    T s0, s1;
    if (absD >= one) { s0 = T(1) - t; s1 = t; }
    else {
      T theta = std::acos(absD);
      T sinTheta = std::sin(theta);
      s0 = std::sin((T(1) - t) * theta) / sinTheta;
      s1 = std::sin((t * theta)) / sinTheta;
    }
    if (d < T(0)) s1 = -s1;
    return Q(s0 * w + s1 * other.w, s0 * x + s1 * other.x, s0 * y + s1 * other.y, s0 * z + s1 * other.z);
  }
  T angularDistance(const Q &other) const {
    Q d = (*this) * other.conjugate();
    return T(2) * std::atan2(d.vec().norm(), std::fabs(d.w));
  }
  template <typename U> Q<U> cast() const { return Q<U>(U(w), U(x), U(y), U(z)); }
};

// math_utils.h:116-128 DeltaQ — UNNORMALISED [1, theta/2] (A.9)
template <typename T>
inline Q<T> DeltaQ(const V3<T> &theta) {
  V3<T> h = theta / T(2);
  return Q<T>(T(1), h.x, h.y, h.z);
}

// 4x4 left/right quaternion matrices (math_utils.h:143-165); only the top-left 3x3 is ever used.
template <typename T> inline M3<T> LeftQuatTL3(const Q<T> &q) { return M3<T>::Identity() * q.w + Skew(q.vec()); }
template <typename T> inline M3<T> RightQuatTL3(const Q<T> &q) { return M3<T>::Identity() * q.w - Skew(q.vec()); }
// full 4x4 (row-major [4][4]) for products Left*Right
template <typename T> inline void LeftQuat4(const Q<T> &q, T out[4][4]) {
  M3<T> a = LeftQuatTL3(q);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) out[i][j] = a(i, j);
  V3<T> v = q.vec();
  for (int j = 0; j < 3; ++j) { out[3][j] = -v[j]; out[j][3] = v[j]; }
  out[3][3] = q.w;
}
template <typename T> inline void RightQuat4(const Q<T> &q, T out[4][4]) {
  M3<T> a = RightQuatTL3(q);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) out[i][j] = a(i, j);
  V3<T> v = q.vec();
  for (int j = 0; j < 3; ++j) { out[3][j] = -v[j]; out[j][3] = v[j]; }
  out[3][3] = q.w;
}

// math_utils.h:186-203 R2ypr (degrees) / :205-232 ypr2R
inline V3<double> R2ypr(const M3<double> &R) {
  V3<double> n{R(0, 0), R(1, 0), R(2, 0)}, o{R(0, 1), R(1, 1), R(2, 1)}, a{R(0, 2), R(1, 2), R(2, 2)};
  double y = std::atan2(n.y, n.x);
  double p = std::atan2(-n.z, n.x * std::cos(y) + n.y * std::sin(y));
  double r = std::atan2(a.x * std::sin(y) - a.y * std::cos(y), -o.x * std::sin(y) + o.y * std::cos(y));
  return V3<double>{y, p, r} / M_PI * 180.0;
}
inline M3<double> ypr2R(const V3<double> &ypr) {
  double y = ypr.x / 180.0 * M_PI, p = ypr.y / 180.0 * M_PI, r = ypr.z / 180.0 * M_PI;
  M3<double> Rz, Ry, Rx;
  Rz(0, 0) = std::cos(y); Rz(0, 1) = -std::sin(y); Rz(1, 0) = std::sin(y); Rz(1, 1) = std::cos(y); Rz(2, 2) = 1;
  Ry(0, 0) = std::cos(p); Ry(0, 2) = std::sin(p); Ry(1, 1) = 1; Ry(2, 0) = -std::sin(p); Ry(2, 2) = std::cos(p);
  Rx(0, 0) = 1; Rx(1, 1) = std::cos(r); Rx(1, 2) = -std::sin(r); Rx(2, 1) = std::sin(r); Rx(2, 2) = std::cos(r);
  return Rz * Ry * Rx;
}

// math_utils.h:39-64
template <typename T> inline T RadToDeg(T rad) { return T(rad * 180.0 / M_PI); }
template <typename T> inline T NormalizeRad(T rad) {
  rad = T(std::fmod(rad + M_PI, 2 * M_PI));
  if (rad < 0) rad = T(rad + 2 * M_PI);
  return T(rad - M_PI);
}
template <typename T> inline T NormalizeDeg(T deg) {
  deg = T(std::fmod(deg + 180.0, 360.0));
  if (deg < 0) deg = T(deg + 360.0);
  return T(deg - 180.0);
}

// Twist<T> (include/utils/Twist.h:39-97).  transform() = (rot.normalized().toRotationMatrix(), pos)
template <typename T>
struct Twist {
  Q<T> rot;
  V3<T> pos;
  Twist() = default;
  Twist(const Q<T> &r, const V3<T> &p) : rot(r), pos(p) {}
  // Twist(Affine): rot = Quaternion(linear).normalized()
  static Twist FromAffine(const M3<T> &lin, const V3<T> &tr) { return Twist(Q<T>::FromMatrix(lin).normalized(), tr); }
  M3<T> linear() const { return rot.normalized().toRotationMatrix(); }
  // inverse(): Affine inverse (general 3x3 inverse), rot = Quaternion(linear) NOT normalised (Twist.h:71-77)
  Twist inverse() const {
    M3<T> li = linear().inverse();
    Twist r;
    r.rot = Q<T>::FromMatrix(li);
    r.pos = -(li * pos);
    return r;
  }
  Twist operator*(const Twist &o) const {
    M3<T> a = linear(), b = o.linear();
    return FromAffine(a * b, a * o.pos + pos);
  }
  template <typename U> Twist<U> cast() const { return Twist<U>(rot.template cast<U>(), pos.template cast<U>()); }
};

// ------------------------------------------------------------------------------------------------
// Small dense linear algebra (double, row-major dynamic).  Documented algorithms, not Eigen's code.
// ------------------------------------------------------------------------------------------------
struct Mat {
  int r = 0, c = 0;
  std::vector<double> a;
  Mat() = default;
  Mat(int r_, int c_) : r(r_), c(c_), a(size_t(r_) * c_, 0.0) {}
  double &operator()(int i, int j) { return a[size_t(i) * c + j]; }
  double operator()(int i, int j) const { return a[size_t(i) * c + j]; }
  void setZero() { std::fill(a.begin(), a.end(), 0.0); }
  static Mat Identity(int n) { Mat m(n, n); for (int i = 0; i < n; ++i) m(i, i) = 1.0; return m; }
  Mat transpose() const { Mat t(c, r); for (int i = 0; i < r; ++i) for (int j = 0; j < c; ++j) t(j, i) = (*this)(i, j); return t; }
};
inline Mat matmul(const Mat &A, const Mat &B) {
  Mat C(A.r, B.c);
  for (int i = 0; i < A.r; ++i)
    for (int k = 0; k < A.c; ++k) {
      double aik = A(i, k);
      if (aik == 0.0) continue;
      for (int j = 0; j < B.c; ++j) C(i, j) += aik * B(k, j);
    }
  return C;
}
inline std::vector<double> matvec(const Mat &A, const std::vector<double> &x) {
  std::vector<double> y(A.r, 0.0);
  for (int i = 0; i < A.r; ++i) { double s = 0; for (int j = 0; j < A.c; ++j) s += A(i, j) * x[j]; y[i] = s; }
  return y;
}

// Cholesky A = L L^T (lower).  Returns false when a pivot is <= 0 (Eigen LLT reports NumericalIssue).
inline bool cholesky(const Mat &A, Mat &L) {
  int n = A.r;
  L = Mat(n, n);
  for (int j = 0; j < n; ++j) {
    double d = A(j, j);
    for (int k = 0; k < j; ++k) d -= L(j, k) * L(j, k);
    if (!(d > 0.0)) return false;
    double ljj = std::sqrt(d);
    L(j, j) = ljj;
    for (int i = j + 1; i < n; ++i) {
      double s = A(i, j);
      for (int k = 0; k < j; ++k) s -= L(i, k) * L(j, k);
      L(i, j) = s / ljj;
    }
  }
  return true;
}
inline void chol_solve(const Mat &L, std::vector<double> &b) {
  int n = L.r;
  for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L(i, k) * b[k]; b[i] = s / L(i, i); }
  for (int i = n - 1; i >= 0; --i) { double s = b[i]; for (int k = i + 1; k < n; ++k) s -= L(k, i) * b[k]; b[i] = s / L(i, i); }
}
// General inverse by Gauss-Jordan with partial pivoting (Eigen: PartialPivLU inverse for n > 4).
inline bool inverse(const Mat &A, Mat &Ainv) {
  int n = A.r;
  Mat M = A;
  Ainv = Mat::Identity(n);
  for (int col = 0; col < n; ++col) {
    int piv = col; double best = std::fabs(M(col, col));
    for (int i = col + 1; i < n; ++i) if (std::fabs(M(i, col)) > best) { best = std::fabs(M(i, col)); piv = i; }
    if (best == 0.0) return false;
    if (piv != col) for (int j = 0; j < n; ++j) { std::swap(M(piv, j), M(col, j)); std::swap(Ainv(piv, j), Ainv(col, j)); }
    double d = M(col, col);
    for (int j = 0; j < n; ++j) { M(col, j) /= d; Ainv(col, j) /= d; }
    for (int i = 0; i < n; ++i) if (i != col) {
      double f = M(i, col);
      if (f == 0.0) continue;
      for (int j = 0; j < n; ++j) { M(i, j) -= f * M(col, j); Ainv(i, j) -= f * Ainv(col, j); }
    }
  }
  return true;
}
// Symmetric eigendecomposition by cyclic Jacobi: eigenvalues ascending, eigenvectors as COLUMNS of V
// (Eigen::SelfAdjointEigenSolver contract, B.4).  T = float or double storage in row-major arrays.
template <typename T>
inline void sym_eigen(int n, const T *A_in, T *evals, T *V) {
  std::vector<double> A(size_t(n) * n), U(size_t(n) * n, 0.0);
  for (int i = 0; i < n * n; ++i) A[i] = double(A_in[i]);
  for (int i = 0; i < n; ++i) U[size_t(i) * n + i] = 1.0;
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0;
    for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) off += A[size_t(i) * n + j] * A[size_t(i) * n + j];
    if (off < 1e-300) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        double apq = A[size_t(p) * n + q];
        if (apq == 0.0) continue;
        double app = A[size_t(p) * n + p], aqq = A[size_t(q) * n + q];
        double tau = (aqq - app) / (2.0 * apq);
        double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
        double c = 1.0 / std::sqrt(1.0 + t * t), s = t * c;
        for (int k = 0; k < n; ++k) {
          double akp = A[size_t(k) * n + p], akq = A[size_t(k) * n + q];
          A[size_t(k) * n + p] = c * akp - s * akq;
          A[size_t(k) * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {
          double apk = A[size_t(p) * n + k], aqk = A[size_t(q) * n + k];
          A[size_t(p) * n + k] = c * apk - s * aqk;
          A[size_t(q) * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          double ukp = U[size_t(k) * n + p], ukq = U[size_t(k) * n + q];
          U[size_t(k) * n + p] = c * ukp - s * ukq;
          U[size_t(k) * n + q] = s * ukp + c * ukq;
        }
      }
  }
  std::vector<int> order(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](int a, int b) { return A[size_t(a) * n + a] < A[size_t(b) * n + b]; });
  for (int k = 0; k < n; ++k) {
    evals[k] = T(A[size_t(order[k]) * n + order[k]]);
    for (int i = 0; i < n; ++i) V[size_t(i) * n + k] = T(U[size_t(i) * n + order[k]]);
  }
}

// Least squares / linear solve by Householder QR with column pivoting in scalar type T
// (Eigen::ColPivHouseholderQR::solve contract: unique LS solution for full column rank, B.4).
// A is m x n row-major (destroyed), b has m entries (destroyed), x gets n entries.
template <typename T>
inline void colpiv_qr_solve(int m, int n, T *A, T *b, T *x) {
  int perm[16];
  T colnorm[16];
  for (int j = 0; j < n; ++j) {
    perm[j] = j;
    T s = 0;
    for (int i = 0; i < m; ++i) s += A[i * n + j] * A[i * n + j];
    colnorm[j] = s;
  }
  int rank = 0;
  T maxnorm0 = 0;
  for (int j = 0; j < n; ++j) maxnorm0 = std::max(maxnorm0, colnorm[j]);
  // Eigen 3.3 ColPivHouseholderQR::computeInPlace: threshold_helper = (max column norm * epsilon)^2 / rows; the count of
  // meaningful pivots stops at the first k whose largest remaining squared column norm is < threshold_helper * (rows - k),
  // and solve() back-substitutes over those pivots only (recollection of the Eigen source, second-sourced in
  // tests/golden/second_source.py; the round-1 restatement had eps^2 * max * rows, 5-6x larger for the 5x3 / 6x6 systems)
  const T thresh_helper = std::numeric_limits<T>::epsilon() * std::numeric_limits<T>::epsilon() * maxnorm0 / T(m);
  int steps = std::min(m, n);
  for (int k = 0; k < steps; ++k) {
    // pivot: recompute remaining column norms exactly (small sizes; avoids downdating drift)
    int piv = k; T best = -1;
    for (int j = k; j < n; ++j) {
      T s = 0;
      for (int i = k; i < m; ++i) s += A[i * n + j] * A[i * n + j];
      colnorm[j] = s;
      if (s > best) { best = s; piv = j; }
    }
    if (best < thresh_helper * T(m - k)) break;
    if (piv != k) {
      for (int i = 0; i < m; ++i) std::swap(A[i * n + piv], A[i * n + k]);
      std::swap(perm[piv], perm[k]);
    }
    // Householder on column k, rows k..m-1
    T alpha = A[k * n + k];
    T sigma = 0;
    for (int i = k + 1; i < m; ++i) sigma += A[i * n + k] * A[i * n + k];
    T normx = std::sqrt(alpha * alpha + sigma);
    if (normx == T(0)) break;
    T beta = (alpha >= T(0)) ? -normx : normx;
    T v0 = alpha - beta;
    // v = [v0, A[k+1..,k]]; H = I - 2 v v^T / (v^T v)
    T vtv = v0 * v0 + sigma;
    if (vtv > T(0)) {
      for (int j = k + 1; j < n; ++j) {
        T s = v0 * A[k * n + j];
        for (int i = k + 1; i < m; ++i) s += A[i * n + k] * A[i * n + j];
        T f = T(2) * s / vtv;
        A[k * n + j] -= f * v0;
        for (int i = k + 1; i < m; ++i) A[i * n + j] -= f * A[i * n + k];
      }
      T s = v0 * b[k];
      for (int i = k + 1; i < m; ++i) s += A[i * n + k] * b[i];
      T f = T(2) * s / vtv;
      b[k] -= f * v0;
      for (int i = k + 1; i < m; ++i) b[i] -= f * A[i * n + k];
    }
    A[k * n + k] = beta;
    for (int i = k + 1; i < m; ++i) A[i * n + k] = T(0);
    ++rank;
  }
  T y[16];
  for (int j = 0; j < n; ++j) y[j] = T(0);
  for (int i = rank - 1; i >= 0; --i) {
    T s = b[i];
    for (int j = i + 1; j < rank; ++j) s -= A[i * n + j] * y[j];
    y[i] = s / A[i * n + i];
  }
  for (int j = 0; j < n; ++j) x[perm[j]] = y[j];
}

}  // namespace orc
