// hmath.h — small fixed-size math shared by the host side and the HIP kernels of the product.
//
// Semantics follow the reference's use of Eigen (quaternion product order, v + 2w(u x v) + 2u x (u x v)
// rotation, Shepperd matrix->quaternion, unnormalised DeltaQ; include/utils/math_utils.h:116-185,
// include/utils/Twist.h:39-97) so that fp32 feature extraction reproduces the same operation order
// as a CPU run of the reference built without FMA contraction.  Compile with -ffp-contract=off.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LIO_HD __host__ __device__ inline
#else
#define LIO_HD inline
#endif

namespace lio {

template <typename T>
struct Vec3 {
  T x, y, z;
  LIO_HD Vec3() : x(0), y(0), z(0) {}
  LIO_HD Vec3(T a, T b, T c) : x(a), y(b), z(c) {}
  LIO_HD T operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
  LIO_HD T &at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
template <typename T> LIO_HD Vec3<T> operator+(const Vec3<T> &a, const Vec3<T> &b) { return Vec3<T>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <typename T> LIO_HD Vec3<T> operator-(const Vec3<T> &a, const Vec3<T> &b) { return Vec3<T>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <typename T> LIO_HD Vec3<T> operator-(const Vec3<T> &a) { return Vec3<T>(-a.x, -a.y, -a.z); }
template <typename T> LIO_HD Vec3<T> operator*(const Vec3<T> &a, T s) { return Vec3<T>(a.x * s, a.y * s, a.z * s); }
template <typename T> LIO_HD Vec3<T> operator*(T s, const Vec3<T> &a) { return Vec3<T>(a.x * s, a.y * s, a.z * s); }
template <typename T> LIO_HD Vec3<T> operator/(const Vec3<T> &a, T s) { return Vec3<T>(a.x / s, a.y / s, a.z / s); }
template <typename T> LIO_HD T dot(const Vec3<T> &a, const Vec3<T> &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T> LIO_HD Vec3<T> cross(const Vec3<T> &a, const Vec3<T> &b) {
  return Vec3<T>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
template <typename T> LIO_HD T norm(const Vec3<T> &a) { return sqrt(dot(a, a)); }

template <typename T>
struct Mat3 {
  T m[9];  // row-major
  LIO_HD Mat3() { for (int i = 0; i < 9; ++i) m[i] = 0; }
  LIO_HD static Mat3 identity() { Mat3 r; r.m[0] = r.m[4] = r.m[8] = 1; return r; }
  LIO_HD T operator()(int i, int j) const { return m[3 * i + j]; }
  LIO_HD T &operator()(int i, int j) { return m[3 * i + j]; }
};
template <typename T> LIO_HD Mat3<T> operator*(const Mat3<T> &a, const Mat3<T> &b) {
  Mat3<T> r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      T s = a(i, 0) * b(0, j);
      s += a(i, 1) * b(1, j);
      s += a(i, 2) * b(2, j);
      r(i, j) = s;
    }
  return r;
}
template <typename T> LIO_HD Vec3<T> operator*(const Mat3<T> &a, const Vec3<T> &v) {
  return Vec3<T>(a(0, 0) * v.x + a(0, 1) * v.y + a(0, 2) * v.z, a(1, 0) * v.x + a(1, 1) * v.y + a(1, 2) * v.z,
                 a(2, 0) * v.x + a(2, 1) * v.y + a(2, 2) * v.z);
}
template <typename T> LIO_HD Mat3<T> operator*(const Mat3<T> &a, T s) { Mat3<T> r; for (int i = 0; i < 9; ++i) r.m[i] = a.m[i] * s; return r; }
template <typename T> LIO_HD Mat3<T> operator+(const Mat3<T> &a, const Mat3<T> &b) { Mat3<T> r; for (int i = 0; i < 9; ++i) r.m[i] = a.m[i] + b.m[i]; return r; }
template <typename T> LIO_HD Mat3<T> operator-(const Mat3<T> &a, const Mat3<T> &b) { Mat3<T> r; for (int i = 0; i < 9; ++i) r.m[i] = a.m[i] - b.m[i]; return r; }
template <typename T> LIO_HD Mat3<T> operator-(const Mat3<T> &a) { Mat3<T> r; for (int i = 0; i < 9; ++i) r.m[i] = -a.m[i]; return r; }
template <typename T> LIO_HD Mat3<T> transpose(const Mat3<T> &a) { Mat3<T> r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r(i, j) = a(j, i); return r; }
template <typename T> LIO_HD T trace(const Mat3<T> &a) { return a(0, 0) + a(1, 1) + a(2, 2); }
template <typename T> LIO_HD Mat3<T> skew(const Vec3<T> &v) {
  Mat3<T> s;
  s(0, 1) = -v.z; s(0, 2) = v.y; s(1, 0) = v.z; s(1, 2) = -v.x; s(2, 0) = -v.y; s(2, 1) = v.x;
  return s;
}
// v^T M as a vector
template <typename T> LIO_HD Vec3<T> rowmul(const Vec3<T> &v, const Mat3<T> &M) {
  return Vec3<T>(v.x * M(0, 0) + v.y * M(1, 0) + v.z * M(2, 0), v.x * M(0, 1) + v.y * M(1, 1) + v.z * M(2, 1),
                 v.x * M(0, 2) + v.y * M(1, 2) + v.z * M(2, 2));
}
// cofactor inverse (Affine::inverse() on a 3x3 linear part)
template <typename T> LIO_HD Mat3<T> inverse(const Mat3<T> &a) {
  Mat3<T> c;
  c(0, 0) = a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1);
  c(0, 1) = a(0, 2) * a(2, 1) - a(0, 1) * a(2, 2);
  c(0, 2) = a(0, 1) * a(1, 2) - a(0, 2) * a(1, 1);
  c(1, 0) = a(1, 2) * a(2, 0) - a(1, 0) * a(2, 2);
  c(1, 1) = a(0, 0) * a(2, 2) - a(0, 2) * a(2, 0);
  c(1, 2) = a(0, 2) * a(1, 0) - a(0, 0) * a(1, 2);
  c(2, 0) = a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0);
  c(2, 1) = a(0, 1) * a(2, 0) - a(0, 0) * a(2, 1);
  c(2, 2) = a(0, 0) * a(1, 1) - a(0, 1) * a(1, 0);
  T det = a(0, 0) * c(0, 0) + a(0, 1) * c(1, 0) + a(0, 2) * c(2, 0);
  return c * (T(1) / det);
}

template <typename T>
struct Quat {
  T x, y, z, w;
  LIO_HD Quat() : x(0), y(0), z(0), w(1) {}
  LIO_HD Quat(T w_, T x_, T y_, T z_) : x(x_), y(y_), z(z_), w(w_) {}
  LIO_HD Vec3<T> vec() const { return Vec3<T>(x, y, z); }
};
template <typename T> LIO_HD Quat<T> operator*(const Quat<T> &a, const Quat<T> &b) {
  return Quat<T>(a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
                 a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x);
}
template <typename T> LIO_HD Quat<T> conj(const Quat<T> &q) { return Quat<T>(q.w, -q.x, -q.y, -q.z); }
template <typename T> LIO_HD T sqnorm(const Quat<T> &q) { return q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w; }
template <typename T> LIO_HD Quat<T> normalized(const Quat<T> &q) {
  T n2 = sqnorm(q);
  if (n2 > T(0)) { T n = sqrt(n2); return Quat<T>(q.w / n, q.x / n, q.y / n, q.z / n); }
  return q;
}
template <typename T> LIO_HD Quat<T> qinverse(const Quat<T> &q) {
  T n2 = sqnorm(q);
  if (n2 > T(0)) return Quat<T>(q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2);
  return Quat<T>(0, 0, 0, 0);
}
// rotate: v + w*(2 u x v) + u x (2 u x v)
template <typename T> LIO_HD Vec3<T> rotate(const Quat<T> &q, const Vec3<T> &v) {
  Vec3<T> u = q.vec();
  Vec3<T> uv = cross(u, v);
  uv = uv + uv;
  Vec3<T> a = v + uv * q.w;
  return a + cross(u, uv);
}
template <typename T> LIO_HD Mat3<T> toRot(const Quat<T> &q) {
  Mat3<T> r;
  const T tx = T(2) * q.x, ty = T(2) * q.y, tz = T(2) * q.z;
  const T twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const T txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const T tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  r(0, 0) = T(1) - (tyy + tzz); r(0, 1) = txy - twz;          r(0, 2) = txz + twy;
  r(1, 0) = txy + twz;          r(1, 1) = T(1) - (txx + tzz); r(1, 2) = tyz - twx;
  r(2, 0) = txz - twy;          r(2, 1) = tyz + twx;          r(2, 2) = T(1) - (txx + tyy);
  return r;
}
template <typename T> LIO_HD Quat<T> fromRot(const Mat3<T> &m) {
  Quat<T> q;
  T t = trace(m);
  if (t > T(0)) {
    t = sqrt(t + T(1));
    q.w = T(0.5) * t;
    t = T(0.5) / t;
    q.x = (m(2, 1) - m(1, 2)) * t;
    q.y = (m(0, 2) - m(2, 0)) * t;
    q.z = (m(1, 0) - m(0, 1)) * t;
  } else {
    int i = 0;
    if (m(1, 1) > m(0, 0)) i = 1;
    if (m(2, 2) > m(i, i)) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(m(i, i) - m(j, j) - m(k, k) + T(1));
    T c[3];
    c[i] = T(0.5) * t;
    t = T(0.5) / t;
    q.w = (m(k, j) - m(j, k)) * t;
    c[j] = (m(j, i) + m(i, j)) * t;
    c[k] = (m(k, i) + m(i, k)) * t;
    q.x = c[0]; q.y = c[1]; q.z = c[2];
  }
  return q;
}
template <typename T> LIO_HD Quat<T> deltaQ(const Vec3<T> &th) { return Quat<T>(T(1), th.x / T(2), th.y / T(2), th.z / T(2)); }

template <typename T> LIO_HD Quat<T> slerp(const Quat<T> &a, T t, const Quat<T> &b, T eps) {
  const T one = T(1) - eps;
  T d = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
  T ad = fabs(d);
  T s0, s1;
  if (ad >= one) { s0 = T(1) - t; s1 = t; }
  else {
    T th = acos(ad);
    T st = sin(th);
    s0 = sin((T(1) - t) * th) / st;
    s1 = sin((t * th)) / st;
  }
  if (d < T(0)) s1 = -s1;
  return Quat<T>(s0 * a.w + s1 * b.w, s0 * a.x + s1 * b.x, s0 * a.y + s1 * b.y, s0 * a.z + s1 * b.z);
}

// Rigid transform as Twist<T> (rotation + translation) with the reference's composition rules.
template <typename T>
struct Rigid {
  Quat<T> rot;
  Vec3<T> pos;
  LIO_HD Rigid() {}
  LIO_HD Rigid(const Quat<T> &r, const Vec3<T> &p) : rot(r), pos(p) {}
};
template <typename T> LIO_HD Mat3<T> linearOf(const Rigid<T> &t) { return toRot(normalized(t.rot)); }
template <typename T> LIO_HD Rigid<T> fromAffine(const Mat3<T> &lin, const Vec3<T> &tr) { return Rigid<T>(normalized(fromRot(lin)), tr); }
template <typename T> LIO_HD Rigid<T> rinverse(const Rigid<T> &t) {
  Mat3<T> li = inverse(linearOf(t));
  return Rigid<T>(fromRot(li), -(li * t.pos));
}
template <typename T> LIO_HD Rigid<T> compose(const Rigid<T> &a, const Rigid<T> &b) {
  Mat3<T> la = linearOf(a), lb = linearOf(b);
  return fromAffine(la * lb, la * b.pos + a.pos);
}

// 5x3 / 6x6 column-pivoted Householder QR solve in scalar type T (m <= 6, n <= 6).
// A row-major m x n (destroyed), b (destroyed), x out.
template <typename T, int M, int N>
LIO_HD void qr_solve(T *A, T *b, T *x, T eps) {
  // Every loop has compile-time bounds and every array index is a loop counter, so after unrolling the whole
  // factorisation lives in registers on the device (a run-time pivot index would send A to scratch memory and
  // cost ~20 us per 6x6 solve); the pivot column is moved with predicated swaps instead of indexed ones.
  int perm[N];
#pragma unroll
  for (int j = 0; j < N; ++j) perm[j] = j;
  T maxnorm0 = 0;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    T s = 0;
#pragma unroll
    for (int i = 0; i < M; ++i) s += A[i * N + j] * A[i * N + j];
    maxnorm0 = s > maxnorm0 ? s : maxnorm0;
  }
  // Eigen 3.3 ColPivHouseholderQR: pivots stop counting at the first k with (largest remaining squared column norm) <
  // (max column norm * epsilon)^2 / rows * (rows - k); solve() uses those pivots only
  const T thresh_helper = eps * eps * maxnorm0 / T(M);
  int rank = 0;
  bool live = true;
  constexpr int steps = M < N ? M : N;
#pragma unroll
  for (int k = 0; k < steps; ++k) {
    if (live) {
      int piv = k;
      T best = -1;
#pragma unroll
      for (int j = k; j < N; ++j) {
        T s = 0;
#pragma unroll
        for (int i = k; i < M; ++i) s += A[i * N + j] * A[i * N + j];
        if (s > best) { best = s; piv = j; }
      }
      if (best < thresh_helper * T(M - k)) live = false;
      if (live) {
#pragma unroll
        for (int j = k + 1; j < N; ++j) {
          if (j == piv) {
#pragma unroll
            for (int i = 0; i < M; ++i) { T tmp = A[i * N + j]; A[i * N + j] = A[i * N + k]; A[i * N + k] = tmp; }
            int tp = perm[j]; perm[j] = perm[k]; perm[k] = tp;
          }
        }
        T alpha = A[k * N + k];
        T sigma = 0;
#pragma unroll
        for (int i = k + 1; i < M; ++i) sigma += A[i * N + k] * A[i * N + k];
        T normx = sqrt(alpha * alpha + sigma);
        if (normx == T(0)) live = false;
        if (live) {
          T beta = (alpha >= T(0)) ? -normx : normx;
          T v0 = alpha - beta;
          T vtv = v0 * v0 + sigma;
          if (vtv > T(0)) {
#pragma unroll
            for (int j = k + 1; j < N; ++j) {
              T s = v0 * A[k * N + j];
#pragma unroll
              for (int i = k + 1; i < M; ++i) s += A[i * N + k] * A[i * N + j];
              T f = T(2) * s / vtv;
              A[k * N + j] -= f * v0;
#pragma unroll
              for (int i = k + 1; i < M; ++i) A[i * N + j] -= f * A[i * N + k];
            }
            T s = v0 * b[k];
#pragma unroll
            for (int i = k + 1; i < M; ++i) s += A[i * N + k] * b[i];
            T f = T(2) * s / vtv;
            b[k] -= f * v0;
#pragma unroll
            for (int i = k + 1; i < M; ++i) b[i] -= f * A[i * N + k];
          }
          A[k * N + k] = beta;
#pragma unroll
          for (int i = k + 1; i < M; ++i) A[i * N + k] = T(0);
          ++rank;
        }
      }
    }
  }
  T y[N];
#pragma unroll
  for (int j = 0; j < N; ++j) y[j] = T(0);
#pragma unroll
  for (int i = steps - 1; i >= 0; --i) {
    if (i < rank) {
      T s = b[i];
#pragma unroll
      for (int j = i + 1; j < steps; ++j) if (j < rank) s -= A[i * N + j] * y[j];
      y[i] = s / A[i * N + i];
    }
  }
#pragma unroll
  for (int j = 0; j < N; ++j) {
#pragma unroll
    for (int t = 0; t < N; ++t) if (perm[j] == t) x[t] = y[j];
  }
}

// Number of eigenvalues of the symmetric NxN matrix A that are < tau, by Sylvester's law of inertia: the count of
// negative pivots of the LDL^T factorisation of (A - tau I).  Replaces a full eigendecomposition where only the
// degeneracy count is needed (Estimator.cc:1313-1333, PointOdometry.cc:589-608: leading eigenvalues below the
// threshold; eigenvalues are ascending, so "leading ones below" == "all below").
template <int N>
LIO_HD int count_eigs_below(const float *Ain, double tau) {
  double A[N * N];
  for (int i = 0; i < N * N; ++i) A[i] = double(Ain[i]);
  for (int i = 0; i < N; ++i) A[i * N + i] -= tau;
  int neg = 0;
  for (int k = 0; k < N; ++k) {
    double d = A[k * N + k];
    if (d == 0.0) d = 1e-300;  // exactly singular shift: nudge (measure-zero event)
    if (d < 0.0) ++neg;
    for (int i = k + 1; i < N; ++i) {
      double f = A[i * N + k] / d;
      for (int j = k + 1; j < N; ++j) A[i * N + j] -= f * A[k * N + j];
    }
  }
  return neg;
}

// Symmetric 3x3: eigenvalues (ascending, float) and the unit eigenvector of the LARGEST one (double), by cyclic
// Jacobi in double.  Used for the line fit of the corner features (PointMapping.cc:411-423: mat_D1, mat_V1 col 2).
LIO_HD void sym_eig3_top(const float *Ain, float *evals, double *vtop) {
  double A[9], U[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int i = 0; i < 9; ++i) A[i] = double(Ain[i]);
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    double dg = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
    if (off <= 1e-32 * dg || off < 1e-300) break;
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double apq = A[p * 3 + q];
        if (apq == 0.0) continue;
        double app = A[p * 3 + p], aqq = A[q * 3 + q];
        double tau = (aqq - app) / (2.0 * apq);
        double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
        double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
        for (int k = 0; k < 3; ++k) {
          double akp = A[k * 3 + p], akq = A[k * 3 + q];
          A[k * 3 + p] = c * akp - s * akq;
          A[k * 3 + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {
          double apk = A[p * 3 + k], aqk = A[q * 3 + k];
          A[p * 3 + k] = c * apk - s * aqk;
          A[q * 3 + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          double ukp = U[k * 3 + p], ukq = U[k * 3 + q];
          U[k * 3 + p] = c * ukp - s * ukq;
          U[k * 3 + q] = s * ukp + c * ukq;
        }
      }
  }
  int o0 = 0, o1 = 1, o2 = 2;
  double d0 = A[0], d1 = A[4], d2 = A[8];
  if (d1 < d0) { double t = d0; d0 = d1; d1 = t; int ti = o0; o0 = o1; o1 = ti; }
  if (d2 < d1) { double t = d1; d1 = d2; d2 = t; int ti = o1; o1 = o2; o2 = ti; }
  if (d1 < d0) { double t = d0; d0 = d1; d1 = t; int ti = o0; o0 = o1; o1 = ti; }
  evals[0] = float(d0); evals[1] = float(d1); evals[2] = float(d2);
  vtop[0] = U[0 * 3 + o2]; vtop[1] = U[1 * 3 + o2]; vtop[2] = U[2 * 3 + o2];
}

// Same contract as sym_eig3_top, closed form (trigonometric eigenvalues, eigenvector of the top eigenvalue from the
// best-conditioned pair of rows of A - lambda I): ~10x fewer fp64 operations per query than the Jacobi sweeps.  The
// top eigenvector is only consumed when lambda_2 > 3 lambda_1 (well separated), where this form is accurate to ~1e-15.
LIO_HD void sym_eig3_top_closed(const float *Ain, float *evals, double *vtop) {
  const double a00 = Ain[0], a01 = Ain[1], a02 = Ain[2], a11 = Ain[4], a12 = Ain[5], a22 = Ain[8];
  const double p1 = a01 * a01 + a02 * a02 + a12 * a12;
  const double q = (a00 + a11 + a22) / 3.0;
  const double b00 = a00 - q, b11 = a11 - q, b22 = a22 - q;
  const double p2 = b00 * b00 + b11 * b11 + b22 * b22 + 2.0 * p1;
  double e0, e1, e2;
  if (p2 <= 0.0) { e0 = e1 = e2 = q; }
  else {
    const double p = sqrt(p2 / 6.0), ip = 1.0 / p;
    const double c00 = b00 * ip, c11 = b11 * ip, c22 = b22 * ip, c01 = a01 * ip, c02 = a02 * ip, c12 = a12 * ip;
    double r = 0.5 * (c00 * (c11 * c22 - c12 * c12) - c01 * (c01 * c22 - c12 * c02) + c02 * (c01 * c12 - c11 * c02));
    r = r < -1.0 ? -1.0 : (r > 1.0 ? 1.0 : r);
    const double phi = acos(r) / 3.0;
    e2 = q + 2.0 * p * cos(phi);
    e0 = q + 2.0 * p * cos(phi + 2.0943951023931954923);  // + 2 pi / 3
    e1 = 3.0 * q - e0 - e2;
  }
  evals[0] = float(e0); evals[1] = float(e1); evals[2] = float(e2);
  // rows of A - lambda I and their pairwise cross products, all in scalars (a pointer into local arrays would push
  // them to scratch memory on the device)
  const double r00 = a00 - e2, r01 = a01, r02 = a02, r10 = a01, r11 = a11 - e2, r12 = a12, r20 = a02, r21 = a12, r22 = a22 - e2;
  const double ax = r01 * r12 - r02 * r11, ay = r02 * r10 - r00 * r12, az = r00 * r11 - r01 * r10;  // r0 x r1
  const double bx = r01 * r22 - r02 * r21, by = r02 * r20 - r00 * r22, bz = r00 * r21 - r01 * r20;  // r0 x r2
  const double cx = r11 * r22 - r12 * r21, cy = r12 * r20 - r10 * r22, cz = r10 * r21 - r11 * r20;  // r1 x r2
  const double na = ax * ax + ay * ay + az * az, nb = bx * bx + by * by + bz * bz, nc = cx * cx + cy * cy + cz * cz;
  double vx = ax, vy = ay, vz = az, nn = na;
  if (nb > nn) { vx = bx; vy = by; vz = bz; nn = nb; }
  if (nc > nn) { vx = cx; vy = cy; vz = cz; nn = nc; }
  if (nn > 0.0) { const double in = 1.0 / sqrt(nn); vtop[0] = vx * in; vtop[1] = vy * in; vtop[2] = vz * in; }
  else { vtop[0] = 1.0; vtop[1] = 0.0; vtop[2] = 0.0; }
}

// Cyclic-Jacobi eigenvalues of a symmetric NxN (N <= 6) matrix, ascending.  Accumulates in double.
template <int N>
LIO_HD void sym_eigvals(const float *Ain, float *evals) {
  double A[N * N];
  for (int i = 0; i < N * N; ++i) A[i] = double(Ain[i]);
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0, dg = 0;
    for (int i = 0; i < N; ++i) { dg += A[i * N + i] * A[i * N + i]; for (int j = i + 1; j < N; ++j) off += A[i * N + j] * A[i * N + j]; }
    if (off <= 1e-30 * dg) break;  // eigenvalues converged to ~1e-15 relative (quadratic convergence)
    for (int p = 0; p < N; ++p)
      for (int q = p + 1; q < N; ++q) {
        double apq = A[p * N + q];
        if (apq == 0.0) continue;
        double app = A[p * N + p], aqq = A[q * N + q];
        double tau = (aqq - app) / (2.0 * apq);
        double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
        double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
        for (int k = 0; k < N; ++k) {
          double akp = A[k * N + p], akq = A[k * N + q];
          A[k * N + p] = c * akp - s * akq;
          A[k * N + q] = s * akp + c * akq;
        }
        for (int k = 0; k < N; ++k) {
          double apk = A[p * N + k], aqk = A[q * N + k];
          A[p * N + k] = c * apk - s * aqk;
          A[q * N + k] = s * apk + c * aqk;
        }
      }
  }
  double d[N];
  for (int i = 0; i < N; ++i) d[i] = A[i * N + i];
  for (int i = 0; i < N; ++i)
    for (int j = i + 1; j < N; ++j)
      if (d[j] < d[i]) { double t = d[i]; d[i] = d[j]; d[j] = t; }
  for (int i = 0; i < N; ++i) evals[i] = float(d[i]);
}

}  // namespace lio
