// odometry.hip — gfx950 kernels + host loop of the LOAM scan-to-scan step (see odometry.h).
//
//   k_odo_corr   one wavefront per feature point: TransformToStart, exact nearest neighbour over the previous
//                sweep's cloud (lane-strided brute force: <= 40 k points, L2 resident), then the +-2.5-ring
//                window scan of PointOdometry.cc:353-380 / :451-488 in 64-wide chunks.  The sequential
//                "first strictly smaller wins" selection is reproduced as argmin over (distance, scan order);
//                the early `break` on the ring bound is reproduced with a ballot inside the chunk.
//   k_odo_rows   edge / plane coefficients (:391-435, :497-531), weights (A.7), rows of A and B (:548-571),
//                reduced to per-block partials.
//   k_odo_update 6x6 solve, degeneracy mask (threshold 10, A.6), update, abort test (:573-650).
//   k_odo_to_end TransformToEnd (:262-292).
// The <= 25 rounds run without a host round trip except a convergence peek every 5 rounds.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <climits>
#include <cstring>

#include "odometry.h"

namespace lio {

struct OdoArgs {
  const float4 *sharp; int nc;
  const float4 *flat; int ns;
  const float4 *lastc; int nlc;
  const float4 *lasts; int nls;
  float time_factor; int no_deskew;
  // 5 m uniform grids over the previous sweep's clouds (cell-sorted copies, original index in .w): every point within the
  // 25 m^2 acceptance gate of a query (PointOdometry.cc:349,448) lies in the 27 cells around it
  const float4 *gc_sorted; const int *gc_cells; GridDesc gc;
  const float4 *gs_sorted; const int *gs_cells; GridDesc gs;
};

__device__ inline bool odo_to_start(const float4 &pi, const Quat<float> &qe, const Vec3<float> &te, float time_factor, int no_deskew,
                                    Vec3<float> &out) {
  float s = time_factor * (pi.w - int(pi.w));
  if (no_deskew) s = 0;
  if (s < 0 || double(s) > 1.001) { out = Vec3<float>(pi.x, pi.y, pi.z); return false; }
  Vec3<float> p(pi.x - s * te.x, pi.y - s * te.y, pi.z - s * te.z);
  Quat<float> qid;
  Quat<float> qs = slerp(qid, s, qe, FLT_EPSILON);
  out = rotate(conj(qs), p);
  return true;
}

__device__ inline float odo_sqdiff(const float4 &a, const Vec3<float> &b) {
  float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
  return dx * dx + dy * dy + dz * dz;
}

// wave-wide argmin over (d, key); returns the winning key (INT_MAX when nobody has d < limit)
__device__ inline void wave_argmin(float &d, int &key) {
  for (int off = 32; off > 0; off >>= 1) {
    float od = __shfl_xor(d, off, 64);
    int ok = __shfl_xor(key, off, 64);
    if (od < d || (od == d && ok < key)) { d = od; key = ok; }
  }
}

__global__ void __launch_bounds__(64) k_odo_corr(OdoArgs a, const OdomState *__restrict__ st, int *__restrict__ idx) {
  if (st->converged) return;
  const int qi = blockIdx.x, lane = threadIdx.x;
  const bool corner = qi < a.nc;
  const float4 pi = corner ? a.sharp[qi] : a.flat[qi - a.nc];
  const float4 *cloud = corner ? a.lastc : a.lasts;
  const int n = corner ? a.nlc : a.nls;
  Quat<float> qe(st->T[3], st->T[0], st->T[1], st->T[2]);
  Vec3<float> te(st->T[4], st->T[5], st->T[6]);
  Vec3<float> sel;
  odo_to_start(pi, qe, te, a.time_factor, a.no_deskew, sel);
  // ---- exact 1-NN inside the 25 m^2 gate, ties -> lower index: lanes stride over the nine x-runs of the 27 neighbouring
  // cells (a nearest neighbour farther than one cell away is rejected by the gate anyway)
  float bd = INFINITY; int bi = INT_MAX;
  {
    const float4 *gmap = corner ? a.gc_sorted : a.gs_sorted;
    const int *gcells = corner ? a.gc_cells : a.gs_cells;
    const GridDesc &g = corner ? a.gc : a.gs;
    const int cx = int(floorf(sel.x * g.inv_cell)) - g.origin[0], cy = int(floorf(sel.y * g.inv_cell)) - g.origin[1],
              cz = int(floorf(sel.z * g.inv_cell)) - g.origin[2];
    if (cx >= 0 && cy >= 0 && cz >= 0 && cx < g.dims[0] && cy < g.dims[1] && cz < g.dims[2]) {
      for (int r = 0; r < 9; ++r) {
        const int z = cz + (r / 3 - 1), y = cy + (r % 3 - 1);
        if (z < 0 || z >= g.dims[2] || y < 0 || y >= g.dims[1]) continue;
        const int row = g.dims[0] * (y + g.dims[1] * z);
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.dims[0] - 1);
        const int rs = gcells[row + x0], re = gcells[row + x1 + 1];
        if (re <= rs) continue;
        for (int j = rs + lane; j < re; j += 64) {
          const float4 p = gmap[j];
          const float d = odo_sqdiff(p, sel);
          const int id = __float_as_int(p.w);
          if (d < bd || (d == bd && id < bi)) { bd = d; bi = id; }
        }
      }
    }
  }
  wave_argmin(bd, bi);
  int closest = -1, second = -1, third = -1;
  if (bi != INT_MAX && bd < 25.f) {
    closest = bi;
    const int cs = int(cloud[closest].w);
    float d2 = 25.f, d3 = 25.f;     // best "second" / "third" squared distances so far
    int k2 = INT_MAX, k3 = INT_MAX;  // their scan-order keys
    // The two window scans walk up to 2.5 rings (~1 500 points of an HDL-64E sweep) in chunks of 64.  A chunk's decision to go on depends
    // on its rings, but its LOADS do not: four chunks' points are requested together and then examined in scan order, stopping at the
    // first ring violation as before (one memory round trip per 256 candidates instead of one per 64: the scan was a chain of ~48
    // dependent L2 round trips per query).
    // upward scan; scan-order key = j - closest (1, 2, ...)
    for (int base4 = closest + 1; base4 < n; base4 += 256) {
      float4 pj[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int j = base4 + 64 * q + lane; pj[q] = cloud[j < n ? j : n - 1]; }
      bool stop = false;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (stop || base4 + 64 * q >= n) continue;   // (uniform)
        const int j = base4 + 64 * q + lane;
        const bool in = j < n;
        const int ring = in ? int(pj[q].w) : INT_MAX;
        const bool viol = in && (double(ring) > double(cs) + 2.5);
        const unsigned long long vm = __ballot(viol);
        const int first_viol = vm ? (__ffsll((long long)vm) - 1) : 64;
        if (in && lane < first_viol) {
          const float d = odo_sqdiff(pj[q], sel);
          const int key = j - closest;
          if (corner) {
            if (ring > cs && (d < d2 || (d == d2 && key < k2 && d < 25.f))) { d2 = d; k2 = key; }
          } else {
            if (ring <= cs) { if (d < d2 || (d == d2 && key < k2 && d < 25.f)) { d2 = d; k2 = key; } }
            else { if (d < d3 || (d == d3 && key < k3 && d < 25.f)) { d3 = d; k3 = key; } }
          }
        }
        if (vm) stop = true;
      }
      if (stop) break;
    }
    // downward scan; keys continue after every possible upward key
    const int KOFF = 1 << 24;
    for (int base4 = closest - 1; base4 >= 0; base4 -= 256) {
      float4 pj[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int j = base4 - 64 * q - lane; pj[q] = cloud[j >= 0 ? j : 0]; }
      bool stop = false;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (stop || base4 - 64 * q < 0) continue;   // (uniform)
        const int j = base4 - 64 * q - lane;
        const bool in = j >= 0;
        const int ring = in ? int(pj[q].w) : INT_MIN;
        const bool viol = in && (double(ring) < double(cs) - 2.5);
        const unsigned long long vm = __ballot(viol);
        const int first_viol = vm ? (__ffsll((long long)vm) - 1) : 64;
        if (in && lane < first_viol) {
          const float d = odo_sqdiff(pj[q], sel);
          const int key = KOFF + (closest - j);
          if (corner) {
            if (ring < cs && (d < d2 || (d == d2 && key < k2 && d < 25.f))) { d2 = d; k2 = key; }
          } else {
            if (ring >= cs) { if (d < d2 || (d == d2 && key < k2 && d < 25.f)) { d2 = d; k2 = key; } }
            else { if (d < d3 || (d == d3 && key < k3 && d < 25.f)) { d3 = d; k3 = key; } }
          }
        }
        if (vm) stop = true;
      }
      if (stop) break;
    }
    wave_argmin(d2, k2);
    if (k2 != INT_MAX && d2 < 25.f) second = k2 < KOFF ? closest + k2 : closest - (k2 - KOFF);
    if (!corner) {
      wave_argmin(d3, k3);
      if (k3 != INT_MAX && d3 < 25.f) third = k3 < KOFF ? closest + k3 : closest - (k3 - KOFF);
    }
  }
  if (lane == 0) {
    if (corner) { idx[2 * qi] = closest; idx[2 * qi + 1] = second; }
    else { int o = 2 * a.nc + 3 * (qi - a.nc); idx[o] = closest; idx[o + 1] = second; idx[o + 2] = third; }
  }
}

#define ODO_ROW_THREADS 256

__global__ void __launch_bounds__(ODO_ROW_THREADS) k_odo_rows(OdoArgs a, const OdomState *__restrict__ st, const int *__restrict__ idx, int iter,
                                                              double *__restrict__ partials) {
  if (st->converged) return;
  Quat<float> qe(st->T[3], st->T[0], st->T[1], st->T[2]);
  Vec3<float> te(st->T[4], st->T[5], st->T[6]);
  Mat3<float> Rt = transpose(toRot(qe));
  double acc[28];
#pragma unroll
  for (int k = 0; k < 28; ++k) acc[k] = 0;
  const int total = a.nc + a.ns;
  for (int qi = blockIdx.x * blockDim.x + threadIdx.x; qi < total; qi += gridDim.x * blockDim.x) {
    const bool corner = qi < a.nc;
    const float4 pi = corner ? a.sharp[qi] : a.flat[qi - a.nc];
    Vec3<float> sel;
    odo_to_start(pi, qe, te, a.time_factor, a.no_deskew, sel);
    float c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    bool ok = false;
    if (corner) {
      int i1 = idx[2 * qi], i2 = idx[2 * qi + 1];
      if (i2 >= 0) {
        float4 t1 = a.lastc[i1], t2 = a.lastc[i2];
        float x0 = sel.x, y0 = sel.y, z0 = sel.z, x1 = t1.x, y1 = t1.y, z1 = t1.z, x2 = t2.x, y2 = t2.y, z2 = t2.z;
        float a012 = sqrtf(((x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1)) * ((x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1)) +
                           ((x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1)) * ((x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1)) +
                           ((y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1)) * ((y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1)));
        float l12 = sqrtf((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2));
        float la = ((y1 - y2) * ((x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1)) + (z1 - z2) * ((x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1))) / a012 / l12;
        float lb = -((x1 - x2) * ((x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1)) - (z1 - z2) * ((y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1))) / a012 / l12;
        float lc = -((x1 - x2) * ((x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1)) + (y1 - y2) * ((y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1))) / a012 / l12;
        float ld2 = a012 / l12;
        float s = 1;
        if (iter >= 5) s = 1 - 1.8f * fabsf(ld2);
        if (double(s) > 0.1 && ld2 != 0) { ok = true; c0 = s * la; c1 = s * lb; c2 = s * lc; c3 = s * ld2; }
      }
    } else {
      int o = 2 * a.nc + 3 * (qi - a.nc);
      int i1 = idx[o], i2 = idx[o + 1], i3 = idx[o + 2];
      if (i2 >= 0 && i3 >= 0) {
        float4 t1 = a.lasts[i1], t2 = a.lasts[i2], t3 = a.lasts[i3];
        float pa = (t2.y - t1.y) * (t3.z - t1.z) - (t3.y - t1.y) * (t2.z - t1.z);
        float pb = (t2.z - t1.z) * (t3.x - t1.x) - (t3.z - t1.z) * (t2.x - t1.x);
        float pc = (t2.x - t1.x) * (t3.y - t1.y) - (t3.x - t1.x) * (t2.y - t1.y);
        float pd = -(pa * t1.x + pb * t1.y + pc * t1.z);
        float ps = sqrtf(pa * pa + pb * pb + pc * pc);
        pa /= ps; pb /= ps; pc /= ps; pd /= ps;
        float pd2 = pa * sel.x + pb * sel.y + pc * sel.z + pd;
        float s = 1;
        if (iter >= 5) s = 1 - 1.8f * fabsf(pd2) / sqrtf(sqrtf(sel.x * sel.x + sel.y * sel.y + sel.z * sel.z));
        if (double(s) > 0.1 && pd2 != 0) { ok = true; c0 = s * pa; c1 = s * pb; c2 = s * pc; c3 = s * pd2; }
      }
    }
    if (!ok) continue;
    Vec3<float> p(pi.x, pi.y, pi.z), w(c0, c1, c2);
    Vec3<float> pmt = p - te;
    Vec3<float> cc = rotate(conj(qe), pmt);
    Mat3<float> S = skew(cc);
    float r[6];
    r[0] = w.x * S(0, 0) + w.y * S(1, 0) + w.z * S(2, 0);
    r[1] = w.x * S(0, 1) + w.y * S(1, 1) + w.z * S(2, 1);
    r[2] = w.x * S(0, 2) + w.y * S(1, 2) + w.z * S(2, 2);
    r[3] = -(w.x * Rt(0, 0) + w.y * Rt(1, 0) + w.z * Rt(2, 0));
    r[4] = -(w.x * Rt(0, 1) + w.y * Rt(1, 1) + w.z * Rt(2, 1));
    r[5] = -(w.x * Rt(0, 2) + w.y * Rt(1, 2) + w.z * Rt(2, 2));
    float bb = float(-0.1 * double(c3));
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = i; j < 6; ++j) acc[k++] += double(r[i] * r[j]);
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[21 + i] += double(r[i] * bb);
    acc[27] += 1.0;
  }
  __shared__ double sm[ODO_ROW_THREADS / 64][28];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 28; ++k) {
    double v = acc[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) sm[wv][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 28) {
    double v = 0;
    for (int w = 0; w < ODO_ROW_THREADS / 64; ++w) v += sm[w][threadIdx.x];
    partials[blockIdx.x * 28 + threadIdx.x] = v;
  }
}

// the serial 6x6 step of one iteration (thread 0); see k_odo_update
__device__ __forceinline__ void odo_update_step(const double *ssum, OdomState *st, int iter);
// mail / sig: at the iterations where the host looks at the convergence flag (every fifth) the state is posted to its mailbox
// (dev.h: HostSignal) instead of being fetched with a copy + stream synchronisation
__global__ void k_odo_update(const double *__restrict__ partials, int nblocks, OdomState *st, int iter, OdomState *mail, HostSignal sig,
                             float *__restrict__ trace) {
  if (!st->converged) {
    __shared__ double ssum[28];
    reduce_partials28(partials, nblocks, ssum);
    if (threadIdx.x == 0) {
      odo_update_step(ssum, st, iter);
      // per-iteration record of transform_es_ (lio_odom_get_iteration_trace); 32 B, read back once per sweep
      for (int k = 0; k < 7; ++k) trace[iter * 8 + k] = st->T[k];
    }
  }
  if (sig.flag) {
    __syncthreads();
    if (threadIdx.x < 64) post_host_mail(sig, mail, st, int(sizeof(OdomState) / 4), threadIdx.x);
  }
}
__device__ __forceinline__ void odo_update_step(const double *ssum, OdomState *st, int iter) {
  double sum[28];
  for (int k = 0; k < 28; ++k) sum[k] = ssum[k];
  st->iters = iter + 1;
  st->kz = st->kz;  // (kept from round 0)
  const int nsel = int(sum[27]);
  st->T[7] = float(nsel);  // pad slot carries the selected-correspondence count back to the host
  if (nsel < 10) return;   // PointOdometry.cc:535
  float AtA[36], AtB[6];
  int k = 0;
  for (int r = 0; r < 6; ++r)
    for (int c = r; c < 6; ++c) { AtA[r * 6 + c] = float(sum[k]); AtA[c * 6 + r] = float(sum[k]); ++k; }
  for (int r = 0; r < 6; ++r) AtB[r] = float(sum[21 + r]);
  float Ac[36], Bc[6], X[6];
  for (int i = 0; i < 36; ++i) Ac[i] = AtA[i];
  for (int i = 0; i < 6; ++i) Bc[i] = AtB[i];
  qr_solve<float, 6, 6>(Ac, Bc, X, FLT_EPSILON);
  if (iter == 0) {
    const int kz = count_eigs_below<6>(AtA, 10.0);
    st->kz = kz;
    st->degenerate = kz > 0;
  }
  if (st->degenerate)
    for (int i = 0; i < st->kz; ++i) X[i] = 0.f;
  Quat<float> q(st->T[3], st->T[0], st->T[1], st->T[2]);
  Quat<float> R0 = normalized(q);
  Vec3<float> t(st->T[4], st->T[5], st->T[6]);
  t.x += X[3]; t.y += X[4]; t.z += X[5];
  q = q * deltaQ(Vec3<float>(X[0], X[1], X[2]));
  if (!isfinite(t.x)) t.x = 0;
  if (!isfinite(t.y)) t.y = 0;
  if (!isfinite(t.z)) t.z = 0;
  st->T[0] = q.x; st->T[1] = q.y; st->T[2] = q.z; st->T[3] = q.w; st->T[4] = t.x; st->T[5] = t.y; st->T[6] = t.z;
  Quat<float> d = R0 * conj(q);
  float ang = 2.f * atan2f(norm(d.vec()), fabsf(d.w));
  float delta_r = float(double(ang) * 180.0 / M_PI);
  double dt0 = double(X[3] * 100), dt1 = double(X[4] * 100), dt2 = double(X[5] * 100);
  float delta_t = float(sqrt(dt0 * dt0 + dt1 * dt1 + dt2 * dt2));
  if (double(delta_r) < 0.1 && double(delta_t) < 0.1) st->converged = 1;
}

__global__ void k_odo_to_end(float4 *pts, int n, const OdomState *__restrict__ st, float time_factor, int no_deskew) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Quat<float> qe(st->T[3], st->T[0], st->T[1], st->T[2]);
  Vec3<float> te(st->T[4], st->T[5], st->T[6]);
  float4 p = pts[i];
  float s = time_factor * (p.w - int(p.w));
  if (no_deskew) s = 0;
  p.x -= s * te.x; p.y -= s * te.y; p.z -= s * te.z;
  p.w = float(int(p.w));
  Quat<float> qid;
  Quat<float> qs = slerp(qid, s, qe, FLT_EPSILON);
  Vec3<float> v = rotate(conj(qs), Vec3<float>(p.x, p.y, p.z));
  v = rotate(qe, v);
  p.x = v.x + te.x; p.y = v.y + te.y; p.z = v.z + te.z;
  pts[i] = p;
}

// ------------------------------------------------------------------------------------------------
OdometryDev::OdometryDev(float scan_period, int io_ratio, int max_iter, bool no_deskew)
    : scan_period_(scan_period), time_factor_(1 / scan_period), io_ratio_(io_ratio), max_iter_(max_iter), no_deskew_(no_deskew) {
  int nd = 0;
  LIO_HIP(hipGetDeviceCount(&nd));
  if (nd <= 0) throw DeviceError("no HIP device: the product has no CPU path");
  LIO_HIP(hipStreamCreate(&stream_));
  d_state_.reserve(1);
}
OdometryDev::~OdometryDev() {
  if (h_bounds_) (void)hipHostFree(h_bounds_);
  if (h_state_) (void)hipHostFree(h_state_);
  if (stream_) (void)hipStreamDestroy(stream_);
}

// kdtree_corner_last_ / kdtree_surf_last_->setInputCloud (PointOdometry.cc:673-676) as 5 m grids over the previous sweep
void OdometryDev::BuildGrids() {
  hipStream_t s = stream_;
  bounds_.reserve(2);
  if (!h_bounds_) LIO_HIP(hipHostMalloc(reinterpret_cast<void **>(&h_bounds_), 2 * sizeof(VoxParams), hipHostMallocDefault));
  launch_cloud_bounds(last_corner_.p, int(n_last_corner_), partial_c_, bounds_.p, s);
  launch_cloud_bounds(last_surf_.p, int(n_last_surf_), partial_s_, bounds_.p + 1, s);
  LIO_HIP(hipMemcpyAsync(h_bounds_, bounds_.p, 2 * sizeof(VoxParams), hipMemcpyDeviceToHost, s));
  LIO_HIP(hipStreamSynchronize(s));
  const float cell = 5.0f * 1.0001f;
  grid_c_.build(last_corner_.p, n_last_corner_, h_bounds_[0].mn, h_bounds_[0].mx, cell, s);
  grid_s_.build(last_surf_.p, n_last_surf_, h_bounds_[1].mn, h_bounds_[1].mx, cell, s);
}

static void upload(DBuf<float4> &b, const float *src, size_t n, hipStream_t s) {
  b.reserve(std::max<size_t>(n, 1));
  if (n) LIO_HIP(hipMemcpyAsync(b.p, src, n * sizeof(float4), hipMemcpyHostToDevice, s));
}

void OdometryDev::Process(const float *sharp, size_t n_sharp, const float *less_sharp, size_t n_ls, const float *flat, size_t n_flat,
                          const float *less_flat, size_t n_lf) {
  iterations_done_ = 0; last_num_sel_ = 0; last_kz_ = 0; es_trace_.clear();
  hipStream_t s = stream_;
  upload(less_sharp_, less_sharp, n_ls, s);
  upload(less_flat_, less_flat, n_lf, s);
  if (!inited_) {  // :302-310
    LIO_HIP(hipStreamSynchronize(s));
    std::swap(last_corner_, less_sharp_); std::swap(last_surf_, less_flat_);
    n_last_corner_ = n_ls; n_last_surf_ = n_lf;
    inited_ = true;
    return;
  }
  if (enable_odom_) {
    if (!h_state_) {   // coherent: k_odo_update posts the state and its completion word here (dev.h: HostSignal)
      LIO_HIP(hipHostMalloc(reinterpret_cast<void **>(&h_state_), 128, hipHostMallocCoherent));
      static_assert(sizeof(OdomState) <= 64, "mailbox layout");
      std::memset(h_state_, 0, 128);
      h_flag_ = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(h_state_) + 64);
    }
    OdomState st{};
    st.T[0] = transform_es_.rot.x; st.T[1] = transform_es_.rot.y; st.T[2] = transform_es_.rot.z; st.T[3] = transform_es_.rot.w;
    st.T[4] = transform_es_.pos.x; st.T[5] = transform_es_.pos.y; st.T[6] = transform_es_.pos.z;
    LIO_HIP(hipMemcpyAsync(d_state_.p, &st, sizeof(st), hipMemcpyHostToDevice, s));
    if (n_last_corner_ > 10 && n_last_surf_ > 100) {
      upload(sharp_, sharp, n_sharp, s);
      upload(flat_, flat, n_flat, s);
      const int nq = int(n_sharp + n_flat);
      idx_.reserve(std::max<size_t>(2 * n_sharp + 3 * n_flat, 1));
      LIO_HIP(hipMemsetAsync(idx_.p, 0xFF, (2 * n_sharp + 3 * n_flat) * sizeof(int), s));
      BuildGrids();
      OdoArgs a{sharp_.p, int(n_sharp), flat_.p, int(n_flat), last_corner_.p, int(n_last_corner_), last_surf_.p, int(n_last_surf_), time_factor_,
                no_deskew_ ? 1 : 0, grid_c_.sorted(), grid_c_.cells(), grid_c_.desc(), grid_s_.sorted(), grid_s_.cells(), grid_s_.desc()};
      const int nb = std::max(1, std::min(cdiv(nq, ODO_ROW_THREADS), 64));
      d_partials_.reserve(size_t(nb) * 28);
      d_trace_.reserve(size_t(max_iter_) * 8);
      const bool mail = host_signal_enabled();
      HostSignal sig{};
      bool have_state = false;
      for (int iter = 0; iter < max_iter_; ++iter) {
        if (iter > 0 && iter % 5 == 0) {  // look at the abort flag where the reference refreshes correspondences
          if (mail) {
            wait_host_signal(sig, s);
          } else {
            LIO_HIP(hipMemcpyAsync(h_state_, d_state_.p, sizeof(st), hipMemcpyDeviceToHost, s));  // pinned landing zone
            LIO_HIP(hipStreamSynchronize(s));
          }
          st = *h_state_;
          if (st.converged) { have_state = true; break; }
        }
        if (nq > 0 && iter % 5 == 0) hipLaunchKernelGGL(k_odo_corr, dim3(nq), dim3(64), 0, s, a, d_state_.p, idx_.p);
        hipLaunchKernelGGL(k_odo_rows, dim3(nb), dim3(ODO_ROW_THREADS), 0, s, a, d_state_.p, idx_.p, iter, d_partials_.p);
        const bool post = mail && (iter % 5 == 4 || iter == max_iter_ - 1);
        HostSignal sg{};
        if (post) { sig.flag = h_flag_; sig.seq = ++seq_; sg = sig; }
        hipLaunchKernelGGL(k_odo_update, dim3(1), dim3(256), 0, s, d_partials_.p, nb, d_state_.p, iter, h_state_, sg, d_trace_.p);
      }
      LIO_HIP(hipGetLastError());
      if (!have_state) {
        if (sig.flag) {
          wait_host_signal(sig, s);
        } else {
          LIO_HIP(hipMemcpyAsync(h_state_, d_state_.p, sizeof(st), hipMemcpyDeviceToHost, s));
          LIO_HIP(hipStreamSynchronize(s));
        }
        st = *h_state_;
      }
      if (st.iters > 0) {   // the records of the iterations that ran (every launch behind them is complete: the state came back)
        h_trace_.resize(size_t(st.iters) * 8);
        LIO_HIP(hipMemcpyAsync(h_trace_.data(), d_trace_.p, size_t(st.iters) * 8 * sizeof(float), hipMemcpyDeviceToHost, s));
        LIO_HIP(hipStreamSynchronize(s));
        for (int k = 0; k < st.iters; ++k) {
          const float *T = &h_trace_[size_t(k) * 8];
          es_trace_.push_back(Rigid<float>(Quat<float>(T[3], T[0], T[1], T[2]), Vec3<float>(T[4], T[5], T[6])));
        }
      }
    } else {
      LIO_HIP(hipMemcpyAsync(h_state_, d_state_.p, sizeof(st), hipMemcpyDeviceToHost, s));
      LIO_HIP(hipStreamSynchronize(s));
      st = *h_state_;
    }
    iterations_done_ = st.iters;
    last_kz_ = st.degenerate ? st.kz : 0;
    last_num_sel_ = int(st.T[7]);
    transform_es_ = Rigid<float>(Quat<float>(st.T[3], st.T[0], st.T[1], st.T[2]), Vec3<float>(st.T[4], st.T[5], st.T[6]));
    // :654-656 accumulate, :660-661 TransformToEnd, :663 normalise
    transform_sum_ = compose(transform_sum_, rinverse(transform_es_));
    if (n_ls) hipLaunchKernelGGL(k_odo_to_end, dim3(cdiv(n_ls, 256)), dim3(256), 0, s, less_sharp_.p, int(n_ls), d_state_.p, time_factor_, no_deskew_ ? 1 : 0);
    if (n_lf) hipLaunchKernelGGL(k_odo_to_end, dim3(cdiv(n_lf, 256)), dim3(256), 0, s, less_flat_.p, int(n_lf), d_state_.p, time_factor_, no_deskew_ ? 1 : 0);
    LIO_HIP(hipGetLastError());
    transform_es_.rot = normalized(transform_es_.rot);
  }
  LIO_HIP(hipStreamSynchronize(s));
  std::swap(last_corner_, less_sharp_); std::swap(last_surf_, less_flat_);
  n_last_corner_ = n_ls; n_last_surf_ = n_lf;
}

size_t OdometryDev::GetLastCloud(int which, float *out) {
  const DBuf<float4> &b = which == 0 ? last_corner_ : last_surf_;
  const size_t n = which == 0 ? n_last_corner_ : n_last_surf_;
  if (out && n) {
    LIO_HIP(hipMemcpyAsync(out, b.p, n * sizeof(float4), hipMemcpyDeviceToHost, stream_));
    LIO_HIP(hipStreamSynchronize(stream_));
  }
  return n;
}

}  // namespace lio
