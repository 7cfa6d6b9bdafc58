// cloud_device.h — device-side bodies shared by the single-window kernels (cloud_kernels.hip) and the batched-window kernels
// (batch_kernels.hip): the K-NN walk over the cell grid, the plane fit of CalculateFeatures (Estimator.cc:1014-1097), the rows and
// the 6x6 step of CalculateLaserOdom (Estimator.cc:1242-1359).  One definition, so a window solved alone and a window solved inside
// a batch run the same instructions.
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <climits>

#include "cloud_kernels.h"
#include "hmath.h"

#if defined(__HIPCC__)
namespace lio {

__device__ inline bool finite3(const float4 &p) { return isfinite(p.x) && isfinite(p.y) && isfinite(p.z); }


__device__ inline int cell_coord(float v, float inv_cell) { return int(floorf(v * inv_cell)); }


// ------------------------------------------------------------------------------------------------
// CalculateFeatures (surf branch)
// ------------------------------------------------------------------------------------------------
// LPQ lanes cooperate on one query: sub-lane `sub` scans candidates sub, sub+LPQ, ... of each 3-cell x-run
// (one contiguous, coalesced stream per query group), keeps its own top-K, then the LPQ partial lists are
// merged by xor-shuffles.  The total order (d2, original index) makes the result independent of the split:
// bit-identical to the single-lane scan.  The serial dependent-load chain per lane shrinks LPQ-fold, which
// is what bounds this kernel (a query touches ~100 candidates; maps are L2-resident).
#define FEAT_LPQ 8
#ifndef KNN_BATCH
#define KNN_BATCH 4   // candidate loads in flight per lane
#endif
// Top-K list ordered by (squared distance, original index).  Both live in ONE 64-bit key — the distance's bit pattern (non-negative
// floats order like their bits) above the index — so a comparison is one v_cmp_lt_u64 instead of three compares and two logic
// ops, and a compare-exchange moves three registers instead of three guarded by that chain.  The search kernels are bound by
// vector-instruction issue (3473 VALU instructions per wave measured in k_odom_round before this form).
__device__ __forceinline__ unsigned long long knn_key(float d, int idx) {
  return (static_cast<unsigned long long>(__float_as_uint(d)) << 32) | static_cast<unsigned int>(idx);
}
template <int K>
__device__ __forceinline__ void knn_insert(unsigned long long key, int j, unsigned long long (&bk)[K], int (&bj)[K]) {
  if (key < bk[K - 1]) {
    bk[K - 1] = key; bj[K - 1] = j;
#pragma unroll
    for (int k = K - 1; k > 0; --k) {
      const bool sw = bk[k - 1] > bk[k];
      const unsigned long long tk = sw ? bk[k - 1] : bk[k];
      const int tj = sw ? bj[k - 1] : bj[k];
      bk[k - 1] = sw ? bk[k] : bk[k - 1];
      bj[k - 1] = sw ? bj[k] : bj[k - 1];
      bk[k] = tk; bj[k] = tj;
    }
  }
}
// The walk is organised around memory latency (the kernel is bound by dependent loads, not by bandwidth: a wave used to issue
// one candidate load, wait for it, compare, and only then issue the next — ~36 round trips per query):
//   1. the run bounds of the nine x-runs (3 x-adjacent cells each) of the 27-cell block: 18 independent loads, one round trip;
//   2. the nine runs seen as ONE flat candidate list of length T; sub-lane `sub` takes the flat positions sub, sub + LPQ, ...
//      and keeps KNN_BATCH loads in flight (position -> address by a select chain over the nine prefix sums, no indexed
//      register arrays), i.e. ceil(T / (LPQ * KNN_BATCH)) round trips (3-4 at ~100 candidates);
//   3. xor-shuffle merge of the LPQ partial lists.
// The candidate SET and the total order (d2, original index) are unchanged, so the result is bit-identical to the serial walk.
template <int K, int LPQ>
__device__ inline void knn_scan_group(const Vec3<float> &q, bool active, int sub, const float4 *__restrict__ map,
                                      const int *__restrict__ cells, const GridDesc &g, float (&bd)[K], int (&bi)[K], int (&bj)[K]) {
  unsigned long long bk[K];
#pragma unroll
  for (int k = 0; k < K; ++k) { bk[k] = knn_key(INFINITY, INT_MAX); bj[k] = 0; }
  int cx = cell_coord(q.x, g.inv_cell) - g.origin[0];
  int cy = cell_coord(q.y, g.inv_cell) - g.origin[1];
  int cz = cell_coord(q.z, g.inv_cell) - g.origin[2];
  if (cx < 0 || cy < 0 || cz < 0 || cx >= g.dims[0] || cy >= g.dims[1] || cz >= g.dims[2]) active = false;
  if (active) {
    // cells x-1..x+1 have consecutive ids => their points are one contiguous run of the cell-sorted array
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.dims[0] - 1);
    if constexpr (LPQ <= 2) {
      // throughput regime (one or two lanes per query — the keyframe batch: tens of millions of queries keep every CU full and
      // the kernel runs under a 64-VGPR cap): the plain run-by-run walk, nothing held in batch registers.  Measured there
      // (profiles/r2_pmc_sq_search_kernels.md): 5-8 k vector instructions per wave, 70 % of the kernel's time in VALU issue, and
      // with a query per lane the ~35-instruction insertion runs for every candidate (some lane always inserts).  So whole
      // rows are skipped: the query's own row is walked first, then the rows sharing a face, then the corners, and a row is
      // entered only if the distance from the query to that row of cells (a lower bound for every point in it, shrunk by
      // 1e-3 cell against rounding of the cell arithmetic) does not exceed the current fifth-best distance.  Exact: a skipped
      // row cannot hold a candidate that would enter the list.
      const float uy = q.y * g.inv_cell, uz = q.z * g.inv_cell;
      const float fy = uy - floorf(uy), fz = uz - floorf(uz);
      const float cell = 1.0f / g.inv_cell;
      const float ey_lo = fmaxf(fy - 1e-3f, 0.f) * cell, ey_hi = fmaxf(1.0f - fy - 1e-3f, 0.f) * cell;
      const float ez_lo = fmaxf(fz - 1e-3f, 0.f) * cell, ez_hi = fmaxf(1.0f - fz - 1e-3f, 0.f) * cell;
      // Three phases — own row | the four rows sharing a face | the four corner rows — and inside a phase ONE flat candidate list over
      // its (up to four) runs: a wave then runs a phase to its slowest lane's SUM of run lengths, not to the sum over the rows of the
      // slowest lane of each (measured on the headline map, 64 consecutive queries: 130 wave iterations row by row, 73 flat, 42 the mean
      // lane).  A row's bound is tested when its phase starts (after the own row / after the face rows), its two table entries are
      // loaded together with the phase's other rows'.  Round 6: features 4.91 -> 4.28 ms, rounds 5.93 -> 4.84 ms at 512 windows, keyframe
      // batch 38.1 k -> 47.2 k keyframes/s.
      auto walk = [&](int j) {
        const float4 pc = map[j];
        float ddx = pc.x - q.x, ddy = pc.y - q.y, ddz = pc.z - q.z;
        float d = ddx * ddx;
        d += ddy * ddy;
        d += ddz * ddz;
        knn_insert<K>(knn_key(d, __float_as_int(pc.w)), j, bk, bj);
      };
      {   // own row
        const int row = g.dims[0] * (cy + g.dims[1] * cz);
        const int a = cells[row + x0], e = cells[row + x1 + 1];
        for (int j = a + sub; j < e; j += LPQ) walk(j);
      }
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        int s0 = 0, s1 = 0, s2 = 0, s3 = 0, c1 = 0, c2 = 0, c3 = 0, T = 0;
        const float worst = __uint_as_float(static_cast<unsigned int>(bk[K - 1] >> 32));
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const int r = 1 + 4 * ph + qq;
          // (dy, dz) + 1 packed two bits each: own row, four face rows, four corner rows
          const int dy = int((0x22161u >> (2 * r)) & 3u) - 1, dz = int((0x28215u >> (2 * r)) & 3u) - 1;
          const int z = cz + dz, y = cy + dy;
          const float ey = dy < 0 ? ey_lo : (dy > 0 ? ey_hi : 0.f), ez = dz < 0 ? ez_lo : (dz > 0 ? ez_hi : 0.f);
          const bool on = z >= 0 && z < g.dims[2] && y >= 0 && y < g.dims[1] && !(ey * ey + ez * ez > worst);
          const int row = on ? g.dims[0] * (y + g.dims[1] * z) : 0;
          // this sub-lane's candidates of the run: positions a + sub, a + sub + LPQ, ... (every run is dealt out from ITS start: the
          // sub-lanes of a query may skip different rows, their lists need not be the same)
          const int a = (on ? cells[row + x0] : 0) + sub, e = on ? cells[row + x1 + 1] : 0;
          const int len = e > a ? (e - a + LPQ - 1) / LPQ : 0;
          if (qq == 0) { s0 = a; c1 = len; }
          if (qq == 1) { s1 = a; c2 = c1 + len; }
          if (qq == 2) { s2 = a; c3 = c2 + len; }
          if (qq == 3) { s3 = a; T = c3 + len; }
        }
        for (int f = 0; f < T; ++f) {
          int j = s0 + f * LPQ;
          j = f >= c1 ? s1 + (f - c1) * LPQ : j;
          j = f >= c2 ? s2 + (f - c2) * LPQ : j;
          j = f >= c3 ? s3 + (f - c3) * LPQ : j;
          walk(j);
        }
      }
    } else {
      int rs[9], pre[10];
      pre[0] = 0;
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        const int z = cz + r / 3 - 1, y = cy + r % 3 - 1;
        const bool in = z >= 0 && z < g.dims[2] && y >= 0 && y < g.dims[1];
        const int row = g.dims[0] * (y + g.dims[1] * z);
        const int a = in ? cells[row + x0] : 0, b = in ? cells[row + x1 + 1] : 0;
        rs[r] = a;
        pre[r + 1] = b - a;   // run length for now
      }
#pragma unroll
      for (int r = 0; r < 9; ++r) pre[r + 1] = pre[r] + max(pre[r + 1], 0);
      const int T = pre[9];
      for (int base = sub; base < T; base += LPQ * KNN_BATCH) {
        float4 p[KNN_BATCH];
        int jj[KNN_BATCH];
#pragma unroll
        for (int b = 0; b < KNN_BATCH; ++b) {
          const int f = base + b * LPQ;
          int j = f + rs[0];
#pragma unroll
          for (int r = 1; r < 9; ++r) j = f >= pre[r] ? f - pre[r] + rs[r] : j;
          jj[b] = f < T ? j : -1;
          p[b] = f < T ? map[j] : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int b = 0; b < KNN_BATCH; ++b) {
          if (jj[b] < 0) continue;
          float ddx = p[b].x - q.x, ddy = p[b].y - q.y, ddz = p[b].z - q.z;
          float d = ddx * ddx;
          d += ddy * ddy;
          d += ddz * ddz;
          knn_insert<K>(knn_key(d, __float_as_int(p[b].w)), jj[b], bk, bj);
        }
      }
    }
  }
  // butterfly merge of the LPQ partial lists (every lane of the wave takes part in the shuffles)
#pragma unroll
  for (int m = 1; m < LPQ; m <<= 1) {
    unsigned long long ok[K]; int oj[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const unsigned int lo = __shfl_xor(static_cast<unsigned int>(bk[k]), m, 64), hi = __shfl_xor(static_cast<unsigned int>(bk[k] >> 32), m, 64);
      ok[k] = (static_cast<unsigned long long>(hi) << 32) | lo;
      oj[k] = __shfl_xor(bj[k], m, 64);
    }
#pragma unroll
    for (int c = 0; c < K; ++c) knn_insert<K>(ok[c], oj[c], bk, bj);
  }
#pragma unroll
  for (int k = 0; k < K; ++k) { bd[k] = __uint_as_float(static_cast<unsigned int>(bk[k] >> 32)); bi[k] = int(static_cast<unsigned int>(bk[k])); }
}

struct FeatScalars { float min_match_sq_dis, min_plane_dis; int mapping_mode; float fixed_pz[3]; };
__device__ __forceinline__ FeatScalars feat_scalars(const FeatArgs &a) {
  return FeatScalars{a.min_match_sq_dis, a.min_plane_dis, a.mapping_mode, {a.fixed_pz[0], a.fixed_pz[1], a.fixed_pz[2]}};
}
// what the owner lane (sub == 0 of an in-range query) of features_eval comes back with
struct FeatResult { bool owner; uint8_t ok; float4 c; float sc; float4 abs; float4 po; int slot; };
// The fit half of a surf feature (Estimator.cc:1021-1097 / PointMapping.cc:503-619) for ONE query whose five nearest map
// points are known: 5x3 column-pivoted QR plane fit, validity, score, FOV.  q, t: the frame's transform; po: the stack point;
// sel: its image; bd4 / bi4: distance and original index of the fifth neighbour; bj: positions of the five in `map`.
template <bool MAPPING>
__device__ __forceinline__ FeatResult features_fit(const FeatScalars &a, int slot, const Quat<float> &q, const Vec3<float> &t, const float4 &po,
                                                   const Vec3<float> &sel, float bd4, int bi4, const int (&bj)[5], const float4 *__restrict__ map) {
  FeatResult res;
  res.owner = true; res.ok = 0; res.c = make_float4(0, 0, 0, 0); res.sc = 0; res.abs = make_float4(0, 0, 0, 0); res.po = po; res.slot = slot;
  uint8_t ok = 0;
  float4 c = make_float4(0, 0, 0, 0);
  float sc = 0;
  if (bi4 != INT_MAX && bd4 < a.min_match_sq_dis) {
    float A[15], B[5] = {-1, -1, -1, -1, -1}, X[3];
    float nx[5], ny[5], nz[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      float4 pn = map[bj[j]];
      nx[j] = pn.x; ny[j] = pn.y; nz[j] = pn.z;
      A[j * 3 + 0] = pn.x; A[j * 3 + 1] = pn.y; A[j * 3 + 2] = pn.z;
    }
    qr_solve<float, 5, 3>(A, B, X, FLT_EPSILON);
    float pa = X[0], pb = X[1], pc = X[2], pd = 1;
    float ps = sqrtf(pa * pa + pb * pb + pc * pc);
    pa /= ps; pb /= ps; pc /= ps; pd /= ps;
    bool plane_valid = true;
#pragma unroll
    for (int j = 0; j < 5; ++j)
      if (fabsf(pa * nx[j] + pb * ny[j] + pc * nz[j] + pd) > a.min_plane_dis) plane_valid = false;
    if (plane_valid) {
      float pd2 = pa * sel.x + pb * sel.y + pc * sel.z + pd;
      float s = 1 - 0.9f * fabsf(pd2) / sqrtf(sqrtf(sel.x * sel.x + sel.y * sel.y + sel.z * sel.z));
      // FOV test (+-60 deg about the sensor z axis, Estimator.cc:1063-1086)
      Vec3<float> rz = rotate(q, Vec3<float>(0.f, 0.f, 10.f));
      Vec3<float> pz(rz.x + t.x, rz.y + t.y, rz.z + t.z);
      if (MAPPING) pz = Vec3<float>(a.fixed_pz[0], a.fixed_pz[1], a.fixed_pz[2]);
      float dx1 = t.x - sel.x, dy1 = t.y - sel.y, dz1 = t.z - sel.z;
      float side1 = dx1 * dx1 + dy1 * dy1 + dz1 * dz1;
      float dx2 = pz.x - sel.x, dy2 = pz.y - sel.y, dz2 = pz.z - sel.z;
      float side2 = dx2 * dx2 + dy2 * dy2 + dz2 * dz2;
      float check1 = 100.0f + side1 - side2 - 10.0f * sqrtf(3.0f) * sqrtf(side1);
      float check2 = 100.0f + side1 - side2 + 10.0f * sqrtf(3.0f) * sqrtf(side1);
      bool in_fov = check1 < 0 && check2 > 0;
      if (double(s) > 0.1 && in_fov) {
        ok = 1;
        c = make_float4(s * pa, s * pb, s * pc, s * pd);
        sc = s;
        if (MAPPING) {  // PointMapping.cc:572-592
          const bool pos = pd2 > 0 || a.mapping_mode == 2;  // MapBuilder::OptimizeMap keeps the fitted sign (MapBuilder.cc:786-789)
          c = pos ? make_float4(s * pa, s * pb, s * pc, s * pd2) : make_float4(-s * pa, -s * pb, -s * pc, -s * pd2);
          res.abs = pos ? make_float4(pa, pb, pc, pd) : make_float4(-pa, -pb, -pc, -pd);
        }
      }
    }
  }
  res.ok = ok; res.c = c; res.sc = sc;
  return res;
}
template <bool MAPPING, int LPQ>
__device__ __forceinline__ FeatResult features_eval(const FeatFrame fr, const FeatScalars a, int block_x, const float *__restrict__ transforms,
                                                    const float4 *__restrict__ map, const int *__restrict__ cells, const GridDesc &g) {
  FeatResult res;
  res.owner = false; res.ok = 0; res.c = make_float4(0, 0, 0, 0); res.sc = 0; res.abs = make_float4(0, 0, 0, 0); res.po = make_float4(0, 0, 0, 0); res.slot = 0;
  const int gt = block_x * blockDim.x + threadIdx.x;
  const int it = gt / LPQ, sub = gt % LPQ;
  const bool active = it < fr.M;
  const int i = (active && fr.order) ? int(fr.order[it]) - fr.slot_off : it;   // (FeatFrame::order: which query this lane group works on)
  const float *tp = transforms + 8 * fr.tf_index;
  Quat<float> q(tp[3], tp[0], tp[1], tp[2]);
  Vec3<float> t(tp[4], tp[5], tp[6]);
  float4 po = active ? fr.stack[i] : make_float4(0, 0, 0, 0);
  Vec3<float> r = rotate(q, Vec3<float>(po.x, po.y, po.z));
  Vec3<float> sel(r.x + t.x, r.y + t.y, r.z + t.z);
  float bd[5]; int bi[5], bj[5];
  knn_scan_group<5, LPQ>(sel, active, sub, map, cells, g, bd, bi, bj);
  if (!active || sub != 0) return res;
  return features_fit<MAPPING>(a, fr.slot_off + i, q, t, po, sel, bd[4], bi[4], bj, map);
}
template <bool MAPPING, int LPQ>
__device__ __forceinline__ void features_body(const FeatFrame fr, const FeatScalars a, int block_x, const float *__restrict__ transforms,
                                              const float4 *__restrict__ map, const int *__restrict__ cells, const GridDesc &g,
                                              uint8_t *__restrict__ valid, float4 *__restrict__ coef, float *__restrict__ score,
                                              float4 *__restrict__ abs_coef) {
  const FeatResult r = features_eval<MAPPING, LPQ>(fr, a, block_x, transforms, map, cells, g);
  if (!r.owner) return;
  valid[r.slot] = r.ok; coef[r.slot] = r.c;
  if (score) score[r.slot] = r.sc;
  if (MAPPING && abs_coef && r.ok) abs_coef[r.slot] = r.abs;
}

// CalculateFeatures for every frame of the launch (blockIdx.y).  Two phases, like k_odom_round below: FEAT_THREADS lanes search
// with LPQ lanes per query and park each query's five neighbours in LDS, then ONE wave runs the plane fit with a query per lane
// (the fit used to occupy one lane in LPQ of every wave while costing all of its issue slots — these kernels are bound by
// vector-instruction issue, not by memory).
#define FEAT_THREADS 256
template <bool MAPPING, int LPQ, int THREADS = FEAT_THREADS>
__device__ __forceinline__ void features_block(const FeatFrame fr, const FeatScalars fs, int block_x, const float *__restrict__ transforms, const float4 *__restrict__ map,
                                               const int *__restrict__ cells, const GridDesc &g, uint8_t *__restrict__ valid,
                                               float4 *__restrict__ coef, float *__restrict__ score, float4 *__restrict__ abs_coef) {
  constexpr int QPB = THREADS / LPQ;
  static_assert(QPB <= 64, "the fit phase is one wave");
  __shared__ int s_bj[QPB][5];
  __shared__ float s_bd4[QPB];
  __shared__ int s_bi4[QPB];
  if (block_x * QPB >= fr.M) return;
  const float *tp = transforms + 8 * fr.tf_index;
  const Quat<float> q(tp[3], tp[0], tp[1], tp[2]);
  const Vec3<float> t(tp[4], tp[5], tp[6]);
  {
    const int ql = int(threadIdx.x) / LPQ, sub = int(threadIdx.x) % LPQ;
    const int i = block_x * QPB + ql;
    const bool active = i < fr.M;
    const float4 po = active ? fr.stack[i] : make_float4(0, 0, 0, 0);
    const Vec3<float> r = rotate(q, Vec3<float>(po.x, po.y, po.z));
    const Vec3<float> sel(r.x + t.x, r.y + t.y, r.z + t.z);
    float bd[5]; int bi[5], bj[5];
    knn_scan_group<5, LPQ>(sel, active, sub, map, cells, g, bd, bi, bj);
    if (sub == 0) {
#pragma unroll
      for (int k = 0; k < 5; ++k) s_bj[ql][k] = bj[k];
      s_bd4[ql] = bd[4]; s_bi4[ql] = bi[4];
    }
  }
  __syncthreads();
  const int ql = threadIdx.x, i = block_x * QPB + ql;
  if (ql >= QPB || i >= fr.M) return;
  const float4 po = fr.stack[i];
  const Vec3<float> r = rotate(q, Vec3<float>(po.x, po.y, po.z));
  const Vec3<float> sel(r.x + t.x, r.y + t.y, r.z + t.z);
  int bj[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) bj[k] = s_bj[ql][k];
  const FeatResult res = features_fit<MAPPING>(fs, fr.slot_off + i, q, t, po, sel, s_bd4[ql], s_bi4[ql], bj, map);
  valid[res.slot] = res.ok; coef[res.slot] = res.c;
  if (score) score[res.slot] = res.sc;
  if (MAPPING && abs_coef && res.ok) abs_coef[res.slot] = res.abs;
}

// one row of (mat_A | mat_B) of a selected feature, added to the 21 + 6 + 1 running sums (Estimator.cc:1272-1301)
__device__ __forceinline__ void odom_row_accumulate(const float4 po, const float4 c, const Quat<float> &q, const Vec3<float> &t, const Mat3<float> &Rm,
                                                    const Mat3<float> &Rinv, int b_from_coef, double (&acc)[28]) {
  Vec3<float> p(po.x, po.y, po.z), w(c.x, c.y, c.z);
  Mat3<float> RS = Rm * skew(p);
  float a[6];
  a[0] = -(w.x * RS(0, 0) + w.y * RS(1, 0) + w.z * RS(2, 0));
  a[1] = -(w.x * RS(0, 1) + w.y * RS(1, 1) + w.z * RS(2, 1));
  a[2] = -(w.x * RS(0, 2) + w.y * RS(1, 2) + w.z * RS(2, 2));
  if (b_from_coef == 2) {  // MapBuilder::OptimizeMap (MapBuilder.cc:903-914): (-w^T R skew(p)) R^-1 diag(5e-3, 5e-3, 1)
    const float t0 = a[0], t1 = a[1], t2 = a[2];
    a[0] = (t0 * Rinv(0, 0) + t1 * Rinv(1, 0) + t2 * Rinv(2, 0)) * 5e-3f;
    a[1] = (t0 * Rinv(0, 1) + t1 * Rinv(1, 1) + t2 * Rinv(2, 1)) * 5e-3f;
    a[2] = (t0 * Rinv(0, 2) + t1 * Rinv(1, 2) + t2 * Rinv(2, 2)) * 1.f;
  }
  a[3] = w.x; a[4] = w.y; a[5] = w.z;
  Vec3<float> rp = rotate(q, p);
  float d2 = w.x * (rp.x + t.x) + w.y * (rp.y + t.y) + w.z * (rp.z + t.z) + c.w;
  float bb = b_from_coef ? -c.w : -d2;
  int k = 0;
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int cc = r; cc < 6; ++cc) acc[k++] += double(a[r] * a[cc]);
#pragma unroll
  for (int r = 0; r < 6; ++r) acc[21 + r] += double(a[r] * bb);
  acc[27] += 1.0;
}


__device__ __forceinline__ void odom_update_from_sums(const double *ssum, OdomState *st, int iter, int min_rows, int left_update) {
  if (threadIdx.x != 0) return;
  double sum[28];
  for (int k = 0; k < 28; ++k) sum[k] = ssum[k];
  st->nsel = int(sum[27]);
  if (min_rows > 0 && st->nsel < min_rows) { st->iters = iter + 1; return; }
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
    const int kz = count_eigs_below<6>(AtA, 100.0);
    st->kz = kz;
    st->degenerate = kz > 0;
  }
  if (st->degenerate)
    for (int i = 0; i < st->kz; ++i) X[i] = 0.f;  // matP = diag(0..0,1..1) (A.6)
  Quat<float> q(st->T[3], st->T[0], st->T[1], st->T[2]);
  Quat<float> R0 = normalized(q);
  Vec3<float> t(st->T[4], st->T[5], st->T[6]);
  t.x += X[3]; t.y += X[4]; t.z += X[5];
  q = left_update ? deltaQ(Vec3<float>(X[0], X[1], X[2])) * q : q * deltaQ(Vec3<float>(X[0], X[1], X[2]));
  if (!isfinite(t.x)) t.x = 0;
  if (!isfinite(t.y)) t.y = 0;
  if (!isfinite(t.z)) t.z = 0;
  st->T[0] = q.x; st->T[1] = q.y; st->T[2] = q.z; st->T[3] = q.w; st->T[4] = t.x; st->T[5] = t.y; st->T[6] = t.z;
  // angularDistance(R0, q): 2*atan2(|vec(R0 * conj(q))|, |w|)
  Quat<float> d = R0 * conj(q);
  float ang = 2.f * atan2f(norm(d.vec()), fabsf(d.w));
  float delta_r = float(double(ang) * 180.0 / M_PI);
  // std::pow(float, int) promotes to double in the reference (Estimator.cc:1352)
  double dt0 = double(X[3] * 100), dt1 = double(X[4] * 100), dt2 = double(X[5] * 100);
  float delta_t = float(sqrt(dt0 * dt0 + dt1 * dt1 + dt2 * dt2));
  st->iters = iter + 1;
  if (double(delta_r) < 0.05 && double(delta_t) < 0.05) st->converged = 1;
}


// ------------------------------------------------------------------------------------------------
// One round of the newest frame's Gauss-Newton loop (Estimator::CalculateLaserOdom, Estimator.cc:1242-1359) in TWO launches
// instead of three: the search / plane-fit kernel also forms the rows of (mat_A | mat_B) of the features it has just fitted
// (and, with keep_features, of the ones it kept from the earlier rounds of the same point, Estimator.cc:978-980) and leaves
// one 28-double partial per block; the update kernel folds them (fixed order), solves the 6x6 system and tests convergence.
//
// The kernel is bound by vector-instruction issue, not by memory (A/B on the MI355X: 4 / 8 lanes per query, 4 / 8 candidate
// loads in flight and a merged first round trip all leave it at 34 us; 16 lanes per query make it slower).  With LPQ lanes per
// query the fit + row half used to run on 1 lane in LPQ while costing the whole wave's issue slots, and it is as long as the
// search half.  So the block works in two phases: all ODOM_ROUND_THREADS lanes search (LPQ per query) and park the five
// neighbours of each of the block's ODOM_ROUND_THREADS / LPQ queries in LDS; then ONE wave fits and forms rows with one query
// per lane (every lane busy) while the other waves retire.  Rows are summed in ascending query order: one partial per block.
#define ODOM_ROUND_THREADS 256
// one block's share of a round at the transform (q, t): phase 1 on all lanes, phase 2 on wave 0, which leaves the block's 28 sums
// in `out28` (lanes 0..27 of wave 0 return them; the other waves return 0 and must not use the value)
template <int LPQ, int THREADS = ODOM_ROUND_THREADS>
__device__ __forceinline__ double odom_round_block(const FeatScalars fs, FeatFrame fr, const Quat<float> q, const Vec3<float> t, const float4 *__restrict__ map,
                                                   const int *__restrict__ cells, const GridDesc &g, uint8_t *__restrict__ valid, float4 *__restrict__ coef,
                                                   float *__restrict__ score, int base_slot, int round, int keep, int block) {
  constexpr int QPB = THREADS / LPQ;   // queries per block
  static_assert(QPB <= 64, "the fit phase is one wave");
  __shared__ int s_bj[QPB][5];
  __shared__ float s_bd4[QPB];
  __shared__ int s_bi4[QPB];
  const int M = fr.M;
  fr.slot_off = base_slot + (keep ? round * M : 0);
  {   // ---- phase 1: search, LPQ lanes per query
    const int ql = threadIdx.x / LPQ, sub = threadIdx.x % LPQ;
    const int i = block * QPB + ql;
    const bool active = i < M;
    const float4 po = active ? fr.stack[i] : make_float4(0, 0, 0, 0);
    const Vec3<float> r = rotate(q, Vec3<float>(po.x, po.y, po.z));
    const Vec3<float> sel(r.x + t.x, r.y + t.y, r.z + t.z);
    float bd[5]; int bi[5], bj[5];
    knn_scan_group<5, LPQ>(sel, active, sub, map, cells, g, bd, bi, bj);
    if (sub == 0) {
#pragma unroll
      for (int k = 0; k < 5; ++k) s_bj[ql][k] = bj[k];
      s_bd4[ql] = bd[4]; s_bi4[ql] = bi[4];
    }
  }
  __syncthreads();
  double v = 0;
  if (threadIdx.x < 64) {
    // ---- phase 2 (wave 0): fit + rows, one query per lane
    const int ql = threadIdx.x;
    double acc[28];
#pragma unroll
    for (int k = 0; k < 28; ++k) acc[k] = 0;
    const int i = block * QPB + ql;
    if (ql < QPB && i < M) {
      const float4 po = fr.stack[i];
      const Vec3<float> r = rotate(q, Vec3<float>(po.x, po.y, po.z));
      const Vec3<float> sel(r.x + t.x, r.y + t.y, r.z + t.z);
      int bj[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) bj[k] = s_bj[ql][k];
      const FeatResult res = features_fit<false>(fs, fr.slot_off + i, q, t, po, sel, s_bd4[ql], s_bi4[ql], bj, map);
      valid[res.slot] = res.ok; coef[res.slot] = res.c;
      if (score) score[res.slot] = res.sc;
      const Mat3<float> Rm = toRot(q), Rinv = Rm;   // Rinv unused for b_from_coef = 0
      if (keep)
        for (int rr = 0; rr < round; ++rr) {   // the factor lists of the earlier rounds stay in the problem: ascending slot order
          const int sl = base_slot + rr * M + i;
          if (valid[sl]) odom_row_accumulate(res.po, coef[sl], q, t, Rm, Rinv, 0, acc);
        }
      if (res.ok) odom_row_accumulate(res.po, res.c, q, t, Rm, Rinv, 0, acc);
    }
    // the block's 28 sums: an xor butterfly over the wave's 64 lanes (lanes without a query hold zeros) — a fixed tree, so the
    // sums do not depend on how many lanes worked on a query, and no LDS (the first form parked 64 x 28 doubles there, which
    // capped the one-lane-per-query launch at ten waves per CU)
#pragma unroll
    for (int k = 0; k < 28; ++k) {
      double sres = acc[k];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sres += __shfl_xor(sres, o, 64);
      if (int(threadIdx.x) == k) v = sres;
    }
  }
  return v;
}


// fold of `nblocks` 28-double partials by a 1024-thread block (32 groups of rows b = g mod 32, ascending, then the group sums
// ascending), followed by the update of odom_update_body
// mail: a copy of the state in coherent pinned host memory, posted with the round's sequence number after every round (also
// by the no-op rounds behind convergence), so the host's look at the convergence flag is a read of its own memory.
__device__ __forceinline__ void odom_update_wide_block(const double *__restrict__ partials, int nblocks, OdomState *st, int iter, int min_rows,
                                                      int left_update, OdomState *mail, const HostSignal &sig) {
  if (st->converged) {
    if (sig.flag && threadIdx.x < 64) post_host_mail(sig, mail, st, int(sizeof(OdomState) / 4), threadIdx.x);
    return;
  }
  __shared__ double part[32][32];
  __shared__ double ssum[28];
  const int c = threadIdx.x & 31, gq = threadIdx.x >> 5;
  double v0 = 0, v1 = 0, v2 = 0, v3 = 0;
  if (c < 28) {
    // The sums are those of the four-at-a-time walk (rows b, b + 32, b + 64, b + 96 into v0 .. v3, the tail into v0, ascending b) — the
    // order the resident form of the loop folds in too — but the LOADS go out eight, then four, then up to three at a time: the
    // fold is a chain of memory round trips (19 rows per lane at 606 blocks: 7 trips before, 3 now), not of additions.
    int b = gq;
    for (; b + 224 < nblocks; b += 256) {
      double x[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) x[k] = partials[size_t(b + 32 * k) * 28 + c];
      v0 += x[0]; v1 += x[1]; v2 += x[2]; v3 += x[3];
      v0 += x[4]; v1 += x[5]; v2 += x[6]; v3 += x[7];
    }
    for (; b + 96 < nblocks; b += 128) {
      const double x0 = partials[size_t(b) * 28 + c], x1 = partials[size_t(b + 32) * 28 + c], x2 = partials[size_t(b + 64) * 28 + c],
                   x3 = partials[size_t(b + 96) * 28 + c];
      v0 += x0; v1 += x1; v2 += x2; v3 += x3;
    }
    {   // at most three rows are left
      const bool h0 = b < nblocks, h1 = b + 32 < nblocks, h2 = b + 64 < nblocks;
      const double x0 = h0 ? partials[size_t(b) * 28 + c] : 0.0, x1 = h1 ? partials[size_t(b + 32) * 28 + c] : 0.0,
                   x2 = h2 ? partials[size_t(b + 64) * 28 + c] : 0.0;
      if (h0) v0 += x0;
      if (h1) v0 += x1;
      if (h2) v0 += x2;
    }
  }
  part[gq][c] = (v0 + v1) + (v2 + v3);
  __syncthreads();
  if (threadIdx.x < 28) {
    double s2 = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) s2 += part[k][threadIdx.x];
    ssum[threadIdx.x] = s2;
  }
  __syncthreads();
  odom_update_from_sums(ssum, st, iter, min_rows, left_update);
  if (sig.flag) {
    __syncthreads();   // thread 0's update of *st is visible to wave 0
    if (threadIdx.x < 64) post_host_mail(sig, mail, st, int(sizeof(OdomState) / 4), threadIdx.x);
  }
}

}  // namespace lio
#endif
