// cloud_kernels.hip — gfx950 kernels for the point-cloud stages (see cloud_kernels.h for the reference
// call sites).  fp32 arithmetic mirrors the CPU operation order (no FMA contraction: -ffp-contract=off)
// so neighbour sets, validity flags and coefficients are reproducible bit for bit.
//
// Layout: clouds are float4 AoS (x,y,z,intensity) — 16 B per lane per load, the coalescing sweet spot
// (cdna_hip_programming.md §2); the K-NN grid stores points cell-sorted so a 3-cell x-run is one
// contiguous stream per lane, served out of L2 (maps are a few MB).
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <cfloat>
#include <climits>

#include "cloud_kernels.h"

#include <atomic>
#include <string>
#include "hmath.h"

namespace lio {

// ------------------------------------------------------------------------------------------------
// rigid transform + concat
// ------------------------------------------------------------------------------------------------
__global__ void k_transform_concat(ConcatArgs a, float4 *__restrict__ dst) {
  int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= a.total) return;
  int sidx = 0;
  for (int k = 1; k < a.nseg; ++k)
    if (gid >= a.seg[k].dst_off) sidx = k;
  const ConcatSeg &sg = a.seg[sidx];
  int local = gid - sg.dst_off;
  float4 p = sg.src[local];
  float4 o;
  if (sg.identity) {
    o = p;
  } else {
    const float *m = sg.tf.m;
    // pcl::transformPointCloud: m00*x + m01*y + m02*z + m03
    o.x = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    o.y = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    o.z = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    o.w = p.w;
  }
  if (sg.set_intensity) o.w = sg.intensity;
  dst[gid] = o;
}

void launch_transform_concat(const ConcatArgs &a, float4 *dst, hipStream_t s) {
  if (a.total <= 0) return;
  hipLaunchKernelGGL(k_transform_concat, dim3(cdiv(a.total, 256)), dim3(256), 0, s, a, dst);
  LIO_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// deskew (TransformToEnd)
// ------------------------------------------------------------------------------------------------
__global__ void k_deskew_to_end(float4 *pts, int n, Quat<float> qe, Vec3<float> te, float time_factor) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 p = pts[i];
  float s = time_factor * (p.w - int(p.w));
  p.x -= s * te.x; p.y -= s * te.y; p.z -= s * te.z;
  p.w -= int(p.w);
  Quat<float> qid;
  Quat<float> qs = slerp(qid, s, qe, FLT_EPSILON);
  Vec3<float> v = rotate(normalized(conj(qs)), Vec3<float>(p.x, p.y, p.z));
  v = rotate(qe, v);
  p.x = v.x + te.x; p.y = v.y + te.y; p.z = v.z + te.z;
  pts[i] = p;
}
void launch_deskew_to_end(float4 *pts, int n, const float q[4], const float p[3], float time_factor, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_deskew_to_end, dim3(cdiv(n, 256)), dim3(256), 0, s, pts, n, Quat<float>(q[3], q[0], q[1], q[2]),
                     Vec3<float>(p[0], p[1], p[2]), time_factor);
  LIO_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// voxel grid
// ------------------------------------------------------------------------------------------------
__device__ inline bool finite3(const float4 &p) { return isfinite(p.x) && isfinite(p.y) && isfinite(p.z); }

__global__ void k_bounds_partial(const float4 *__restrict__ pts, int n, float *__restrict__ partial) {
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  int cnt = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float4 p = pts[i];
    if (!finite3(p)) continue;
    ++cnt;
    mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
    mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
  }
  __shared__ float sm[7][256];
  int t = threadIdx.x;
  for (int d = 0; d < 3; ++d) { sm[d][t] = mn[d]; sm[3 + d][t] = mx[d]; }
  sm[6][t] = float(cnt);
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (t < st) {
      for (int d = 0; d < 3; ++d) { sm[d][t] = fminf(sm[d][t], sm[d][t + st]); sm[3 + d][t] = fmaxf(sm[3 + d][t], sm[3 + d][t + st]); }
      sm[6][t] += sm[6][t + st];
    }
    __syncthreads();
  }
  if (t < 7) partial[blockIdx.x * 8 + t] = sm[t][0];
}

__global__ void __launch_bounds__(256) k_bounds_final(const float *__restrict__ partial, int nb, float inv_leaf, VoxParams *out) {
  __shared__ float sm[7][256];
  const int t = threadIdx.x;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  float cnt = 0;
  for (int b = t; b < nb; b += 256) {
    for (int d = 0; d < 3; ++d) { mn[d] = fminf(mn[d], partial[b * 8 + d]); mx[d] = fmaxf(mx[d], partial[b * 8 + 3 + d]); }
    cnt += partial[b * 8 + 6];
  }
  for (int d = 0; d < 3; ++d) { sm[d][t] = mn[d]; sm[3 + d][t] = mx[d]; }
  sm[6][t] = cnt;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (t < st) {
      for (int d = 0; d < 3; ++d) { sm[d][t] = fminf(sm[d][t], sm[d][t + st]); sm[3 + d][t] = fmaxf(sm[3 + d][t], sm[3 + d][t + st]); }
      sm[6][t] += sm[6][t + st];
    }
    __syncthreads();
  }
  if (t != 0) return;
  for (int d = 0; d < 3; ++d) { mn[d] = sm[d][0]; mx[d] = sm[3 + d][0]; }
  cnt = sm[6][0];
  VoxParams v;
  long long dd[3];
  for (int d = 0; d < 3; ++d) {
    v.mn[d] = mn[d]; v.mx[d] = mx[d];
    dd[d] = (long long)((mx[d] - mn[d]) * inv_leaf) + 1;
    v.minb[d] = int(floorf(mn[d] * inv_leaf));
    int maxb = int(floorf(mx[d] * inv_leaf));
    v.divb[d] = maxb - v.minb[d] + 1;
  }
  v.overflow = (cnt > 0 && dd[0] * dd[1] * dd[2] > (long long)INT_MAX) ? 1 : 0;
  v.n_valid = int(cnt);
  *out = v;
}

__global__ void k_vox_keys(const float4 *__restrict__ pts, int n, float inv_leaf, const VoxParams *__restrict__ vp,
                           uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 p = pts[i];
  uint32_t key = 0xFFFFFFFFu;
  if (finite3(p)) {
    int i0 = int(floorf(p.x * inv_leaf) - float(vp->minb[0]));
    int i1 = int(floorf(p.y * inv_leaf) - float(vp->minb[1]));
    int i2 = int(floorf(p.z * inv_leaf) - float(vp->minb[2]));
    key = uint32_t(i0 + i1 * vp->divb[0] + i2 * vp->divb[0] * vp->divb[1]);
  }
  keys[i] = key;
  vals[i] = uint32_t(i);
}

// The fast path needs no bounds before the keys.  PCL's voxel index i0 + i1 * div0 + i2 * div0 * div1 (cells relative to the
// cloud's minimum) orders the voxels lexicographically by (cell_z, cell_y, cell_x), and so does any key that packs the ABSOLUTE
// cells floor(p * inverse_leaf) with a fixed offset per axis: 10 bits for z, 11 for y and x (+-204 m x +-409 m x +-409 m at a
// 0.4 m leaf).  A cloud that leaves that range raises `range_overflow` and the filter reruns with PCL's own index (exact path
// below).  The same pass leaves the per-block bounds the later kernels and the host need (VoxParams); they are folded by the
// extra block of k_vox_tile_heads, after the sort, so nothing waits for them.
#define VOX_KEY_THREADS 256
__global__ void __launch_bounds__(VOX_KEY_THREADS) k_vox_keys_abs(const float4 *__restrict__ pts, int n, float inv_leaf, uint32_t *__restrict__ keys,
                                                                 uint32_t *__restrict__ vals, float *__restrict__ partial, int *__restrict__ range_overflow) {
  const int i = blockIdx.x * VOX_KEY_THREADS + threadIdx.x;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  float cnt = 0;
  if (i < n) {
    const float4 p = pts[i];
    uint32_t key = 0xFFFFFFFFu;
    if (finite3(p)) {
      cnt = 1.f;
      mn[0] = mx[0] = p.x; mn[1] = mx[1] = p.y; mn[2] = mx[2] = p.z;
      const float cx = floorf(p.x * inv_leaf), cy = floorf(p.y * inv_leaf), cz = floorf(p.z * inv_leaf);
      if (fabsf(cx) < 1024.f && fabsf(cy) < 1024.f && fabsf(cz) < 511.f) key = (uint32_t(int(cz) + 512) << 22) | (uint32_t(int(cy) + 1024) << 11) | uint32_t(int(cx) + 1024);
      else *range_overflow = 1;
    }
    keys[i] = key;
    vals[i] = uint32_t(i);
  }
  __shared__ float sm[7][VOX_KEY_THREADS / 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    for (int d = 0; d < 3; ++d) { mn[d] = fminf(mn[d], __shfl_xor(mn[d], o, 64)); mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], o, 64)); }
    cnt += __shfl_xor(cnt, o, 64);
  }
  if (lane == 0) { for (int d = 0; d < 3; ++d) { sm[d][wv] = mn[d]; sm[3 + d][wv] = mx[d]; } sm[6][wv] = cnt; }
  __syncthreads();
  if (threadIdx.x < 7) {
    const int t = threadIdx.x;
    float v = sm[t][0];
    for (int w = 1; w < VOX_KEY_THREADS / 64; ++w) v = t < 3 ? fminf(v, sm[t][w]) : (t < 6 ? fmaxf(v, sm[t][w]) : v + sm[t][w]);
    partial[size_t(blockIdx.x) * 8 + t] = v;
  }
}

// heads (first entry of a run of equal keys) in each VOX_TILE-entry tile of the sorted keys
#define VOX_TILE 256
__device__ __forceinline__ bool vox_is_head(const uint32_t *__restrict__ keys, int i, int n, uint32_t k) {
  return i < n && k != 0xFFFFFFFFu && (i == 0 || keys[i - 1] != k);
}
__global__ void __launch_bounds__(VOX_TILE) k_vox_tile_heads(const uint32_t *__restrict__ keys, int n, int *__restrict__ tile_heads,
                                                             const float *__restrict__ partial, int npartial, float inv_leaf, VoxParams *__restrict__ params) {
  if (blockIdx.x == gridDim.x - 1) {
    // the extra block (fast path only: npartial > 0): bounds of the cloud from k_vox_keys_abs's per-block partials -> VoxParams
    if (npartial <= 0) return;
    __shared__ float sm[7][VOX_TILE];
    const int t = threadIdx.x;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    float cnt = 0;
    for (int b = t; b < npartial; b += VOX_TILE) {
      for (int d = 0; d < 3; ++d) { mn[d] = fminf(mn[d], partial[size_t(b) * 8 + d]); mx[d] = fmaxf(mx[d], partial[size_t(b) * 8 + 3 + d]); }
      cnt += partial[size_t(b) * 8 + 6];   // integers below 2^24: exact in any order
    }
    for (int d = 0; d < 3; ++d) { sm[d][t] = mn[d]; sm[3 + d][t] = mx[d]; }
    sm[6][t] = cnt;
    __syncthreads();
    for (int st = VOX_TILE / 2; st > 0; st >>= 1) {
      if (t < st) {
        for (int d = 0; d < 3; ++d) { sm[d][t] = fminf(sm[d][t], sm[d][t + st]); sm[3 + d][t] = fmaxf(sm[3 + d][t], sm[3 + d][t + st]); }
        sm[6][t] += sm[6][t + st];
      }
      __syncthreads();
    }
    if (t != 0) return;
    VoxParams v;
    long long dd[3];
    for (int d = 0; d < 3; ++d) {
      v.mn[d] = sm[d][0]; v.mx[d] = sm[3 + d][0];
      dd[d] = (long long)((v.mx[d] - v.mn[d]) * inv_leaf) + 1;
      v.minb[d] = int(floorf(v.mn[d] * inv_leaf));
      const int maxb = int(floorf(v.mx[d] * inv_leaf));
      v.divb[d] = maxb - v.minb[d] + 1;
    }
    v.overflow = (sm[6][0] > 0 && dd[0] * dd[1] * dd[2] > (long long)INT_MAX) ? 1 : 0;
    v.n_valid = int(sm[6][0]);
    *params = v;
    return;
  }
  __shared__ int swave[VOX_TILE / 64];
  const int i = blockIdx.x * VOX_TILE + threadIdx.x;
  const uint32_t k = i < n ? keys[i] : 0xFFFFFFFFu;
  const unsigned long long b = __ballot(vox_is_head(keys, i, n, k));
  if ((threadIdx.x & 63) == 0) swave[threadIdx.x >> 6] = __popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) tile_heads[blockIdx.x] = (swave[0] + swave[1]) + (swave[2] + swave[3]);
}

// Centroids of the sorted runs.  The tile's points are gathered into LDS by all lanes at once; the thread of a run's first
// entry then adds the run up in sorted order (stable sort => ascending original index inside a voxel: the within-voxel order
// the oracle fixes) out of LDS, and out of global memory only for the part of a run that leaves the tile.  One thread per run
// walking global memory was a chain of dependent gathers: 57 us on the 150 k-point local map against 4 us like this.
// The output slot of a run = heads in the tiles before this one + heads before it in the tile.  The last tile's block knows
// the total and posts it (with the bounds) to the host's mailbox.
struct VoxMail { int count; VoxParams params; int range_overflow; };
__global__ void __launch_bounds__(VOX_TILE) k_vox_centroids(const float4 *__restrict__ pts, const uint32_t *__restrict__ keys,
                                                            const uint32_t *__restrict__ vals, const int *__restrict__ tile_heads, int n,
                                                            float4 *__restrict__ out, int *__restrict__ count, const VoxParams *__restrict__ params,
                                                            int *__restrict__ range_overflow, VoxMail *mail, HostSignal sig) {
  __shared__ float4 sp[VOX_TILE];
  __shared__ uint32_t sk[VOX_TILE];
  __shared__ int swave[VOX_TILE / 64], sbase[VOX_TILE / 64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int base_i = blockIdx.x * VOX_TILE, i = base_i + tid;
  int before = 0;
  for (int b = tid; b < int(blockIdx.x); b += VOX_TILE) before += tile_heads[b];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
  const uint32_t k = i < n ? keys[i] : 0xFFFFFFFFu;
  sk[tid] = k;
  if (k != 0xFFFFFFFFu) sp[tid] = pts[vals[i]];
  const bool head = vox_is_head(keys, i, n, k);
  const unsigned long long hb = __ballot(head);
  if (lane == 0) { swave[wv] = __popcll(hb); sbase[wv] = before; }
  __syncthreads();
  int pos = (sbase[0] + sbase[1]) + (sbase[2] + sbase[3]);
  for (int w = 0; w < wv; ++w) pos += swave[w];
  pos += __popcll(hb & ((1ull << lane) - 1ull));
  if (blockIdx.x == gridDim.x - 1) {   // the last tile knows the total
    __shared__ VoxMail smail;
    if (tid == VOX_TILE - 1) {
      const int total = pos + (head ? 1 : 0);
      *count = total;
      smail.count = total; smail.params = *params; smail.range_overflow = *range_overflow;
      if (sig.flag) *range_overflow = 0;   // the mail carries it: leave the flag clear for the next run (no fill command in front of it)
    }
    if (sig.flag) {
      __syncthreads();
      if (tid < 64) post_host_mail(sig, mail, &smail, int(sizeof(VoxMail) / 4), tid);
    }
  }
  if (!head) return;
  float ax = 0, ay = 0, az = 0, ai = 0;
  int e = tid;
  while (e < VOX_TILE && sk[e] == k) { const float4 p = sp[e]; ax += p.x; ay += p.y; az += p.z; ai += p.w; ++e; }
  int cnt = e - tid;
  if (e == VOX_TILE) {
    int g = base_i + VOX_TILE;
    while (g < n && keys[g] == k) { const float4 p = pts[vals[g]]; ax += p.x; ay += p.y; az += p.z; ai += p.w; ++g; ++cnt; }
  }
  const float c = float(cnt);
  out[pos] = make_float4(ax / c, ay / c, az / c, ai / c);
}

// ================================================================================================
// The filter as ONE launch (round 4).  The library sort of the (voxel, index) keys is launch latency — a block sort and seven
// merges of ~6 us for 150 k keys — so the sort is replaced by what the voxel index allows: a COUNTING sort over the cloud's own
// box of cells, all phases inside one kernel whose blocks are co-resident and meet at five grid barriers:
//   0  every thread keeps its <= VOXF_PPT points in registers; bounds of the cloud (block partials, folded by every block after
//      barrier 1) -> VoxParams exactly as k_bounds_final forms it, and with it PCL's own voxel index c = i0 + i1 d0 + i2 d0 d1
//   1  per-cell counters, one BYTE per cell (four cells to a word), incremented by agent-scope atomics (executed at the memory
//      side, hence coherent over the eight L2s); the old value hands the point its arrival slot in the voxel
//   2  (after barrier 2) the counter table is read once, 64-cell segments at a time: per segment the number of points and of
//      occupied cells in front of it inside the reading wave's stretch (a wave scan), per wave the totals
//   3  (after barrier 3) a point looks up  points-before / voxels-before  = wave base + segment prefix + the bytes in front of its
//      cell in its own 64-byte segment; it writes its index to sorted_idx[points-before + slot]
//   4  (after barrier 4) a point of a voxel with company counts the voxel's indices below its own — its position in the order the
//      stable sort gave, which is the order the oracle's sum runs in — and puts its coordinates there in `ordered`
//   5  (after barrier 5) the point that arrived first in a voxel adds the voxel's points up in that order (contiguous, eight loads
//      in flight) and writes the centroid at voxels-before; every point zeroes its counter word: the table is clean for the next run.
// No fence anywhere (an agent-scope release writes back the L2's dirty lines, dev.h): everything one block reads of another is
// written with agent-scope stores (written through) and read with agent-scope loads, behind `s_waitcnt vmcnt(0)` + the barrier.
// Falls back to the sorted path (status 2 in the mail) when the box has more cells than the table, a voxel holds more than
// VOXF_MAX_CNT points, or the cloud is larger than the grid's registers; status 3 = a barrier timed out (blocks not co-resident).
#define VOXF_THREADS 1024           // one block per CU: the grid barrier sees <= 256 arrivals, the CU still runs 16 waves
#define VOXF_PPT 1
#define VOXF_BAR_LINES 16             // arrival counters per barrier, one cache line each (same-address atomics serialise at the memory side)
#define VOXF_MAX_CNT 32
#define VOXF_SEG_WORDS 16            // 64 cells
#define VOXF_LEADER_MAX 8u           // voxels of up to this many points are ordered by their leader in registers (phase 5)
struct VoxFusedArgs {
  const float4 *pts; int n; float inv_leaf;
  uint32_t *table; unsigned table_words;      // capacity; a multiple of 512 (one wave iteration of the scan)
  unsigned long long *prefix;                 // per 32-cell group that holds a point: (points << 32 | voxels) in front of it inside its block's stretch
  unsigned long long *wtot;                   // per block: the stretch's totals
  uint32_t *sorted_idx;
  float4 *ordered;                            // the points of crowded voxels in voxel order, ascending original index inside a voxel
  long long *stamps;                          // optional (LIO_DEBUG_TIMING): wall clock of block 0 and of the last block at the phase boundaries
  unsigned *acc;                              // the bounds of the cloud: seven accumulators, VOXF_ACC_STRIDE words apart (mn[3] and mx[3] as
                                              // order-preserving codes, the count of finite points); reset by the kernel itself
  unsigned *bar; unsigned target;             // 5 barriers x VOXF_BAR_LINES arrival counters (16 words apart) that only grow; a line is
                                              // complete for this launch at `target` (the grid is a multiple of VOXF_BAR_LINES blocks)
  unsigned target0;                           // the same for barrier 0, which only the launches WITHOUT a given box pass
  int box_given;                              // 1: the box of cells comes with the launch (the union of the boxes this filter has seen, plus a
  int box_minb[3], box_divb[3];               // margin): no bounds phase, no barrier 0; a point outside it ends the launch with status 4
  unsigned *abort_flag; int *bail_flag;
  float4 *out; int *count; VoxParams *params; VoxMail *mail; HostSignal sig;
  long long timeout_ticks;
};
template <typename T> __device__ __forceinline__ void agent_store(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ T agent_load(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// all blocks of the grid; false: timed out or another block gave up (the caller leaves).  Every thread's agent-scope stores are
// acknowledged before its block arrives.
__device__ __forceinline__ bool voxf_grid_sync(unsigned *ctr, unsigned target, unsigned *abort_flag, long long timeout_ticks) {
  __shared__ int s_ok;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x < 64) {
    // block b arrives on line b mod 16; `target` = arrivals per line once every block of this launch (and of all launches before
    // it) is in.  The 16 lines are polled by 16 lanes of the block's first wave.
    const int lane = threadIdx.x;
    if (lane == 0) __hip_atomic_fetch_add(ctr + (blockIdx.x % VOXF_BAR_LINES) * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long t0 = wall_clock64();
    int ok = 1;
    for (;;) {
      const bool in = lane >= VOXF_BAR_LINES || int(agent_load(ctr + lane * 16) - target) >= 0;
      if (__all(in)) break;
      if (agent_load(abort_flag) != 0u) { ok = 0; break; }
      if (wall_clock64() - t0 > timeout_ticks) { if (lane == 0) agent_store(abort_flag, 1u); ok = 0; break; }
      __builtin_amdgcn_s_sleep(2);
    }
    if (lane == 0) s_ok = ok;
  }
  __syncthreads();
  return s_ok != 0;
}
__device__ __forceinline__ unsigned voxf_bytes_sum(uint32_t x) { return __builtin_amdgcn_sad_u8(x, 0u, 0u); }
__device__ __forceinline__ unsigned voxf_bytes_nonzero(uint32_t x) { return __popc((((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u); }
#define VOXF_STAMP(k) do { if (a.stamps && tid == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) a.stamps[(blockIdx.x ? 16 : 0) + (k)] = wall_clock64(); } while (0)
// order-preserving code of a float for unsigned atomicMin / atomicMax (the bounds of the cloud are folded at the memory side: one
// atomic per block and value instead of every block reading every block's partials — 458 k coherent loads of the same 8 KB took
// 10-25 us in the first form of this kernel)
__device__ __forceinline__ unsigned voxf_enc(float f) { const unsigned b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float voxf_dec(unsigned e) { return __uint_as_float((e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e); }
#define VOXF_ACC_STRIDE 16   // words between accumulators (a cache line each)
#define VOXF_ACC_WAYS 16     // copies of each of the seven accumulators mn[3], mx[3], count: same-address atomics serialise at
                            // ~27 ns each at the memory side (tools/micro/grid_sync.hip), 256 of them cost a barrier and a half

__global__ void __launch_bounds__(VOXF_THREADS) k_vox_fused(VoxFusedArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int G = int(gridDim.x), nthreads = G * VOXF_THREADS, gid = int(blockIdx.x) * VOXF_THREADS + tid;
  constexpr int WPB = VOXF_THREADS / 64;   // waves per block
  const int nwaves = G * WPB, gwave = int(blockIdx.x) * WPB + wv;
  __shared__ float sred[7][WPB];
  __shared__ VoxParams svp;
  __shared__ int s_bail;
  __shared__ unsigned long long swtot[WPB], sbase[256];   // wave totals of this block; exclusive scan of the blocks' totals (G <= 256)
  __shared__ unsigned long long s_grand;
  __shared__ VoxMail smail;
  VOXF_STAMP(0);
  // ---- phase 0: the thread's points and the bounds
  float4 pt[VOXF_PPT];
  bool fin[VOXF_PPT];
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  float cnt = 0;
  // (Dealing the points out so that neighbouring lanes do not hit the same 64-byte line of the counter table — the clouds are mostly
  // concatenations of voxel-ordered stacks — was measured and lost: the count phase got shorter, the gathers and the scattered
  // list accesses of the later phases cost more than that, profiles/r4_vox_fused_v3_stamps.txt.)
  int pidx[VOXF_PPT];
#pragma unroll
  for (int q = 0; q < VOXF_PPT; ++q) pidx[q] = q * nthreads + gid;
#pragma unroll
  for (int q = 0; q < VOXF_PPT; ++q) {
    const int i = pidx[q];
    fin[q] = false;
    pt[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < a.n) {
      pt[q] = a.pts[i];
      fin[q] = finite3(pt[q]);
      if (fin[q]) {
        cnt += 1.f;
        mn[0] = fminf(mn[0], pt[q].x); mn[1] = fminf(mn[1], pt[q].y); mn[2] = fminf(mn[2], pt[q].z);
        mx[0] = fmaxf(mx[0], pt[q].x); mx[1] = fmaxf(mx[1], pt[q].y); mx[2] = fmaxf(mx[2], pt[q].z);
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    for (int d = 0; d < 3; ++d) { mn[d] = fminf(mn[d], __shfl_xor(mn[d], o, 64)); mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], o, 64)); }
    cnt += __shfl_xor(cnt, o, 64);
  }
  if (lane == 0) { for (int d = 0; d < 3; ++d) { sred[d][wv] = mn[d]; sred[3 + d][wv] = mx[d]; } sred[6][wv] = cnt; }
  __syncthreads();
  if (tid < 7) {
    float v = sred[tid][0];
    for (int w = 1; w < WPB; ++w) v = tid < 3 ? fminf(v, sred[tid][w]) : (tid < 6 ? fmaxf(v, sred[tid][w]) : v + sred[tid][w]);
    unsigned *acc = a.acc + (tid * VOXF_ACC_WAYS + int(blockIdx.x) % VOXF_ACC_WAYS) * VOXF_ACC_STRIDE;
    if (tid < 3) { if (v != FLT_MAX) __hip_atomic_fetch_min(acc, voxf_enc(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    else if (tid < 6) { if (v != -FLT_MAX) __hip_atomic_fetch_max(acc, voxf_enc(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    else if (v > 0.f) __hip_atomic_fetch_add(acc, unsigned(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  VOXF_STAMP(1);
  if (a.box_given) {
    // The voxel ORDER does not depend on where the box starts (PCL's index orders the cells by (z, y, x) whatever its minimum), so any
    // box that holds the cloud gives the same output; the true bounds still go to the accumulators and are read at the very end.
    if (tid == 0) {
      VoxParams v{};
      for (int d = 0; d < 3; ++d) { v.minb[d] = a.box_minb[d]; v.divb[d] = a.box_divb[d]; }
      svp = v;
      s_bail = 0;
    }
  } else {
  if (!voxf_grid_sync(a.bar + 0 * VOXF_BAR_LINES * 16, a.target0, a.abort_flag, a.timeout_ticks)) goto aborted;
  VOXF_STAMP(2);
  if (tid < 64) {
    // VoxParams exactly as k_bounds_final forms it (min / max are order-free, the count is an integer): lanes 0..15 fold the copies
    unsigned e[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      e[k] = lane < VOXF_ACC_WAYS ? agent_load(a.acc + (k * VOXF_ACC_WAYS + lane) * VOXF_ACC_STRIDE) : (k < 3 ? 0xFFFFFFFFu : 0u);
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        const unsigned t = __shfl_xor(e[k], o, 64);
        e[k] = k < 3 ? min(e[k], t) : (k < 6 ? max(e[k], t) : e[k] + t);
      }
    }
    if (tid == 0) {
    VoxParams v;
    long long dd[3];
    const unsigned total = e[6];
    for (int d = 0; d < 3; ++d) {
      const float lo = total ? voxf_dec(e[d]) : FLT_MAX, hi = total ? voxf_dec(e[3 + d]) : -FLT_MAX;
      v.mn[d] = lo; v.mx[d] = hi;
      dd[d] = (long long)((hi - lo) * a.inv_leaf) + 1;
      v.minb[d] = int(floorf(lo * a.inv_leaf));
      const int maxb = int(floorf(hi * a.inv_leaf));
      v.divb[d] = maxb - v.minb[d] + 1;
    }
    v.overflow = (total > 0 && dd[0] * dd[1] * dd[2] > (long long)INT_MAX) ? 1 : 0;
    v.n_valid = int(total);
    svp = v;
    const long long cells = total > 0 ? (long long)v.divb[0] * v.divb[1] * v.divb[2] : 0;
    // the scan reads whole wave iterations (512 words = 2048 cells): the box must fit the table with that rounding
    s_bail = (v.overflow || total == 0 || v.divb[0] <= 0 || v.divb[1] <= 0 || v.divb[2] <= 0 || (cells + 2047) / 2048 * 512 > (long long)a.table_words) ? 1 : 0;
    }
  }
  }
  __syncthreads();
  VOXF_STAMP(12);
  {
    const VoxParams vp = svp;
    bool bail = s_bail != 0;   // uniform over the grid (a function of the folded bounds)
    // ---- phase 1: counters.  Neighbouring lanes hold neighbouring points, and the clouds are mostly concatenations of voxel-ordered
    // stacks: runs of lanes whose cells share a counter WORD send ONE atomic (the run's head adds the run's four byte increments at
    // once; a run is at most 64 long, so no byte carries) and every lane takes its arrival slot from the old word plus the number
    // of earlier lanes of the run in its own cell.  176 k single atomics on such input took 10-25 us (same-line queueing at the
    // memory side, tools/micro/grid_sync.hip).
    static_assert(VOXF_PPT == 1, "the aggregation pairs one point per lane with its neighbours");
    unsigned cell[VOXF_PPT], slot[VOXF_PPT];
    int over = 0;
    {
      cell[0] = 0xFFFFFFFFu; slot[0] = 0;
      unsigned c = 0xFFFFFFFFu;
      if (!bail && fin[0]) {
        const int i0 = int(floorf(pt[0].x * a.inv_leaf) - float(vp.minb[0]));
        const int i1 = int(floorf(pt[0].y * a.inv_leaf) - float(vp.minb[1]));
        const int i2 = int(floorf(pt[0].z * a.inv_leaf) - float(vp.minb[2]));
        if (a.box_given && (i0 < 0 || i1 < 0 || i2 < 0 || i0 >= vp.divb[0] || i1 >= vp.divb[1] || i2 >= vp.divb[2])) over = 2;
        else c = unsigned(i0 + i1 * vp.divb[0] + i2 * vp.divb[0] * vp.divb[1]);
      }
      const bool valid = c != 0xFFFFFFFFu;
      cell[0] = c;
      const unsigned w = valid ? (c >> 2) : (0xFFFFFFFFu - unsigned(lane));   // (an invalid lane matches nobody)
      const unsigned pw = __shfl_up(w, 1, 64);
      const bool head = valid && (lane == 0 || pw != w);
      const unsigned long long hm = __ballot(head);
      const unsigned long long le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
      const int hl = (valid && (hm & le)) ? 63 - __clzll((long long)(hm & le)) : lane;     // the head of this lane's run
      const unsigned sh = 8u * (c & 3u);
      const unsigned inc = valid ? (1u << sh) : 0u;
      unsigned sc = inc;   // inclusive scan of the byte increments inside the run
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(sc, o, 64);
        if (lane >= o && lane - o >= hl) sc += t;
      }
      // the run ends in front of the next head or the next invalid lane
      const unsigned long long bounds = __ballot(head || !valid);
      const unsigned long long after = (lane == 63) ? 0ull : (bounds & ~((2ull << lane) - 1ull));
      const int last = (after ? __ffsll((long long)after) - 1 : 64) - 1;
      const unsigned total = __shfl(sc, head ? last : lane, 64);
      unsigned old = 0;
      if (head) old = __hip_atomic_fetch_add(a.table + w, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      old = __shfl(old, hl, 64);
      if (valid) {
        slot[0] = ((old >> sh) & 0xFFu) + (((sc - inc) >> sh) & 0xFFu);
        if (slot[0] >= VOXF_MAX_CNT) over = over ? over : 1;   // (a byte cannot carry into its neighbour before 255 arrivals; the run is dropped at 32)
      }
    }
    if (a.stamps && tid == 0) a.stamps[32 + blockIdx.x] = wall_clock64();
    if (over) __hip_atomic_fetch_max(a.bail_flag, over, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // 2 (outside the given box) wins over 1
    VOXF_STAMP(3);
    if (!voxf_grid_sync(a.bar + 1 * VOXF_BAR_LINES * 16, a.target, a.abort_flag, a.timeout_ticks)) goto aborted;
    VOXF_STAMP(4);
    if (tid == 0) s_bail = bail ? 1 : agent_load(a.bail_flag);   // one reader per block: written before the barrier, read behind it (0, 1 or 2)
    __syncthreads();
    const int bail_code = s_bail;
    bail = bail_code != 0;   // uniform over the grid again
    // ---- phase 2: one pass over the counters of the box, a lane taking EIGHT words (32 cells) per iteration; a wave keeps the prefixes
    // of its <= 8 iterations in registers until the block has scanned its waves' totals, so they go out relative to the BLOCK's
    // stretch (one total per block to exchange) — and only for the 32-cell groups that hold a point: nobody looks the others up
    const unsigned cells = bail ? 0u : unsigned(vp.divb[0]) * unsigned(vp.divb[1]) * unsigned(vp.divb[2]);
    const unsigned iters_total = (cells + 2047u) / 2048u;                          // wave iterations of 512 words
    const unsigned iters_per_wave = (iters_total + unsigned(nwaves) - 1u) / unsigned(nwaves);   // <= 8 (table_words / 512 / nwaves)
    {
      unsigned run_p = 0, run_v = 0;
      const unsigned it0 = unsigned(gwave) * iters_per_wave;
      const unsigned it_end = min(it0 + min(iters_per_wave, 8u), iters_total);
      unsigned ep[8], ev[8], pp[8];
#pragma unroll
      for (int jb = 0; jb < 8; jb += 4) {   // the loads of four iterations go out together
        unsigned long long x[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int k = 0; k < 4; ++k) x[j][k] = 0;
          if (it0 + jb + j < it_end) {
            const unsigned long long *src = reinterpret_cast<const unsigned long long *>(a.table + size_t(it0 + jb + j) * 512u + unsigned(lane) * 8u);
#pragma unroll
            for (int k = 0; k < 4; ++k) x[j][k] = agent_load(src + k);
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          ep[jb + j] = 0; ev[jb + j] = 0; pp[jb + j] = 0;
          if (it0 + jb + j < it_end) {   // wave-uniform
            unsigned p = 0, v = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              p += voxf_bytes_sum(uint32_t(x[j][k])) + voxf_bytes_sum(uint32_t(x[j][k] >> 32));
              v += voxf_bytes_nonzero(uint32_t(x[j][k])) + voxf_bytes_nonzero(uint32_t(x[j][k] >> 32));
            }
            unsigned ip = p, iv = v;   // inclusive scan over the lanes
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
              const unsigned tp = __shfl_up(ip, o, 64), tv = __shfl_up(iv, o, 64);
              if (lane >= o) { ip += tp; iv += tv; }
            }
            ep[jb + j] = run_p + ip - p; ev[jb + j] = run_v + iv - v; pp[jb + j] = p;
            run_p += __shfl(ip, 63, 64); run_v += __shfl(iv, 63, 64);
          }
        }
      }
      if (lane == 0) swtot[wv] = (static_cast<unsigned long long>(run_p) << 32) | run_v;
      __syncthreads();
      unsigned long long woff = 0, btot = 0;
      for (int w = 0; w < WPB; ++w) { if (w < wv) woff += swtot[w]; btot += swtot[w]; }
      const unsigned op = unsigned(woff >> 32), ov = unsigned(woff);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (it0 + j < it_end && pp[j] != 0u)
          agent_store(a.prefix + (size_t(it0 + j) * 64u + unsigned(lane)), (static_cast<unsigned long long>(op + ep[j]) << 32) | (ov + ev[j]));
      if (tid == 0) agent_store(a.wtot + blockIdx.x, btot);
    }
    VOXF_STAMP(5);
    if (!voxf_grid_sync(a.bar + 2 * VOXF_BAR_LINES * 16, a.target, a.abort_flag, a.timeout_ticks)) goto aborted;
    VOXF_STAMP(6);
    // ---- phase 3: bases of the blocks' stretches (256 loads per block), then every point's place
    {
      unsigned long long v = 0;
      if (tid < 256) v = (tid < G && !bail) ? agent_load(a.wtot + tid) : 0ull;
      unsigned long long inc = v;
      if (tid < 256) {
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const unsigned lo_ = __shfl_up(unsigned(inc), o, 64), hi_ = __shfl_up(unsigned(inc >> 32), o, 64);
          if (lane >= o) inc += (static_cast<unsigned long long>(hi_) << 32) | lo_;   // (no carry between the halves: both totals are < 2^32)
        }
        if (lane == 63) swtot[wv] = inc;   // waves 0..3
      }
      __syncthreads();
      if (tid < 256) {
        unsigned long long before = inc - v;
        for (int w = 0; w < wv; ++w) before += swtot[w];
        sbase[tid] = before;
        if (tid == 255) s_grand = before + v;
      }
      __syncthreads();
    }
    const int n_out = bail ? 0 : int(unsigned(s_grand));   // occupied cells
    unsigned start[VOXF_PPT], rank[VOXF_PPT], ccount[VOXF_PPT];
#pragma unroll
    for (int q = 0; q < VOXF_PPT; ++q) {
      start[q] = 0; rank[q] = 0; ccount[q] = 0;
      if (!bail && cell[q] != 0xFFFFFFFFu) {
        const unsigned c = cell[q], grp = c >> 5, it = c >> 11;
        const unsigned long long pre = agent_load(a.prefix + grp) + sbase[(it / iters_per_wave) / unsigned(WPB)];
        unsigned p = unsigned(pre >> 32), v = unsigned(pre);
        const unsigned wsel = (c >> 2) & 7u, bsel = c & 3u;
        const unsigned long long *sp = reinterpret_cast<const unsigned long long *>(a.table + size_t(grp) * 8u);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned long long x = agent_load(sp + k);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const uint32_t wd = h ? uint32_t(x >> 32) : uint32_t(x);
            const unsigned wi = unsigned(2 * k + h);
            const uint32_t below = wi < wsel ? wd : (wi == wsel ? (wd & ((1u << (8u * bsel)) - 1u)) : 0u);
            p += voxf_bytes_sum(below); v += voxf_bytes_nonzero(below);
            if (wi == wsel) ccount[q] = (wd >> (8u * bsel)) & 0xFFu;
          }
        }
        start[q] = p; rank[q] = v;
        if (ccount[q] > 1u) agent_store(a.sorted_idx + p + slot[q], uint32_t(pidx[q]));   // a voxel of one point needs no list
      }
    }
    VOXF_STAMP(7);
    if (a.stamps && tid == 0) a.stamps[32 + 256 + blockIdx.x] = wall_clock64();
    if (!voxf_grid_sync(a.bar + 3 * VOXF_BAR_LINES * 16, a.target, a.abort_flag, a.timeout_ticks)) goto aborted;
    VOXF_STAMP(8);
    // ---- phase 4: a point's position inside its voxel = the voxel's points with a smaller original index (the order the stable
    // sort gave and the oracle's sum runs in); the points of voxels with company go to `ordered` at that position
#pragma unroll
    for (int q = 0; q < VOXF_PPT; ++q) {
      if (!bail && cell[q] != 0xFFFFFFFFu && ccount[q] > VOXF_LEADER_MAX) {   // (up to VOXF_LEADER_MAX points the leader orders them itself)
        const uint32_t me = uint32_t(pidx[q]);
        unsigned r = 0;
#pragma unroll
        for (int k = 0; k < VOXF_MAX_CNT; ++k)
          if (unsigned(k) < ccount[q]) r += (agent_load(a.sorted_idx + start[q] + k) < me) ? 1u : 0u;
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(a.ordered + start[q] + r);
        agent_store(dst, (static_cast<unsigned long long>(__float_as_uint(pt[q].y)) << 32) | __float_as_uint(pt[q].x));
        agent_store(dst + 1, (static_cast<unsigned long long>(__float_as_uint(pt[q].w)) << 32) | __float_as_uint(pt[q].z));
      }
    }
    VOXF_STAMP(9);
    if (a.stamps && tid == 0) a.stamps[32 + 512 + blockIdx.x] = wall_clock64();
    if (!voxf_grid_sync(a.bar + 4 * VOXF_BAR_LINES * 16, a.target, a.abort_flag, a.timeout_ticks)) goto aborted;
    VOXF_STAMP(10);
    // ---- phase 5: centroids by the first arrival of every voxel, eight loads in flight at a time; counters back to zero
    if (!bail) {
#pragma unroll
      for (int q = 0; q < VOXF_PPT; ++q) {
        const bool leader = cell[q] != 0xFFFFFFFFu && slot[q] == 0;
        const int m = leader ? int(ccount[q]) : 0;
        int wmax = m;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) wmax = max(wmax, __shfl_xor(wmax, o, 64));
        if (wmax == 0) continue;   // wave-uniform
        float ax = 0, ay = 0, az = 0, ai = 0;
        if (m == 1) { ax += pt[q].x; ay += pt[q].y; az += pt[q].z; ai += pt[q].w; }
        int wsmall = (m > 1 && m <= int(VOXF_LEADER_MAX)) ? 1 : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) wsmall |= __shfl_xor(wsmall, o, 64);
        if (wsmall) {   // wave-uniform: some leader of the wave has 2 .. 8 points to order itself
          const bool mine = m > 1 && m <= int(VOXF_LEADER_MAX);
          uint32_t id[VOXF_LEADER_MAX];
          float4 pp[VOXF_LEADER_MAX];
#pragma unroll
          for (int k = 0; k < int(VOXF_LEADER_MAX); ++k) id[k] = (mine && k < m) ? agent_load(a.sorted_idx + start[q] + k) : 0xFFFFFFFFu;
#pragma unroll
          for (int k = 0; k < int(VOXF_LEADER_MAX); ++k) pp[k] = (mine && k < m) ? a.pts[id[k]] : make_float4(0.f, 0.f, 0.f, 0.f);
          int rk[VOXF_LEADER_MAX];
#pragma unroll
          for (int k = 0; k < int(VOXF_LEADER_MAX); ++k) {
            int r = 0;
#pragma unroll
            for (int j = 0; j < int(VOXF_LEADER_MAX); ++j) r += (id[j] < id[k]) ? 1 : 0;   // distinct indices; the padding ranks last
            rk[k] = r;
          }
#pragma unroll
          for (int r = 0; r < int(VOXF_LEADER_MAX); ++r) {   // the adds in ascending original index
            float4 sel = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < int(VOXF_LEADER_MAX); ++k) if (rk[k] == r) sel = pp[k];
            if (mine && r < m) { ax += sel.x; ay += sel.y; az += sel.z; ai += sel.w; }
          }
        }
        int wbig = (m > int(VOXF_LEADER_MAX)) ? m : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) wbig = max(wbig, __shfl_xor(wbig, o, 64));
        if (wbig > 0) {
          const bool big = m > int(VOXF_LEADER_MAX);
          const unsigned long long *src = reinterpret_cast<const unsigned long long *>(a.ordered + start[q]);
          for (int k0 = 0; k0 < wbig; k0 += 8) {   // wave-uniform trip count
            unsigned long long lo[8], hi[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              lo[j] = 0; hi[j] = 0;
              if (big && k0 + j < m) { lo[j] = agent_load(src + 2 * (k0 + j)); hi[j] = agent_load(src + 2 * (k0 + j) + 1); }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (big && k0 + j < m) {
                ax += __uint_as_float(uint32_t(lo[j])); ay += __uint_as_float(uint32_t(lo[j] >> 32));
                az += __uint_as_float(uint32_t(hi[j])); ai += __uint_as_float(uint32_t(hi[j] >> 32));
              }
          }
        }
        if (leader) { const float c = float(m); a.out[rank[q]] = make_float4(ax / c, ay / c, az / c, ai / c); }
      }
    }
#pragma unroll
    for (int q = 0; q < VOXF_PPT; ++q)
      if (cell[q] != 0xFFFFFFFFu) a.table[cell[q] >> 2] = 0u;   // plain store: the next launch starts behind this kernel's end
    VOXF_STAMP(11);
    if (blockIdx.x == 0) {
      __shared__ VoxParams s_true;
      __shared__ int s_status;
      if (tid == 0) { s_true = vp; s_status = bail ? (bail_code == 2 ? 4 : 2) : 0; }
      __syncthreads();
      if (a.box_given && tid < 64) {
        // the cloud's own bounds for the host (VoxParams exactly as k_bounds_final forms it), from the accumulators every block fed
        // in phase 0: all of them are in (five barriers ago for the slowest)
        unsigned e[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
          e[k] = lane < VOXF_ACC_WAYS ? agent_load(a.acc + (k * VOXF_ACC_WAYS + lane) * VOXF_ACC_STRIDE) : (k < 3 ? 0xFFFFFFFFu : 0u);
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) {
            const unsigned t = __shfl_xor(e[k], o, 64);
            e[k] = k < 3 ? min(e[k], t) : (k < 6 ? max(e[k], t) : e[k] + t);
          }
        }
        if (tid == 0) {
          VoxParams v;
          long long dd[3];
          const unsigned total = e[6];
          for (int d = 0; d < 3; ++d) {
            const float lo = total ? voxf_dec(e[d]) : FLT_MAX, hi = total ? voxf_dec(e[3 + d]) : -FLT_MAX;
            v.mn[d] = lo; v.mx[d] = hi;
            dd[d] = (long long)((hi - lo) * a.inv_leaf) + 1;
            v.minb[d] = int(floorf(lo * a.inv_leaf));
            const int maxb = int(floorf(hi * a.inv_leaf));
            v.divb[d] = maxb - v.minb[d] + 1;
          }
          v.overflow = (total > 0 && dd[0] * dd[1] * dd[2] > (long long)INT_MAX) ? 1 : 0;
          v.n_valid = int(total);
          s_true = v;
          if (total == 0 && s_status == 0) s_status = 2;   // no finite point: the sorted path writes the canonical empty result
        }
      }
      __syncthreads();
      if (tid == 0) {
        *a.count = n_out;
        *a.params = s_true;
        smail.count = n_out; smail.params = s_true; smail.range_overflow = s_status;
        if (bail) __hip_atomic_store(a.bail_flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (nobody reads it any more in this launch)
      }
      __syncthreads();
      if (tid < 7 * VOXF_ACC_WAYS) a.acc[tid * VOXF_ACC_STRIDE] = tid < 3 * VOXF_ACC_WAYS ? 0xFFFFFFFFu : 0u;   // the bounds' accumulators, ready for the next launch
      if (a.sig.flag) {
        __syncthreads();
        if (tid < 64) post_host_mail(a.sig, a.mail, &smail, int(sizeof(VoxMail) / 4), tid);
      }
    }
    return;
  }
aborted:
  // a barrier timed out: the blocks are not all resident.  Status 3 goes to the host, which clears the table and takes the sorted path.
  if (blockIdx.x == 0) {
    if (tid == 0) { smail.count = 0; smail.params = VoxParams(); smail.range_overflow = 3; *a.count = 0; }
    if (a.sig.flag) {
      __syncthreads();
      if (tid < 64) post_host_mail(a.sig, a.mail, &smail, int(sizeof(VoxMail) / 4), tid);
    }
  }
}

void launch_cloud_bounds(const float4 *pts, int n, DBuf<float> &partial, VoxParams *d_out, hipStream_t s) {
  const int nb = std::max(1, std::min(cdiv(n, 256), 512));
  partial.reserve(size_t(nb) * 8);
  hipLaunchKernelGGL(k_bounds_partial, dim3(nb), dim3(256), 0, s, pts, n, partial.p);
  hipLaunchKernelGGL(k_bounds_final, dim3(1), dim3(256), 0, s, partial.p, nb, 1.0f, d_out);
  LIO_HIP(hipGetLastError());
}

bool host_signal_enabled() {
  static const bool on = [] { const char *e = std::getenv("LIO_HOST_SIGNAL"); return e ? std::atoi(e) != 0 : true; }();
  return on;
}

// One k_vox_fused at a time per process: its blocks take a whole CU each, so two of them in flight on two streams could each
// hold half the chip and wait for the other half at their first barrier.  A filter that finds the slot taken uses the sorted path.
static std::atomic<int> g_vox_fused_inflight{0};
static std::atomic<long long> g_vox_fused_launched{0}, g_vox_fused_fell_back{0};
void vox_fused_stats(long long *launched, long long *fell_back) { *launched = g_vox_fused_launched.load(); *fell_back = g_vox_fused_fell_back.load(); }

// launch() enqueues the whole filter on `s` (no host sync); finish() waits for it and returns the output count.  Two
// filters launched on two streams overlap (the scan-to-map step filters its corner and surf stacks that way).
void VoxelGridDev::launch(const float4 *in, size_t n, float leaf, DBuf<float4> &out, hipStream_t s) {
  p_in_ = in; p_n_ = n; p_out_ = &out; p_stream_ = s; p_leaf_ = leaf;
  if (n == 0) return;
  int expect = 0;
  if (fused_eligible() && g_vox_fused_inflight.compare_exchange_strong(expect, 1)) {
    fused_slot_ = true;
    try { enqueue_fused(true); } catch (...) { fused_slot_ = false; g_vox_fused_inflight.store(0); throw; }
  } else {
    enqueue(false);
  }
}

// LIO_VOX_FUSED=1 turns the one-launch form on (opt-in: measured on the MI355X it takes 47-56 us for 44 k points against ~52 us for
// the sorted path and 63 against 75 us for 150 k — every phase is one or two round trips to the memory side, where the agent-scope
// accesses that keep the eight L2s out of the picture are served, and there are nine of them between the five barriers; the solve as
// a whole gained 1.3-1.5 %, inside the run-to-run spread.  profiles/r4_vox_fused_*_stamps.txt, profiles/r4_grid_sync_micro.txt.)
static std::atomic<int> g_vox_fused_override{-1};   // lio_vox_fused_set: -1 = the environment decides, 0 / 1 = off / on
int vox_fused_set(int on) { return g_vox_fused_override.exchange(on < 0 ? -1 : (on ? 1 : 0)); }
static bool vox_fused_enabled() {
  static const bool env_on = [] { const char *e = std::getenv("LIO_VOX_FUSED"); return e ? std::atoi(e) != 0 : false; }();
  const int o = g_vox_fused_override.load(std::memory_order_relaxed);
  return o < 0 ? env_on : o != 0;
}
// blocks of k_vox_fused the device keeps resident at once (a multiple of VOXF_BAR_LINES, at most one per CU up to 256)
static int vox_fused_grid() {
  static const int g = [] {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_vox_fused, VOXF_THREADS, 0) != hipSuccess) { (void)hipGetLastError(); return 0; }
    const int blocks = std::min(256, cus * std::min(per_cu, 1));
    return blocks / VOXF_BAR_LINES * VOXF_BAR_LINES;
  }();
  return g;
}
#define VOXF_TABLE_WORDS (16u << 20)   // 64 M cells (the headline's local map: 240 x 240 x 34 m at a 0.4 m leaf = 30 M), 64 MB, zero between runs

bool VoxelGridDev::fused_eligible() const {
  if (!vox_fused_enabled() || fused_off_ || !(use_signal_ && host_signal_enabled())) return false;
  const int g = vox_fused_grid();
  return g >= VOXF_BAR_LINES && p_n_ <= size_t(g) * VOXF_THREADS * VOXF_PPT;
}

// the bounds accumulators as a launch expects them (the kernel leaves them like this; needed once, and after an aborted launch)
void VoxelGridDev::reset_fused_acc(hipStream_t s) {
  static unsigned init[7 * VOXF_ACC_WAYS * VOXF_ACC_STRIDE];
  for (unsigned &v : init) v = 0u;
  for (int k = 0; k < 3 * VOXF_ACC_WAYS; ++k) init[k * VOXF_ACC_STRIDE] = 0xFFFFFFFFu;
  LIO_HIP(hipMemcpyAsync(f_acc_.p, init, sizeof(init), hipMemcpyHostToDevice, s));
  LIO_HIP(hipStreamSynchronize(s));   // `init` is on the stack
}

// the one-launch form (k_vox_fused); finish() falls back to enqueue(false) when the kernel reports that it could not run
void VoxelGridDev::enqueue_fused(bool with_box) {
  hipStream_t s = p_stream_;
  const int g = vox_fused_grid();
  if (!f_table_.p) {
    f_table_.reserve(VOXF_TABLE_WORDS);
    LIO_HIP(hipMemsetAsync(f_table_.p, 0, size_t(VOXF_TABLE_WORDS) * sizeof(uint32_t), s));
    f_prefix_.reserve(VOXF_TABLE_WORDS / 8);   // one entry per 32-cell group (written only where a point is)
    f_wtot_.reserve(256);
    f_acc_.reserve(7 * VOXF_ACC_WAYS * VOXF_ACC_STRIDE);
    reset_fused_acc(s);
    f_bar_.reserve(5 * VOXF_BAR_LINES * 16 + 32);   // the barrier lines, then the abort flag and the bail flag (a line each)
    LIO_HIP(hipMemsetAsync(f_bar_.p, 0, f_bar_.cap * sizeof(unsigned), s));
    f_epoch_ = 0; f_epoch0_ = 0;
  }
  f_sorted_.reserve(p_n_); f_ordered_.reserve(p_n_);
  static const bool dbg_stamps = std::getenv("LIO_DEBUG_TIMING") != nullptr;
  if (dbg_stamps && !f_stamps_.p) { f_stamps_.reserve(32 + 768); LIO_HIP(hipMemsetAsync(f_stamps_.p, 0, (32 + 768) * sizeof(long long), s)); }
  params_.reserve(1);
  if (count_.cap < 2) { count_.reserve(2); LIO_HIP(hipMemsetAsync(count_.p, 0, count_.cap * sizeof(int), s)); }
  p_out_->reserve(p_n_);
  if (!h_count_) {
    LIO_HIP(hipHostMalloc(reinterpret_cast<void **>(&h_count_), 256, hipHostMallocCoherent));
    std::memset(h_count_, 0, 256);
    h_params_ = reinterpret_cast<VoxParams *>(h_count_ + 1);
    h_flag_ = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(h_count_) + 128);
  }
  ++f_epoch_;
  VoxFusedArgs a{};
  // The box of cells: the union of the boxes this filter has seen at this leaf, a few cells wider — the clouds of one caller move
  // slowly — unless that is not known yet, does not fit the table or was just found too small (finish(): status 4).
  a.box_given = 0;
  if (with_box && spec_valid_ && spec_leaf_ == p_leaf_) {
    long long cells = 1;
    for (int d = 0; d < 3; ++d) { a.box_minb[d] = spec_lo_[d] - 4; a.box_divb[d] = spec_hi_[d] - spec_lo_[d] + 1 + 8; cells *= a.box_divb[d]; }
    if ((cells + 2047) / 2048 * 512 <= (long long)VOXF_TABLE_WORDS) a.box_given = 1;
  }
  if (!a.box_given) ++f_epoch0_;
  fused_with_box_ = a.box_given != 0;
  a.pts = p_in_; a.n = int(p_n_); a.inv_leaf = 1.0f / p_leaf_;
  a.table = f_table_.p; a.table_words = VOXF_TABLE_WORDS;
  a.prefix = f_prefix_.p; a.wtot = f_wtot_.p; a.sorted_idx = f_sorted_.p; a.ordered = f_ordered_.p; a.stamps = f_stamps_.p; a.acc = f_acc_.p;
  a.bar = f_bar_.p; a.target = f_epoch_ * unsigned(g / VOXF_BAR_LINES); a.target0 = f_epoch0_ * unsigned(g / VOXF_BAR_LINES);
  a.abort_flag = f_bar_.p + 5 * VOXF_BAR_LINES * 16; a.bail_flag = reinterpret_cast<int *>(f_bar_.p + 5 * VOXF_BAR_LINES * 16 + 16);
  a.out = p_out_->p; a.count = count_.p; a.params = params_.p; a.mail = reinterpret_cast<VoxMail *>(h_count_);
  sig_ = HostSignal();
  sig_.flag = h_flag_; sig_.seq = ++seq_;
  a.sig = sig_;
  int khz = 0, dev = 0;
  (void)hipGetDevice(&dev);
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;
  a.timeout_ticks = (long long)(0.05 * 1e3 * khz);   // 50 ms at a barrier: the blocks are not all resident
  hipLaunchKernelGGL(k_vox_fused, dim3(g), dim3(VOXF_THREADS), 0, s, a);
  LIO_HIP(hipGetLastError());
  fused_pending_ = true;
  g_vox_fused_launched.fetch_add(1);
}

// exact == false: absolute-cell keys, bounds folded on the side (4 stages: keys, sort, tile heads, centroids);
// exact == true: PCL's own index from the bounds (two more launches in front), used when the cloud leaves the key's range.
void VoxelGridDev::enqueue(bool exact) {
  fused_pending_ = false;
  const float4 *in = p_in_;
  const size_t n = p_n_;
  DBuf<float4> &out = *p_out_;
  hipStream_t s = p_stream_;
  const int ni = int(n);
  const float inv_leaf = 1.0f / p_leaf_;
  const int nkb = cdiv(ni, VOX_KEY_THREADS);
  partial_.reserve(size_t(std::max(nkb, 512)) * 8);
  params_.reserve(1);
  keys_.reserve(n); keys2_.reserve(n); vals_.reserve(n); vals2_.reserve(n);
  if (count_.cap < 2) { count_.reserve(2); LIO_HIP(hipMemsetAsync(count_.p, 0, count_.cap * sizeof(int), s)); }
  out.reserve(n);
  if (!h_count_) {
    // coherent pinned memory: k_vox_centroids posts the count, the bounds and the range flag here (VoxMail), then the completion word
    LIO_HIP(hipHostMalloc(reinterpret_cast<void **>(&h_count_), 256, hipHostMallocCoherent));
    static_assert(sizeof(VoxMail) <= 128, "mailbox layout");
    std::memset(h_count_, 0, 256);
    h_params_ = reinterpret_cast<VoxParams *>(h_count_ + 1);
    h_flag_ = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(h_count_) + 128);
  }
  int *d_range = count_.p + 1;   // zero between runs: with the mailbox k_vox_centroids clears it after reading it (one fill command less)
  const bool use_sig = use_signal_ && host_signal_enabled();
  if (!use_sig) LIO_HIP(hipMemsetAsync(d_range, 0, sizeof(int), s));   // copy-back path: the host reads the flag behind the kernels, so they cannot clear it
  int npartial = 0;
  if (exact) {
    const int nb = std::min(cdiv(ni, 256), 512);
    hipLaunchKernelGGL(k_bounds_partial, dim3(nb), dim3(256), 0, s, in, ni, partial_.p);
    hipLaunchKernelGGL(k_bounds_final, dim3(1), dim3(256), 0, s, partial_.p, nb, inv_leaf, params_.p);
    hipLaunchKernelGGL(k_vox_keys, dim3(cdiv(ni, 256)), dim3(256), 0, s, in, ni, inv_leaf, params_.p, keys_.p, vals_.p);
  } else {
    hipLaunchKernelGGL(k_vox_keys_abs, dim3(nkb), dim3(VOX_KEY_THREADS), 0, s, in, ni, inv_leaf, keys_.p, vals_.p, partial_.p, d_range);
    npartial = nkb;
  }
  size_t tmp_bytes = 0;
  // LIO_VOX_SORT=onesweep: the library's radix form (histograms + four 8-bit passes) instead of the block sort + merges it picks
  // below a million keys — an A/B switch; the result is the same stable order either way
  static const bool onesweep = [] { const char *e = std::getenv("LIO_VOX_SORT"); return e && std::string(e) == "onesweep"; }();
  using OnesweepCfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;
  if (onesweep) {
    LIO_HIP(rocprim::radix_sort_pairs<OnesweepCfg>(nullptr, tmp_bytes, keys_.p, keys2_.p, vals_.p, vals2_.p, n, 0, 32, s));
    tmp_.reserve(tmp_bytes + 256);
    LIO_HIP(rocprim::radix_sort_pairs<OnesweepCfg>(tmp_.p, tmp_bytes, keys_.p, keys2_.p, vals_.p, vals2_.p, n, 0, 32, s));
  } else {
    LIO_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_.p, keys2_.p, vals_.p, vals2_.p, n, 0, 32, s));
    tmp_.reserve(tmp_bytes + 256);
    LIO_HIP(rocprim::radix_sort_pairs(tmp_.p, tmp_bytes, keys_.p, keys2_.p, vals_.p, vals2_.p, n, 0, 32, s));
  }
  const int ntiles = cdiv(ni, VOX_TILE);
  tile_heads_.reserve(ntiles);
  hipLaunchKernelGGL(k_vox_tile_heads, dim3(ntiles + 1), dim3(VOX_TILE), 0, s, keys2_.p, ni, tile_heads_.p, partial_.p, npartial, inv_leaf, params_.p);
  sig_ = HostSignal();
  if (use_sig) { sig_.flag = h_flag_; sig_.seq = ++seq_; }

  hipLaunchKernelGGL(k_vox_centroids, dim3(ntiles), dim3(VOX_TILE), 0, s, in, keys2_.p, vals2_.p, tile_heads_.p, ni, out.p, count_.p, params_.p, d_range,
                     reinterpret_cast<VoxMail *>(h_count_), sig_);
  LIO_HIP(hipGetLastError());
  if (!sig_.flag) {
    VoxMail *m = reinterpret_cast<VoxMail *>(h_count_);
    LIO_HIP(hipMemcpyAsync(&m->count, count_.p, sizeof(int), hipMemcpyDeviceToHost, s));
    LIO_HIP(hipMemcpyAsync(&m->params, params_.p, sizeof(VoxParams), hipMemcpyDeviceToHost, s));
    LIO_HIP(hipMemcpyAsync(&m->range_overflow, d_range, sizeof(int), hipMemcpyDeviceToHost, s));
  }
}

size_t VoxelGridDev::finish(VoxParams *host_params) {
  if (p_n_ == 0) {
    if (host_params) std::memset(host_params, 0, sizeof(*host_params));
    return 0;
  }
  const VoxMail *m = reinterpret_cast<const VoxMail *>(h_count_);
  struct SlotRelease { bool &held; ~SlotRelease() { if (held) { held = false; g_vox_fused_inflight.store(0); } } } slot_release{fused_slot_};
  bool redone = false, exact_done = false;
  for (;;) {
    if (sig_.flag) wait_host_signal(sig_, p_stream_);   // the count is out; the centroids follow in stream order
    else LIO_HIP(hipStreamSynchronize(p_stream_));
    const int status = m->range_overflow;
    if (fused_pending_ && f_stamps_.p && status == 0) {   // LIO_DEBUG_TIMING: the phase boundaries of block 0 and of the last block
      static int seen = 0, printed = 0;
      ++seen;
      if ((seen <= 3 || seen % 97 == 0) && printed < 12) {   // the first launches and a sample of the steady state
        ++printed;
        long long st[32];
        LIO_HIP(hipStreamSynchronize(p_stream_));
        LIO_HIP(hipMemcpy(st, f_stamps_.p, sizeof(st), hipMemcpyDeviceToHost));
        std::fprintf(stderr, "[lio_hip timing] k_vox_fused n %zu%s, 10 ns ticks from the first block's start (block 0 | last block): ", p_n_, fused_with_box_ ? " (box given)" : "");
        static const char *nm[12] = {"start", "bounds", "B1", "count", "B2", "scan", "B3", "place", "B4", "order", "B5", "centroids+clean"};
        for (int k = 0; k < 12; ++k) std::fprintf(stderr, "%s %lld|%lld  ", nm[k], st[k] - st[0], st[16 + k] - st[0]);
        std::fprintf(stderr, "(box known %lld|%lld)", st[12] - st[0], st[16 + 12] - st[0]);
        std::fprintf(stderr, "\n");
        {   // where the slowest block of a phase sits: end of count / place / order per block
          static long long blk[768];
          LIO_HIP(hipMemcpy(blk, f_stamps_.p + 32, sizeof(blk), hipMemcpyDeviceToHost));
          const int g = vox_fused_grid();
          const char *ph[3] = {"count", "place", "order"};
          for (int q = 0; q < 3; ++q) {
            int arg = 0; long long mx = -1, sum = 0;
            for (int b = 0; b < g; ++b) { const long long v = blk[q * 256 + b] - st[0]; sum += v; if (v > mx) { mx = v; arg = b; } }
            std::fprintf(stderr, "[lio_hip timing]   %s done: mean %lld, slowest block %d at %lld; blocks 0/32/64/96/128/160/192/224: %lld %lld %lld %lld %lld %lld %lld %lld\n", ph[q], sum / g, arg, mx,
                         blk[q * 256 + 0] - st[0], blk[q * 256 + 32] - st[0], blk[q * 256 + 64] - st[0], blk[q * 256 + 96] - st[0], blk[q * 256 + 128] - st[0],
                         blk[q * 256 + 160] - st[0], blk[q * 256 + 192] - st[0], blk[q * 256 + 224] - st[0]);
          }
        }
      }
    }
    // posted = past its last barrier: the slot is free for other filters — unless this one is about to launch again (status 4) or
    // has to drain first (3); the guard above releases it on the way out then
    if (fused_slot_ && status != 3 && status != 4) { fused_slot_ = false; g_vox_fused_inflight.store(0); }
    if (fused_pending_ && status == 4) {
      // a point lay outside the box that came with the launch: once more, the bounds taken inside the kernel this time (the union
      // of the boxes seen so far stays and takes this cloud's box in when that launch reports)
      ++fused_reboxed_;
      enqueue_fused(false);
      redone = true;
      continue;
    }
    if (fused_pending_ && status == 0) {
      // the box for the next launch: the union with what this cloud occupied (reset when the leaf changes)
      const VoxParams &vp = m->params;
      if (!spec_valid_ || spec_leaf_ != p_leaf_) {
        for (int d = 0; d < 3; ++d) { spec_lo_[d] = vp.minb[d]; spec_hi_[d] = vp.minb[d] + vp.divb[d] - 1; }
        spec_valid_ = true; spec_leaf_ = p_leaf_;
      } else {
        for (int d = 0; d < 3; ++d) { spec_lo_[d] = std::min(spec_lo_[d], vp.minb[d]); spec_hi_[d] = std::max(spec_hi_[d], vp.minb[d] + vp.divb[d] - 1); }
      }
    }
    if (fused_pending_ && (status == 2 || status == 3)) {
      // the one-launch form could not run (2: box larger than the counter table, a crowded voxel, no finite point; 3: a grid
      // barrier timed out): the sorted path takes the filter
      if (status == 3) {
        LIO_HIP(hipStreamSynchronize(p_stream_));
        LIO_HIP(hipMemsetAsync(f_table_.p, 0, size_t(VOXF_TABLE_WORDS) * sizeof(uint32_t), p_stream_));
        LIO_HIP(hipMemsetAsync(f_bar_.p, 0, f_bar_.cap * sizeof(unsigned), p_stream_));
        reset_fused_acc(p_stream_);
        f_epoch_ = 0; f_epoch0_ = 0;
        fused_off_ = true;
        std::fprintf(stderr, "[lio_hip] VoxelGrid: the one-launch form timed out at a grid barrier (blocks not co-resident); this filter takes the sorted path from now on\n");
      }
      ++fused_fallbacks_;
      g_vox_fused_fell_back.fetch_add(1);
      enqueue(false);
      redone = true;
      continue;
    }
    if (status == 1 && !exact_done) {
      enqueue(true);   // the cloud spans more cells than the absolute key holds
      exact_done = true; redone = true;
      continue;
    }
    // cold path: a second pass was enqueued AFTER launch() returned, i.e. after the caller may have recorded the event other
    // streams wait on — those consumers are not ordered behind it, so the output must be complete before finish() returns
    if (redone) LIO_HIP(hipStreamSynchronize(p_stream_));
    break;
  }
  int count = m->count;
  const VoxParams hp = m->params;
  if (hp.overflow) {  // PCL: "Leaf size is too small for the input dataset" -> output = input
    p_out_->reserve(p_n_);  // the caller may have swapped the buffer since launch()
    LIO_HIP(hipMemcpyAsync(p_out_->p, p_in_, p_n_ * sizeof(float4), hipMemcpyDeviceToDevice, p_stream_));
    LIO_HIP(hipStreamSynchronize(p_stream_));  // cold path: consumers on OTHER streams read the output right after finish()
    count = int(p_n_);
  }
  if (host_params) *host_params = hp;
  return size_t(count);
}

size_t VoxelGridDev::run(const float4 *in, size_t n, float leaf, DBuf<float4> &out, hipStream_t s, VoxParams *host_params) {
  launch(in, n, leaf, out, s);
  return finish(host_params);
}

VoxelGridDev::~VoxelGridDev() {
  if (fused_slot_) { fused_slot_ = false; g_vox_fused_inflight.store(0); }
  if (h_count_) (void)hipHostFree(h_count_);
}

// ------------------------------------------------------------------------------------------------
// K-NN grid
// ------------------------------------------------------------------------------------------------
__device__ inline int cell_coord(float v, float inv_cell) { return int(floorf(v * inv_cell)); }

// Counting sort by cell: a histogram with atomics hands every point a slot inside its cell, an exclusive scan over the
// dense cell table turns the counts into run starts, and a scatter places the points.  The order INSIDE a cell depends on
// the atomics and is not reproducible — by design: every consumer ranks candidates by the total order (distance, original
// index), so results are identical whatever that order is.  (A radix/merge sort of the keys cost ~45 us of dependent
// launches for 77 k points; this is three short kernels.)
__global__ void k_cell_count(const float4 *__restrict__ pts, int n, GridDesc g, uint32_t *__restrict__ keys, uint32_t *__restrict__ slot,
                             int *__restrict__ cnt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 p = pts[i];
  int cx = cell_coord(p.x, g.inv_cell) - g.origin[0];
  int cy = cell_coord(p.y, g.inv_cell) - g.origin[1];
  int cz = cell_coord(p.z, g.inv_cell) - g.origin[2];
  cx = min(max(cx, 0), g.dims[0] - 1); cy = min(max(cy, 0), g.dims[1] - 1); cz = min(max(cz, 0), g.dims[2] - 1);
  const uint32_t c = uint32_t(cx + g.dims[0] * (cy + g.dims[1] * cz));
  keys[i] = c;
  slot[i] = uint32_t(atomicAdd(&cnt[c], 1));
}

// cnt: the histogram of k_cell_count, already scanned into `starts`; every point puts its cell's count back to zero (plain
// stores of the same value), so the table is all zeros again when the build ends and the next build needs no fill command
__global__ void k_cell_place(const float4 *__restrict__ pts, const uint32_t *__restrict__ keys, const uint32_t *__restrict__ slot, int n,
                             const int *__restrict__ starts, float4 *__restrict__ sorted, int *__restrict__ cnt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 p = pts[i];
  p.w = __int_as_float(i);
  const uint32_t c = keys[i];
  sorted[starts[c] + int(slot[i])] = p;
  cnt[c] = 0;
}

void KnnGrid::build(const float4 *pts, size_t n, const float mn[3], const float mx[3], float cell, hipStream_t s) {
  desc_.inv_cell = 1.0f / cell;
  desc_.n_points = int(n);
  size_t ncells = 1;
  for (int d = 0; d < 3; ++d) {
    int lo = int(std::floor(mn[d] * desc_.inv_cell)) - 1;
    int hi = int(std::floor(mx[d] * desc_.inv_cell)) + 1;
    desc_.origin[d] = lo;
    desc_.dims[d] = hi - lo + 1;
    ncells *= size_t(desc_.dims[d]);
  }
  if (ncells > (size_t(1) << 30)) throw DeviceError("KnnGrid: cell table too large");
  cells_.reserve(ncells + 1);
  if (cnt_.cap < ncells + 1 || cnt_dirty_) {   // a fresh table starts zeroed; after that k_cell_place leaves it zeroed (no fill per build)
    cnt_.reserve(ncells + 1);
    LIO_HIP(hipMemsetAsync(cnt_.p, 0, cnt_.cap * sizeof(int), s));
  }
  cnt_dirty_ = true;   // until k_cell_place has been enqueued: a build that throws in between leaves counts behind, the next one clears them
  keys_.reserve(std::max<size_t>(n, 1)); vals_.reserve(std::max<size_t>(n, 1)); sorted_.reserve(std::max<size_t>(n, 1));
  const int ni = int(n);
  if (ni) hipLaunchKernelGGL(k_cell_count, dim3(cdiv(ni, 256)), dim3(256), 0, s, pts, ni, desc_, keys_.p, vals_.p, cnt_.p);
  size_t tmp_bytes = 0;
  LIO_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, cnt_.p, cells_.p, 0, ncells + 1, rocprim::plus<int>(), s));
  tmp_.reserve(tmp_bytes + 256);
  LIO_HIP(rocprim::exclusive_scan(tmp_.p, tmp_bytes, cnt_.p, cells_.p, 0, ncells + 1, rocprim::plus<int>(), s));
  if (ni) hipLaunchKernelGGL(k_cell_place, dim3(cdiv(ni, 256)), dim3(256), 0, s, pts, keys_.p, vals_.p, ni, cells_.p, sorted_.p, cnt_.p);
  LIO_HIP(hipGetLastError());
  cnt_dirty_ = false;
}

// K nearest (K <= 5 kept in registers) over the 27 neighbouring cells; total order (d2, original index).
template <int K>
__device__ inline void knn_scan(const Vec3<float> &q, const float4 *__restrict__ map, const int *__restrict__ cells, const GridDesc &g,
                                float (&bd)[K], int (&bi)[K], int (&bj)[K]) {
#pragma unroll
  for (int k = 0; k < K; ++k) { bd[k] = INFINITY; bi[k] = INT_MAX; bj[k] = 0; }
  int cx = cell_coord(q.x, g.inv_cell) - g.origin[0];
  int cy = cell_coord(q.y, g.inv_cell) - g.origin[1];
  int cz = cell_coord(q.z, g.inv_cell) - g.origin[2];
  if (cx < 0 || cy < 0 || cz < 0 || cx >= g.dims[0] || cy >= g.dims[1] || cz >= g.dims[2]) return;
  for (int dz = -1; dz <= 1; ++dz) {
    int z = cz + dz;
    if (z < 0 || z >= g.dims[2]) continue;
    for (int dy = -1; dy <= 1; ++dy) {
      int y = cy + dy;
      if (y < 0 || y >= g.dims[1]) continue;
      int row = g.dims[0] * (y + g.dims[1] * z);
      for (int dx = -1; dx <= 1; ++dx) {
        int x = cx + dx;
        if (x < 0 || x >= g.dims[0]) continue;
        const int c0 = cells[row + x], c1 = cells[row + x + 1];
        for (int j = c0; j < c1; ++j) {
          float4 p = map[j];
          float ddx = p.x - q.x, ddy = p.y - q.y, ddz = p.z - q.z;
          float d = ddx * ddx;
          d += ddy * ddy;
          d += ddz * ddz;
          int idx = __float_as_int(p.w);
          if (d < bd[K - 1] || (d == bd[K - 1] && idx < bi[K - 1])) {
            // sorted insertion, fully unrolled so the arrays stay in registers
            bd[K - 1] = d; bi[K - 1] = idx; bj[K - 1] = j;
#pragma unroll
            for (int k = K - 1; k > 0; --k) {
              bool sw = bd[k - 1] > bd[k] || (bd[k - 1] == bd[k] && bi[k - 1] > bi[k]);
              float td = sw ? bd[k - 1] : bd[k];
              int ti = sw ? bi[k - 1] : bi[k];
              int tj = sw ? bj[k - 1] : bj[k];
              bd[k - 1] = sw ? bd[k] : bd[k - 1];
              bi[k - 1] = sw ? bi[k] : bi[k - 1];
              bj[k - 1] = sw ? bj[k] : bj[k - 1];
              bd[k] = td; bi[k] = ti; bj[k] = tj;
            }
          }
        }
      }
    }
  }
}

template <int K>
__global__ void k_knn(const float4 *__restrict__ query, int m, float radius_sq, const float4 *__restrict__ map,
                      const int *__restrict__ cells, GridDesc g, int32_t *__restrict__ idx, float *__restrict__ sqd, int kout) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  float4 q4 = query[i];
  float bd[K]; int bi[K], bj[K];
  knn_scan<K>(Vec3<float>(q4.x, q4.y, q4.z), map, cells, g, bd, bi, bj);
  for (int k = 0; k < kout; ++k) {
    bool ok = bi[k] != INT_MAX && bd[k] < radius_sq;
    idx[i * kout + k] = ok ? bi[k] : -1;
    sqd[i * kout + k] = ok ? bd[k] : INFINITY;
  }
}

void launch_knn(const float4 *query, int m, int k, float radius_sq, const float4 *map_sorted, const int *cells, const GridDesc &g,
                int32_t *idx, float *sqd, hipStream_t s) {
  if (m <= 0) return;
  if (k == 1) hipLaunchKernelGGL(k_knn<1>, dim3(cdiv(m, 128)), dim3(128), 0, s, query, m, radius_sq, map_sorted, cells, g, idx, sqd, k);
  else hipLaunchKernelGGL(k_knn<5>, dim3(cdiv(m, 128)), dim3(128), 0, s, query, m, radius_sq, map_sorted, cells, g, idx, sqd, k);
  LIO_HIP(hipGetLastError());
}

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
      for (int r = 0; r < 9; ++r) {
        // (dy, dz) + 1 packed two bits each: own row, four face rows, four corner rows
        const int dy = int((0x22161u >> (2 * r)) & 3u) - 1, dz = int((0x28215u >> (2 * r)) & 3u) - 1;
        const int z = cz + dz, y = cy + dy;
        if (z < 0 || z >= g.dims[2] || y < 0 || y >= g.dims[1]) continue;
        const float ey = dy < 0 ? ey_lo : (dy > 0 ? ey_hi : 0.f), ez = dz < 0 ? ez_lo : (dz > 0 ? ez_hi : 0.f);
        if (ey * ey + ez * ez > __uint_as_float(static_cast<unsigned int>(bk[K - 1] >> 32))) continue;
        const int row = g.dims[0] * (y + g.dims[1] * z);
        const int a = cells[row + x0], e = cells[row + x1 + 1];
        for (int j = a + sub; j < e; j += LPQ) {
          const float4 pc = map[j];
          float ddx = pc.x - q.x, ddy = pc.y - q.y, ddz = pc.z - q.z;
          float d = ddx * ddx;
          d += ddy * ddy;
          d += ddz * ddz;
          knn_insert<K>(knn_key(d, __float_as_int(pc.w)), j, bk, bj);
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
  const int i = gt / LPQ, sub = gt % LPQ;
  const bool active = i < fr.M;
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
// One (frame, block) share of CalculateFeatures by FEAT_THREADS lanes.  SUBS > 1: the calling block is SUBS x FEAT_THREADS wide and its
// SUBS quarters take the blocks SUBS * block_x + 0 .. SUBS - 1 (the 1024-thread launch that rides behind the rounds' update block).
template <bool MAPPING, int LPQ, int SUBS>
__device__ __forceinline__ void features_block(const FeatArgs &a, int frame, int block_x, const float *__restrict__ transforms, const float4 *__restrict__ map,
                                               const int *__restrict__ cells, const GridDesc &g, uint8_t *__restrict__ valid,
                                               float4 *__restrict__ coef, float *__restrict__ score, float4 *__restrict__ abs_coef) {
  constexpr int QPB = FEAT_THREADS / LPQ;
  static_assert(QPB <= 64, "the fit phase is one wave");
  __shared__ int s_bj[SUBS][QPB][5];
  __shared__ float s_bd4[SUBS][QPB];
  __shared__ int s_bi4[SUBS][QPB];
  const int sb = SUBS > 1 ? int(threadIdx.x) / FEAT_THREADS : 0, lt = SUBS > 1 ? int(threadIdx.x) % FEAT_THREADS : int(threadIdx.x);
  const int blk = block_x * SUBS + sb;
  const FeatFrame fr = a.fr[frame];
  const bool blk_active = blk * QPB < fr.M;
  if (SUBS == 1 && !blk_active) return;
  const FeatScalars fs = feat_scalars(a);
  const float *tp = transforms + 8 * fr.tf_index;
  const Quat<float> q(tp[3], tp[0], tp[1], tp[2]);
  const Vec3<float> t(tp[4], tp[5], tp[6]);
  if (blk_active) {
    const int ql = lt / LPQ, sub = lt % LPQ;
    const int i = blk * QPB + ql;
    const bool active = i < fr.M;
    const float4 po = active ? fr.stack[i] : make_float4(0, 0, 0, 0);
    const Vec3<float> r = rotate(q, Vec3<float>(po.x, po.y, po.z));
    const Vec3<float> sel(r.x + t.x, r.y + t.y, r.z + t.z);
    float bd[5]; int bi[5], bj[5];
    knn_scan_group<5, LPQ>(sel, active, sub, map, cells, g, bd, bi, bj);
    if (sub == 0) {
#pragma unroll
      for (int k = 0; k < 5; ++k) s_bj[sb][ql][k] = bj[k];
      s_bd4[sb][ql] = bd[4]; s_bi4[sb][ql] = bi[4];
    }
  }
  __syncthreads();
  const int ql = lt, i = blk * QPB + ql;
  if (!blk_active || ql >= QPB || i >= fr.M) return;
  const float4 po = fr.stack[i];
  const Vec3<float> r = rotate(q, Vec3<float>(po.x, po.y, po.z));
  const Vec3<float> sel(r.x + t.x, r.y + t.y, r.z + t.z);
  int bj[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) bj[k] = s_bj[sb][ql][k];
  const FeatResult res = features_fit<MAPPING>(fs, fr.slot_off + i, q, t, po, sel, s_bd4[sb][ql], s_bi4[sb][ql], bj, map);
  valid[res.slot] = res.ok; coef[res.slot] = res.c;
  if (score) score[res.slot] = res.sc;
  if (MAPPING && abs_coef && res.ok) abs_coef[res.slot] = res.abs;
}
// CalculateFeatures for every frame of the launch (blockIdx.y)
template <bool MAPPING, int LPQ>
__global__ void __launch_bounds__(FEAT_THREADS) k_features(FeatArgs a, const float *__restrict__ transforms, const float4 *__restrict__ map,
                                                          const int *__restrict__ cells, GridDesc g, uint8_t *__restrict__ valid,
                                                          float4 *__restrict__ coef, float *__restrict__ score, const int *__restrict__ skip_flag,
                                                          float4 *__restrict__ abs_coef) {
  if (skip_flag && *skip_flag) return;
  features_block<MAPPING, LPQ, 1>(a, int(blockIdx.y), int(blockIdx.x), transforms, map, cells, g, valid, coef, score, abs_coef);
}

// Corner branch of the scan-to-map step: one query per FEAT_LPQ lanes, 5-NN, covariance of the 5 neighbours, line
// direction = eigenvector of the largest eigenvalue (accepted when it dominates 3x the middle one).
template <int LPQ = FEAT_LPQ>
__device__ __forceinline__ void line_features_body(int block_x, const float4 *__restrict__ stack, int M, int slot_off, const float *__restrict__ tp,
                                                   const Vec3<float> &pz, float min_match_sq_dis, const float4 *__restrict__ map,
                                                   const int *__restrict__ cells, const GridDesc &g, uint8_t *__restrict__ valid,
                                                   float4 *__restrict__ coef) {
  const int gt = block_x * blockDim.x + threadIdx.x;
  const int i = gt / LPQ, sub = gt % LPQ;
  const bool active = i < M;
  Quat<float> q(tp[3], tp[0], tp[1], tp[2]);
  Vec3<float> t(tp[4], tp[5], tp[6]);
  float4 po = active ? stack[i] : make_float4(0, 0, 0, 0);
  Vec3<float> r = rotate(q, Vec3<float>(po.x, po.y, po.z));
  Vec3<float> sel(r.x + t.x, r.y + t.y, r.z + t.z);
  float bd[5]; int bi[5], bj[5];
  knn_scan_group<5, LPQ>(sel, active, sub, map, cells, g, bd, bi, bj);
  if (!active || sub != 0) return;
  const int slot = slot_off + i;
  uint8_t ok = 0;
  float4 c = make_float4(0, 0, 0, 0);
  if (bi[4] != INT_MAX && bd[4] < min_match_sq_dis) {
    float nx[5], ny[5], nz[5];
    Vec3<float> vc(0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      float4 pn = map[bj[j]];
      nx[j] = pn.x; ny[j] = pn.y; nz[j] = pn.z;
      vc.x += pn.x; vc.y += pn.y; vc.z += pn.z;
    }
    vc.x /= 5.0f; vc.y /= 5.0f; vc.z /= 5.0f;
    float a00 = 0, a10 = 0, a20 = 0, a11 = 0, a21 = 0, a22 = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      float ax = nx[j] - vc.x, ay = ny[j] - vc.y, az = nz[j] - vc.z;
      a00 += ax * ax; a10 += ax * ay; a20 += ax * az; a11 += ay * ay; a21 += ay * az; a22 += az * az;
    }
    a00 /= 5.0f; a10 /= 5.0f; a20 /= 5.0f; a11 /= 5.0f; a21 /= 5.0f; a22 /= 5.0f;
    const float A1[9] = {a00, a10, a20, a10, a11, a21, a20, a21, a22};
    float D1[3]; double v[3];
    sym_eig3_top_closed(A1, D1, v);
    if (D1[2] > 3 * D1[1]) {
      const float x0 = sel.x, y0 = sel.y, z0 = sel.z;
      const float v0 = float(v[0]), v1 = float(v[1]), v2 = float(v[2]);  // mat_V1 is a float matrix
      const float x1 = float(double(vc.x) + 0.1 * double(v0)), y1 = float(double(vc.y) + 0.1 * double(v1)), z1 = float(double(vc.z) + 0.1 * double(v2));
      const float x2 = float(double(vc.x) - 0.1 * double(v0)), y2 = float(double(vc.y) - 0.1 * double(v1)), z2 = float(double(vc.z) - 0.1 * double(v2));
      Vec3<float> X0(x0, y0, z0), X1(x1, y1, z1), X2(x2, y2, z2);
      Vec3<float> a012v = cross(X0 - X1, X0 - X2);
      Vec3<float> nt = cross(X1 - X2, a012v);
      const float n2 = dot(nt, nt);
      if (n2 > 0.f) nt = nt / sqrtf(n2);
      const float a012 = norm(a012v), l12 = norm(X1 - X2);
      const float ld2 = a012 / l12;
      const float s = 1 - 0.9f * fabsf(ld2);
      float dx1 = t.x - sel.x, dy1 = t.y - sel.y, dz1 = t.z - sel.z;
      float side1 = dx1 * dx1 + dy1 * dy1 + dz1 * dz1;
      float dx2 = pz.x - sel.x, dy2 = pz.y - sel.y, dz2 = pz.z - sel.z;
      float side2 = dx2 * dx2 + dy2 * dy2 + dz2 * dz2;
      float check1 = 100.0f + side1 - side2 - 10.0f * sqrtf(3.0f) * sqrtf(side1);
      float check2 = 100.0f + side1 - side2 + 10.0f * sqrtf(3.0f) * sqrtf(side1);
      if (double(s) > 0.1 && check1 < 0 && check2 > 0) {
        ok = 1;
        c = make_float4(s * nt.x, s * nt.y, s * nt.z, s * ld2);
      }
    }
  }
  valid[slot] = ok; coef[slot] = c;
}

__global__ void __launch_bounds__(128) k_line_features(const float4 *__restrict__ stack, int M, int slot_off, const float *__restrict__ tp,
                                                      Vec3<float> pz, float min_match_sq_dis, const float4 *__restrict__ map,
                                                      const int *__restrict__ cells, GridDesc g, uint8_t *__restrict__ valid,
                                                      float4 *__restrict__ coef, const int *__restrict__ skip_flag) {
  if (skip_flag && *skip_flag) return;
  line_features_body(blockIdx.x, stack, M, slot_off, tp, pz, min_match_sq_dis, map, cells, g, valid, coef);
}

// One round of the scan-to-map search in ONE launch: blockIdx.y = 0 runs the corner (line) branch against the corner map,
// blockIdx.y = 1 the surf (plane) branch against the surf map.  No cross-stream events, one dispatch.
struct MapRoundArgs {
  const float4 *corner_stack; int Mc;
  const float4 *corner_map; const int *corner_cells; GridDesc corner_grid;
  const float4 *surf_map; const int *surf_cells; GridDesc surf_grid;
  int blocks_corner, blocks_surf;
};
__global__ void __launch_bounds__(128) k_map_round(FeatArgs a, MapRoundArgs m, const float *__restrict__ tp, uint8_t *__restrict__ valid,
                                                  float4 *__restrict__ coef, float4 *__restrict__ abs_coef, const int *__restrict__ skip_flag) {
  if (skip_flag && *skip_flag) return;
  if (blockIdx.y == 0) {
    if (int(blockIdx.x) >= m.blocks_corner) return;
    line_features_body(blockIdx.x, m.corner_stack, m.Mc, 0, tp, Vec3<float>(a.fixed_pz[0], a.fixed_pz[1], a.fixed_pz[2]), a.min_match_sq_dis,
                       m.corner_map, m.corner_cells, m.corner_grid, valid, coef);
  } else {
    if (int(blockIdx.x) >= m.blocks_surf) return;
    features_body<true, 8>(a.fr[0], feat_scalars(a), blockIdx.x, tp, m.surf_map, m.surf_cells, m.surf_grid, valid, coef, nullptr, abs_coef);
  }
}

void launch_map_round(const FeatArgs &surf, const float4 *corner_stack, int Mc, const float *transform, const float4 *corner_map,
                      const int *corner_cells, const GridDesc &corner_grid, const float4 *surf_map, const int *surf_cells,
                      const GridDesc &surf_grid, uint8_t *valid, float4 *coef, float4 *abs_coef, const int *skip_flag, hipStream_t s) {
  MapRoundArgs m{corner_stack, Mc, corner_map, corner_cells, corner_grid, surf_map, surf_cells, surf_grid, cdiv((long long)Mc * FEAT_LPQ, 128),
                 cdiv((long long)surf.max_M * 8, 128)};
  const int bx = std::max(1, std::max(m.blocks_corner, m.blocks_surf));
  hipLaunchKernelGGL(k_map_round, dim3(bx, 2), dim3(128), 0, s, surf, m, transform, valid, coef, abs_coef, skip_flag);
  LIO_HIP(hipGetLastError());
}

void launch_line_features(const float4 *stack, int M, int slot_off, const float *transform, const float fixed_pz[3], float min_match_sq_dis,
                          const float4 *map_sorted, const int *cells, const GridDesc &g, uint8_t *valid, float4 *coef,
                          const int *skip_flag, hipStream_t s) {
  if (M <= 0) return;
  hipLaunchKernelGGL(k_line_features, dim3(cdiv((long long)M * FEAT_LPQ, 128)), dim3(128), 0, s, stack, M, slot_off, transform,
                     Vec3<float>(fixed_pz[0], fixed_pz[1], fixed_pz[2]), min_match_sq_dis, map_sorted, cells, g, valid, coef, skip_flag);
  LIO_HIP(hipGetLastError());
}

void launch_features(const FeatArgs &a, const float *transforms, const float4 *map_sorted, const int *cells, const GridDesc &g,
                     uint8_t *valid, float4 *coef, float *score, const int *skip_flag, hipStream_t s, float4 *abs_coef) {
  if (a.nframes <= 0 || a.max_M <= 0) return;
  // lanes per query: 8 when the launch is small (latency-bound: shorter per-lane candidate walks), 4 when it already fills
  // the GPU several times over (throughput-bound: fewer shuffle-merge rounds per query)
  const bool big = (long long)a.max_M * a.nframes >= 50000;
  const dim3 grid(cdiv((long long)a.max_M * (big ? 4 : 8), FEAT_THREADS), a.nframes);
  if (a.mapping_mode)
    hipLaunchKernelGGL((k_features<true, 8>), dim3(cdiv((long long)a.max_M * 8, FEAT_THREADS), a.nframes), dim3(FEAT_THREADS), 0, s, a, transforms, map_sorted,
                       cells, g, valid, coef, score, skip_flag, abs_coef);
  else if (big)
    hipLaunchKernelGGL((k_features<false, 4>), grid, dim3(FEAT_THREADS), 0, s, a, transforms, map_sorted, cells, g, valid, coef, score, skip_flag, abs_coef);
  else
    hipLaunchKernelGGL((k_features<false, 8>), grid, dim3(FEAT_THREADS), 0, s, a, transforms, map_sorted, cells, g, valid, coef, score, skip_flag, abs_coef);
  LIO_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// CalculateLaserOdom: rows of (mat_A | mat_B), reduced; then the 6x6 step
// ------------------------------------------------------------------------------------------------
#define ODOM_ROW_THREADS 256
int odom_rows_blocks(int nslots) { return std::max(1, std::min(cdiv(nslots, ODOM_ROW_THREADS * 2), 256)); }

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

__device__ __forceinline__ void odom_rows_body(int block_x, int nblocks, const float4 *__restrict__ stack, int M, int nslots,
                                               const uint8_t *__restrict__ valid, const float4 *__restrict__ coef,
                                               const OdomState *__restrict__ st, double *__restrict__ partials, int b_from_coef) {
  Quat<float> q(st->T[3], st->T[0], st->T[1], st->T[2]);
  Vec3<float> t(st->T[4], st->T[5], st->T[6]);
  Mat3<float> Rm = toRot(q);
  Mat3<float> Rinv = toRot(qinverse(q));
  double acc[28];
#pragma unroll
  for (int k = 0; k < 28; ++k) acc[k] = 0;
  for (int sidx = block_x * blockDim.x + threadIdx.x; sidx < nslots; sidx += nblocks * blockDim.x) {
    if (!valid[sidx]) continue;
    odom_row_accumulate(stack[sidx % M], coef[sidx], q, t, Rm, Rinv, b_from_coef, acc);
  }
  // wave reduce (64 lanes) then cross-wave through LDS
  __shared__ double sm[ODOM_ROW_THREADS / 64][28];
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
    for (int w = 0; w < ODOM_ROW_THREADS / 64; ++w) v += sm[w][threadIdx.x];
    partials[block_x * 28 + threadIdx.x] = v;
  }
}

__global__ void __launch_bounds__(ODOM_ROW_THREADS) k_odom_rows(const float4 *__restrict__ stack, int M, int nslots,
                                                                const uint8_t *__restrict__ valid, const float4 *__restrict__ coef,
                                                                const OdomState *__restrict__ st, double *__restrict__ partials,
                                                                int b_from_coef) {
  if (st->converged) return;
  odom_rows_body(blockIdx.x, gridDim.x, stack, M, nslots, valid, coef, st, partials, b_from_coef);
}

void launch_odom_rows(const float4 *stack, int M, int nslots, const uint8_t *valid, const float4 *coef, const OdomState *st,
                      double *partials, int nblocks, hipStream_t s, int b_from_coef) {
  hipLaunchKernelGGL(k_odom_rows, dim3(nblocks), dim3(ODOM_ROW_THREADS), 0, s, stack, M, nslots, valid, coef, st, partials, b_from_coef);
  LIO_HIP(hipGetLastError());
}

__device__ __forceinline__ void odom_update_from_sums(const double *ssum, OdomState *st, int iter, int min_rows, int left_update);
__device__ __forceinline__ void odom_update_body(const double *__restrict__ partials, int nblocks, OdomState *st, int iter, int min_rows,
                                                 int left_update) {
  // column k of the partials is summed by lane k (fixed order), then lane 0 runs the scalar 6x6 step
  __shared__ double ssum[28];
  reduce_partials28(partials, nblocks, ssum);
  odom_update_from_sums(ssum, st, iter, min_rows, left_update);
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

__global__ void k_odom_update(const double *__restrict__ partials, int nblocks, OdomState *st, int iter, int min_rows, int left_update, OdomState *mail,
                              HostSignal sig) {
  if (!st->converged) odom_update_body(partials, nblocks, st, iter, min_rows, left_update);
  if (sig.flag) {   // the rounds at which the host looks at the convergence flag post the state to its mailbox (dev.h)
    __syncthreads();
    if (threadIdx.x < 64) post_host_mail(sig, mail, st, int(sizeof(OdomState) / 4), threadIdx.x);
  }
}

void launch_odom_update(const double *partials, int nblocks, OdomState *st, int iter, hipStream_t s, int min_rows, int left_update, OdomState *mail,
                        const HostSignal &sig) {
  hipLaunchKernelGGL(k_odom_update, dim3(1), dim3(256), 0, s, partials, nblocks, st, iter, min_rows, left_update, mail, sig);
  LIO_HIP(hipGetLastError());
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
template <int LPQ>
__device__ __forceinline__ double odom_round_block(const FeatArgs &a, FeatFrame fr, const Quat<float> q, const Vec3<float> t, const float4 *__restrict__ map,
                                                   const int *__restrict__ cells, const GridDesc &g, uint8_t *__restrict__ valid, float4 *__restrict__ coef,
                                                   float *__restrict__ score, int base_slot, int round, int keep, int block) {
  constexpr int QPB = ODOM_ROUND_THREADS / LPQ;   // queries per block
  static_assert(QPB <= 64, "the fit phase is one wave");
  __shared__ int s_bj[QPB][5];
  __shared__ float s_bd4[QPB];
  __shared__ int s_bi4[QPB];
  __shared__ double rows[QPB][29];
  const int M = fr.M;
  fr.slot_off = base_slot + (keep ? round * M : 0);
  const FeatScalars fs = feat_scalars(a);
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
    if (ql < QPB) {
#pragma unroll
      for (int k = 0; k < 28; ++k) rows[ql][k] = acc[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (threadIdx.x < 28) {
#pragma unroll
      for (int qq = 0; qq < QPB; ++qq) v += rows[qq][threadIdx.x];
    }
  }
  return v;
}

template <int LPQ>
__global__ void __launch_bounds__(ODOM_ROUND_THREADS) k_odom_round(FeatArgs a, const OdomState *__restrict__ st, const float4 *__restrict__ map,
                                                                  const int *__restrict__ cells, GridDesc g, uint8_t *__restrict__ valid,
                                                                  float4 *__restrict__ coef, float *__restrict__ score, double *__restrict__ partials,
                                                                  int base_slot, int round, int keep) {
  if (st->converged) return;
  const float *tp = st->T;
  const Quat<float> q(tp[3], tp[0], tp[1], tp[2]);
  const Vec3<float> t(tp[4], tp[5], tp[6]);
  const double v = odom_round_block<LPQ>(a, a.fr[0], q, t, map, cells, g, valid, coef, score, base_slot, round, keep, int(blockIdx.x));
  if (threadIdx.x < 28) partials[size_t(blockIdx.x) * 28 + threadIdx.x] = v;
}

// ------------------------------------------------------------------------------------------------
// All rounds of the newest frame's loop in ONE launch.  Grid = the search blocks of k_odom_round + one update
// block.  A round: every search block computes its 28 sums at the current transform, parks them in HBM (agent-scope stores,
// acknowledged) and raises its flag to the round's number; the update block waits for all flags, folds the partials in
// k_odom_update_wide's order, takes the 6x6 step on its copy of the state, republishes the state and then the round number the
// search blocks are waiting for.  After convergence or the last round the update block posts the state to the host's mailbox.
// No atomics, no fences: flags and data travel as agent-scope stores / loads, the writer waits for its data to be acknowledged
// before it raises the flag.  Every waiter gives up after `timeout_ticks` of the wall clock (a dead peer must not hang the GPU).
struct OdomRoundsCtl {
  double *partials;          // nb x 28
  unsigned *block_flag;      // nb
  unsigned *state_seq;       // 1: number of the round whose INPUT state is published
  unsigned seq0;             // number of round 0 of this launch (monotonic over the life of the handle)
  int max_rounds;
  long long timeout_ticks;
  long long *stamps;         // optional (LIO_DEBUG_TIMING): wall clock of the update block at its start, [1 + 2 r] all flags of round r in, [2 + 2 r] state republished
};
template <int LPQ>
__global__ void __launch_bounds__(ODOM_ROUND_THREADS) k_odom_rounds_resident(FeatArgs a, OdomState *st, const float4 *__restrict__ map, const int *__restrict__ cells,
                                                                            GridDesc g, uint8_t *__restrict__ valid, float4 *__restrict__ coef,
                                                                            float *__restrict__ score, int base_slot, int keep, OdomRoundsCtl ctl, OdomState *mail,
                                                                            HostSignal sig) {
  const int nb = int(gridDim.x) - 1;
  const int tid = threadIdx.x, lane = tid & 63;
  __shared__ OdomState s_st;
  __shared__ int s_go;
  unsigned *su = reinterpret_cast<unsigned *>(&s_st);
  constexpr int NW = int(sizeof(OdomState) / 4);
  if (int(blockIdx.x) < nb) {
    // ---------------- search block
    for (int round = 0; round < ctl.max_rounds; ++round) {
      const unsigned seq = ctl.seq0 + unsigned(round);
      if (tid < 64) {
        int ok = 1;
        if (round > 0) {   // round 0's state was uploaded in front of the launch
          const long long t0 = wall_clock64();
          while (__hip_atomic_load(ctl.state_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != seq) {
            if (wall_clock64() - t0 > ctl.timeout_ticks) { ok = 0; break; }
            __builtin_amdgcn_s_sleep(1);
          }
        }
        if (lane < NW) su[lane] = __hip_atomic_load(reinterpret_cast<unsigned *>(st) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0) s_go = ok;
      }
      __syncthreads();
      if (!s_go || s_st.converged) return;
      const Quat<float> q(s_st.T[3], s_st.T[0], s_st.T[1], s_st.T[2]);
      const Vec3<float> t(s_st.T[4], s_st.T[5], s_st.T[6]);
      const double v = odom_round_block<LPQ>(a, a.fr[0], q, t, map, cells, g, valid, coef, score, base_slot, round, keep, int(blockIdx.x));
      if (tid < 64) {
        if (tid < 28) __hip_atomic_store(ctl.partials + size_t(blockIdx.x) * 28 + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) __hip_atomic_store(ctl.block_flag + blockIdx.x, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();   // the next round overwrites s_st and the LDS tables of odom_round_block
    }
    return;
  }
  // ---------------- update block
  __shared__ double part[32][32];
  __shared__ double ssum[28];
  if (tid < NW) su[tid] = reinterpret_cast<const unsigned *>(st)[tid];   // uploaded in front of the launch
  if (ctl.stamps && tid == 0) ctl.stamps[0] = wall_clock64();
  __syncthreads();
  int round = 0;
  for (; round < ctl.max_rounds && !s_st.converged; ++round) {
    const unsigned seq = ctl.seq0 + unsigned(round);
    if (tid < 64) {
      const long long t0 = wall_clock64();
      int ok = 1;
      for (;;) {
        bool all = true;
        for (int b0 = 0; b0 < nb; b0 += 64) {
          const int b = b0 + lane;
          const unsigned f = b < nb ? __hip_atomic_load(ctl.block_flag + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : seq;
          all = all && __all(f == seq);
        }
        if (all) break;
        if (wall_clock64() - t0 > ctl.timeout_ticks) { ok = 0; break; }
      }
      if (lane == 0) s_go = ok;
    }
    __syncthreads();
    if (!s_go) break;
    if (ctl.stamps && tid == 0) ctl.stamps[1 + 2 * round] = wall_clock64();
    // fold in k_odom_update_wide's order: 32 groups g of rows b = g, g + 32, ..., each as four interleaved chains combined
    // (v0 + v1) + (v2 + v3), then the group sums in ascending g.  256 threads stand in for its 1024: thread (c, g0) takes the
    // groups g0, g0 + 8, g0 + 16, g0 + 24 one after the other.
    {
      const int c = tid & 31, g0 = tid >> 5;
      for (int gq = g0; gq < 32; gq += 8) {
        double v0 = 0, v1 = 0, v2 = 0, v3 = 0;
        if (c < 28) {
          int b = gq;
          for (; b + 96 < nb; b += 128) {
            const double x0 = __hip_atomic_load(ctl.partials + size_t(b) * 28 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double x1 = __hip_atomic_load(ctl.partials + size_t(b + 32) * 28 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double x2 = __hip_atomic_load(ctl.partials + size_t(b + 64) * 28 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double x3 = __hip_atomic_load(ctl.partials + size_t(b + 96) * 28 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v0 += x0; v1 += x1; v2 += x2; v3 += x3;
          }
          for (; b < nb; b += 32) v0 += __hip_atomic_load(ctl.partials + size_t(b) * 28 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        part[gq][c] = (v0 + v1) + (v2 + v3);
      }
    }
    __syncthreads();
    if (tid < 28) {
      double s2 = 0;
#pragma unroll
      for (int k = 0; k < 32; ++k) s2 += part[k][tid];
      ssum[tid] = s2;
    }
    __syncthreads();
    odom_update_from_sums(ssum, &s_st, round, 0, 0);   // thread 0, on the LDS copy
    __syncthreads();
    // republish: the state first, acknowledged, then the number of the round that may start from it
    if (tid < 64) {
      if (lane < NW) __hip_atomic_store(reinterpret_cast<unsigned *>(st) + lane, su[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(ctl.state_seq, seq + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (ctl.stamps && lane == 0) ctl.stamps[2 + 2 * round] = wall_clock64();
    }
    __syncthreads();
  }
  // the loop ended by convergence, by the round limit or by a timeout: release waiting search blocks (a converged / final state is
  // already published; after a timeout nothing more can be done for them: they give up on their own), then tell the host
  if (sig.flag && tid < 64) post_host_mail(sig, mail, &s_st, NW, tid);
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
__global__ void __launch_bounds__(1024) k_odom_update_wide(const double *__restrict__ partials, int nblocks, OdomState *st, int iter, int min_rows,
                                                           int left_update, OdomState *mail, HostSignal sig) {
  odom_update_wide_block(partials, nblocks, st, iter, min_rows, left_update, mail, sig);
}
// The update block with a share of the older frames' features riding in the same launch (round 4).  The update is ONE block: for its
// 12 us the chip is idle (kernel trace, profiles/r4_solve_timeline.md), while the batched k_features on a side stream used to run
// beside round 0's search kernel and slowed it from 19 to 37 us (both are bound by vector issue).  Block 0 is the update; the blocks
// behind it are 1024 threads wide and take four 256-lane feature blocks each (features_block<.., 4>), a frame's blocks contiguous.
// No second stream, no fork / join events (an event between two kernels of a stream costs a 7 us bubble).  Same results: the
// features do not depend on the split, and they finish before the launch does, i.e. before anything later on the stream.
template <int LPQ>
__global__ void __launch_bounds__(1024) k_odom_update_with_features(const double *__restrict__ partials, int nblocks, OdomState *st, int iter, int min_rows,
                                                                    int left_update, OdomState *mail, HostSignal sig, FeatArgs af,
                                                                    const float *__restrict__ transforms, const float4 *__restrict__ map,
                                                                    const int *__restrict__ cells, GridDesc g, uint8_t *__restrict__ valid,
                                                                    float4 *__restrict__ coef, float *__restrict__ score, int fblocks) {
  if (blockIdx.x == 0) { odom_update_wide_block(partials, nblocks, st, iter, min_rows, left_update, mail, sig); return; }
  const int fb = int(blockIdx.x) - 1;
  features_block<false, LPQ, 4>(af, fb / fblocks, fb % fblocks, transforms, map, cells, g, valid, coef, score, nullptr);
}

__global__ void __launch_bounds__(256) k_solve_setup(SolveSetup a, float *__restrict__ d_transforms, OdomState *__restrict__ d_odom, uint8_t *__restrict__ valid,
                                                     size_t n_valid) {
  const size_t i = (size_t(blockIdx.x) * 256 + threadIdx.x) * 16;
  if (i + 16 <= n_valid) *reinterpret_cast<uint4 *>(valid + i) = make_uint4(0, 0, 0, 0);
  else for (size_t k = i; k < n_valid; ++k) valid[k] = 0;
  if (blockIdx.x == 0) {
    for (int k = threadIdx.x; k < a.ntf * 8; k += 256) d_transforms[k] = a.tf[k >> 3][k & 7];
    if (a.set_odom) {
      unsigned *o = reinterpret_cast<unsigned *>(d_odom);
      const int nw = int(sizeof(OdomState) / 4);
      if (int(threadIdx.x) < nw) o[threadIdx.x] = threadIdx.x < 8 ? __float_as_uint(a.odom_T[threadIdx.x]) : 0u;
    }
  }
}
void launch_solve_setup(const SolveSetup &a, float *d_transforms, OdomState *d_odom, uint8_t *valid, size_t n_valid, hipStream_t s) {
  static_assert(offsetof(OdomState, T) == 0 && sizeof(OdomState) <= 256 * 4, "state layout: T first, the rest zero");
  const int nb = std::max(1, cdiv((long long)n_valid, 256 * 16));
  hipLaunchKernelGGL(k_solve_setup, dim3(nb), dim3(256), 0, s, a, d_transforms, d_odom, valid, n_valid);
  LIO_HIP(hipGetLastError());
}

int odom_round_blocks(int M, int lpq) { return std::max(1, cdiv((long long)M * lpq, ODOM_ROUND_THREADS)); }
void launch_odom_rounds_resident(const FeatArgs &a, int base_slot, int keep, int max_rounds, OdomState *st, const float4 *map_sorted, const int *cells,
                                 const GridDesc &g, uint8_t *valid, float4 *coef, float *score, double *partials, unsigned *block_flag, unsigned *state_seq,
                                 unsigned seq0, long long timeout_ticks, hipStream_t s, OdomState *mail, const HostSignal &sig, long long *stamps, int lpq) {
  const int M = a.fr[0].M;
  if (M <= 0) return;
  const int nb = odom_round_blocks(M, lpq);
  OdomRoundsCtl ctl{partials, block_flag, state_seq, seq0, max_rounds, timeout_ticks, stamps};
  if (lpq == 4)
    hipLaunchKernelGGL(k_odom_rounds_resident<4>, dim3(nb + 1), dim3(ODOM_ROUND_THREADS), 0, s, a, st, map_sorted, cells, g, valid, coef, score, base_slot, keep, ctl,
                       mail, sig);
  else
  hipLaunchKernelGGL(k_odom_rounds_resident<8>, dim3(nb + 1), dim3(ODOM_ROUND_THREADS), 0, s, a, st, map_sorted, cells, g, valid, coef, score, base_slot, keep, ctl,
                     mail, sig);
  LIO_HIP(hipGetLastError());
}
void launch_odom_round(const FeatArgs &a, int base_slot, int round, int keep, OdomState *st, const float4 *map_sorted, const int *cells, const GridDesc &g,
                       uint8_t *valid, float4 *coef, float *score, double *partials, hipStream_t s, OdomState *mail, const HostSignal &sig, int lpq,
                       hipEvent_t after_search, const FeatArgs *ride, const float *transforms) {
  const int M = a.fr[0].M;
  if (M <= 0) return;
  const int nb = odom_round_blocks(M, lpq);
  if (lpq == 4)
    hipLaunchKernelGGL(k_odom_round<4>, dim3(nb), dim3(ODOM_ROUND_THREADS), 0, s, a, st, map_sorted, cells, g, valid, coef, score, partials, base_slot, round, keep);
  else
  hipLaunchKernelGGL(k_odom_round<8>, dim3(nb), dim3(ODOM_ROUND_THREADS), 0, s, a, st, map_sorted, cells, g, valid, coef, score, partials, base_slot, round, keep);
  if (after_search) LIO_HIP(hipEventRecord(after_search, s));   // the one-block update kernel behind it leaves the chip idle: work of another stream can start here
  if (ride && ride->nframes > 0 && ride->max_M > 0 && !ride->mapping_mode) {
    // the older frames' share of this round rides with the update block: four lanes per query, 64 queries per 256-lane quarter
    const int fblocks = cdiv(cdiv((long long)ride->max_M * 4, FEAT_THREADS), 4);
    hipLaunchKernelGGL(k_odom_update_with_features<4>, dim3(1 + fblocks * ride->nframes), dim3(1024), 0, s, partials, nb, st, round, 0, 0, mail, sig, *ride,
                       transforms, map_sorted, cells, g, valid, coef, score, fblocks);
  } else
  hipLaunchKernelGGL(k_odom_update_wide, dim3(1), dim3(1024), 0, s, partials, nb, st, round, 0, 0, mail, sig);
  LIO_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// Batched keyframe refinement: B independent scan-to-map loops advance together, one launch per stage per round
// ------------------------------------------------------------------------------------------------
template <int LPQ>
__device__ __forceinline__ void kf_round_body(const KfDesc *__restrict__ kd, const KfMapDesc *__restrict__ md, const OdomState *__restrict__ st,
                                                 const float4 *__restrict__ stack_all, float min_match_sq_dis, float min_plane_dis, int mapping_mode,
                                                 uint8_t *__restrict__ valid, float4 *__restrict__ coef) {
  const int k = blockIdx.z;
  if (st[k].converged) return;
  const KfDesc d = kd[k];
  const float *tp = st[k].T;
  if (blockIdx.y == 0) {
    if (int(blockIdx.x) * 128 >= d.Mc * LPQ) return;
    const KfMapDesc &m = md[d.map];
    line_features_body<LPQ>(blockIdx.x, stack_all + d.slot_off, d.Mc, d.slot_off, tp, Vec3<float>(d.pz[0], d.pz[1], d.pz[2]), min_match_sq_dis, m.corner_sorted,
                       m.corner_cells, m.corner_grid, valid, coef);
  } else {
    if (int(blockIdx.x) * 128 >= d.Ms * LPQ) return;
    const KfMapDesc &m = md[d.map];
    const FeatFrame fr{stack_all + d.slot_off + d.Mc, d.Ms, d.slot_off + d.Mc, 0};
    const FeatScalars fs{min_match_sq_dis, min_plane_dis, mapping_mode, {d.pz[0], d.pz[1], d.pz[2]}};
    features_body<true, LPQ>(fr, fs, blockIdx.x, tp, m.surf_sorted, m.surf_cells, m.surf_grid, valid, coef, nullptr, nullptr);
  }
}

template <int LPQ>
__global__ void __launch_bounds__(128) k_kf_round(const KfDesc *__restrict__ kd, const KfMapDesc *__restrict__ md, const OdomState *__restrict__ st,
                                                 const float4 *__restrict__ stack_all, float min_match_sq_dis, float min_plane_dis, int mapping_mode,
                                                 uint8_t *__restrict__ valid, float4 *__restrict__ coef) {
  kf_round_body<LPQ>(kd, md, st, stack_all, min_match_sq_dis, min_plane_dis, mapping_mode, valid, coef);
}
#define KF_OCC_VARIANT(W)                                                                                                                      \
  __global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(W, W)))                                                            \
  k_kf_round1_w##W(const KfDesc *__restrict__ kd, const KfMapDesc *__restrict__ md, const OdomState *__restrict__ st,                          \
                   const float4 *__restrict__ stack_all, float min_match_sq_dis, float min_plane_dis, int mapping_mode,                        \
                   uint8_t *__restrict__ valid, float4 *__restrict__ coef) {                                                                   \
    kf_round_body<1>(kd, md, st, stack_all, min_match_sq_dis, min_plane_dis, mapping_mode, valid, coef);                                       \
  }
// the one-lane-per-query form is bound by gather latency: 8 waves per SIMD (64 VGPRs, the fit phase spills a little) beats
// the 5 waves the default allocation gives by 15% (54.4 -> 46.1 ms at 1000 HDL-64 keyframes)
KF_OCC_VARIANT(8)

__global__ void __launch_bounds__(ODOM_ROW_THREADS) k_kf_rows(const KfDesc *__restrict__ kd, const OdomState *__restrict__ st,
                                                              const float4 *__restrict__ stack_all, const uint8_t *__restrict__ valid,
                                                              const float4 *__restrict__ coef, double *__restrict__ partials, int b_from_coef) {
  const int k = blockIdx.y;
  if (st[k].converged) return;
  const KfDesc d = kd[k];
  if (int(blockIdx.x) >= d.nb) return;
  const int M = d.Mc + d.Ms;
  odom_rows_body(blockIdx.x, d.nb, stack_all + d.slot_off, M, M, valid + d.slot_off, coef + d.slot_off, st + k, partials + size_t(d.part_off) * 28,
                 b_from_coef);
}

__global__ void k_kf_update(const KfDesc *__restrict__ kd, OdomState *st, const double *__restrict__ partials, int iter, int min_rows, int left_update,
                            int *n_converged) {
  const int k = blockIdx.x;
  if (st[k].converged) return;
  const KfDesc d = kd[k];
  odom_update_body(partials + size_t(d.part_off) * 28, d.nb, st + k, iter, min_rows, left_update);
  if (threadIdx.x == 0 && st[k].converged) atomicAdd(n_converged, 1);
}

void launch_kf_round(const KfDesc *kd, const KfMapDesc *md, const OdomState *st, int n_keyframes, int max_Mc, int max_Ms, long long total_queries,
                     const float4 *stack_all, float min_match_sq_dis, float min_plane_dis, int mapping_mode, uint8_t *valid, float4 *coef, hipStream_t s) {
  // lanes per query: a small batch is latency-bound (8 lanes shorten each query's dependent candidate walk); once the batch
  // fills the GPU many times over, one lane per query wins 1.8x (no merge rounds, no idle lanes in the fit): measured
  // 96 / 69 / 59 / 55 ms for 8 / 4 / 2 / 1 lanes at 1000 HDL-64 keyframes.  The result does not depend on the split.
  static const int lpq_env = [] { const char *e = std::getenv("LIO_KF_LPQ"); return e ? std::atoi(e) : 0; }();
  const int lpq = lpq_env ? lpq_env : (total_queries >= 400000 ? 1 : total_queries >= 60000 ? 4 : 8);
  const int bx = std::max(1, cdiv((long long)std::max(max_Mc, max_Ms) * lpq, 128));
  const dim3 grid(bx, 2, n_keyframes);
#define KF_ROUND(L) hipLaunchKernelGGL(k_kf_round<L>, grid, dim3(128), 0, s, kd, md, st, stack_all, min_match_sq_dis, min_plane_dis, mapping_mode, valid, coef)
  if (lpq == 1) hipLaunchKernelGGL(k_kf_round1_w8, grid, dim3(128), 0, s, kd, md, st, stack_all, min_match_sq_dis, min_plane_dis, mapping_mode, valid, coef);
  else if (lpq == 2) KF_ROUND(2); else if (lpq == 4) KF_ROUND(4); else KF_ROUND(8);
#undef KF_ROUND
  LIO_HIP(hipGetLastError());
}
void launch_kf_rows(const KfDesc *kd, const OdomState *st, int n_keyframes, int max_nb, const float4 *stack_all, const uint8_t *valid, const float4 *coef,
                    double *partials, int b_from_coef, hipStream_t s) {
  hipLaunchKernelGGL(k_kf_rows, dim3(max_nb, n_keyframes), dim3(ODOM_ROW_THREADS), 0, s, kd, st, stack_all, valid, coef, partials, b_from_coef);
  LIO_HIP(hipGetLastError());
}
void launch_kf_update(const KfDesc *kd, OdomState *st, int n_keyframes, const double *partials, int iter, int min_rows, int left_update, int *n_converged,
                      hipStream_t s) {
  hipLaunchKernelGGL(k_kf_update, dim3(n_keyframes), dim3(256), 0, s, kd, st, partials, iter, min_rows, left_update, n_converged);
  LIO_HIP(hipGetLastError());
}

}  // namespace lio
