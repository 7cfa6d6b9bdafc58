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
#include "cloud_device.h"

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

// launch() enqueues the whole filter on `s` (no host sync); finish() waits for it and returns the output count.  Two
// filters launched on two streams overlap (the scan-to-map step filters its corner and surf stacks that way).
void VoxelGridDev::launch(const float4 *in, size_t n, float leaf, DBuf<float4> &out, hipStream_t s) {
  p_in_ = in; p_n_ = n; p_out_ = &out; p_stream_ = s; p_leaf_ = leaf;
  if (n == 0) return;
  enqueue(false);
}

// exact == false: absolute-cell keys, bounds folded on the side (4 stages: keys, sort, tile heads, centroids);
// exact == true: PCL's own index from the bounds (two more launches in front), used when the cloud leaves the key's range.
void VoxelGridDev::enqueue(bool exact) {
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
  LIO_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_.p, keys2_.p, vals_.p, vals2_.p, n, 0, 32, s));
  tmp_.reserve(tmp_bytes + 256);
  LIO_HIP(rocprim::radix_sort_pairs(tmp_.p, tmp_bytes, keys_.p, keys2_.p, vals_.p, vals2_.p, n, 0, 32, s));
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
  bool redone = false;
  for (;;) {
    if (sig_.flag) wait_host_signal(sig_, p_stream_);   // the count is out; the centroids follow in stream order
    else LIO_HIP(hipStreamSynchronize(p_stream_));
    if (m->range_overflow == 1 && !redone) {
      enqueue(true);   // the cloud spans more cells than the absolute key holds
      redone = true;
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
  if (h_count_) (void)hipHostFree(h_count_);
}

// ------------------------------------------------------------------------------------------------
// K-NN grid
// ------------------------------------------------------------------------------------------------
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

// CalculateFeatures for every frame of the launch (blockIdx.y)
template <bool MAPPING, int LPQ>
__global__ void __launch_bounds__(FEAT_THREADS) k_features(FeatArgs a, const float *__restrict__ transforms, const float4 *__restrict__ map,
                                                          const int *__restrict__ cells, GridDesc g, uint8_t *__restrict__ valid,
                                                          float4 *__restrict__ coef, float *__restrict__ score, const int *__restrict__ skip_flag,
                                                          float4 *__restrict__ abs_coef) {
  if (skip_flag && *skip_flag) return;
  features_block<MAPPING, LPQ>(a.fr[blockIdx.y], feat_scalars(a), int(blockIdx.x), transforms, map, cells, g, valid, coef, score, abs_coef);
}

// Corner branch of the scan-to-map step: one query per FEAT_LPQ lanes, 5-NN, covariance of the 5 neighbours, line
// direction = eigenvector of the largest eigenvalue (accepted when it dominates 3x the middle one).
template <int LPQ = FEAT_LPQ>
__device__ __forceinline__ void line_features_body(int block_x, const float4 *__restrict__ stack, int M, int slot_off, const float *__restrict__ tp,
                                                   const Vec3<float> &pz, float min_match_sq_dis, const float4 *__restrict__ map,
                                                   const int *__restrict__ cells, const GridDesc &g, uint8_t *__restrict__ valid,
                                                   float4 *__restrict__ coef, const uint32_t *__restrict__ order = nullptr) {
  const int gt = block_x * blockDim.x + threadIdx.x;
  const int it = gt / LPQ, sub = gt % LPQ;
  const bool active = it < M;
  const int i = (active && order) ? int(order[it]) - slot_off : it;   // (processing order: FeatFrame::order)
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
// eight slots per thread: the 28 running sums of a wave meet through 28 x 6 shuffle steps of doubles, which at two slots per thread cost more
// than the rows themselves (k_kf_rows: 467 us per round of 1000 keyframes; round 6)
int odom_rows_blocks(int nslots) { return std::max(1, std::min(cdiv(nslots, ODOM_ROW_THREADS * 8), 256)); }

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

__device__ __forceinline__ void odom_update_body(const double *__restrict__ partials, int nblocks, OdomState *st, int iter, int min_rows,
                                                 int left_update) {
  // column k of the partials is summed by lane k (fixed order), then lane 0 runs the scalar 6x6 step
  __shared__ double ssum[28];
  reduce_partials28(partials, nblocks, ssum);
  odom_update_from_sums(ssum, st, iter, min_rows, left_update);
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

template <int LPQ>
__global__ void __launch_bounds__(ODOM_ROUND_THREADS) k_odom_round(FeatArgs a, const OdomState *__restrict__ st, const float4 *__restrict__ map,
                                                                  const int *__restrict__ cells, GridDesc g, uint8_t *__restrict__ valid,
                                                                  float4 *__restrict__ coef, float *__restrict__ score, double *__restrict__ partials,
                                                                  int base_slot, int round, int keep) {
  if (st->converged) return;
  const float *tp = st->T;
  const Quat<float> q(tp[3], tp[0], tp[1], tp[2]);
  const Vec3<float> t(tp[4], tp[5], tp[6]);
  const double v = odom_round_block<LPQ>(feat_scalars(a), a.fr[0], q, t, map, cells, g, valid, coef, score, base_slot, round, keep, int(blockIdx.x));
  if (threadIdx.x < 28) partials[size_t(blockIdx.x) * 28 + threadIdx.x] = v;
}

__global__ void __launch_bounds__(1024) k_odom_update_wide(const double *__restrict__ partials, int nblocks, OdomState *st, int iter, int min_rows,
                                                           int left_update, OdomState *mail, HostSignal sig) {
  odom_update_wide_block(partials, nblocks, st, iter, min_rows, left_update, mail, sig);
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
void launch_odom_round(const FeatArgs &a, int base_slot, int round, int keep, OdomState *st, const float4 *map_sorted, const int *cells, const GridDesc &g,
                       uint8_t *valid, float4 *coef, float *score, double *partials, hipStream_t s, OdomState *mail, const HostSignal &sig, int lpq) {
  const int M = a.fr[0].M;
  if (M <= 0) return;
  const int nb = odom_round_blocks(M, lpq);
  if (lpq == 4)
    hipLaunchKernelGGL(k_odom_round<4>, dim3(nb), dim3(ODOM_ROUND_THREADS), 0, s, a, st, map_sorted, cells, g, valid, coef, score, partials, base_slot, round, keep);
  else
    hipLaunchKernelGGL(k_odom_round<8>, dim3(nb), dim3(ODOM_ROUND_THREADS), 0, s, a, st, map_sorted, cells, g, valid, coef, score, partials, base_slot, round, keep);
  hipLaunchKernelGGL(k_odom_update_wide, dim3(1), dim3(1024), 0, s, partials, nb, st, round, 0, 0, mail, sig);
  LIO_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// Batched keyframe refinement: B independent scan-to-map loops advance together, one launch per stage per round
// ------------------------------------------------------------------------------------------------
template <int LPQ>
__device__ __forceinline__ void kf_round_body(const KfDesc *__restrict__ kd, const KfMapDesc *__restrict__ md, const OdomState *__restrict__ st,
                                                 const float4 *__restrict__ stack_all, const uint32_t *__restrict__ order, float min_match_sq_dis, float min_plane_dis,
                                                 int mapping_mode, uint8_t *__restrict__ valid, float4 *__restrict__ coef) {
  const int k = blockIdx.z;
  if (st[k].converged) return;
  const KfDesc d = kd[k];
  const float *tp = st[k].T;
  if (blockIdx.y == 0) {
    if (int(blockIdx.x) * 128 >= d.Mc * LPQ) return;
    const KfMapDesc &m = md[d.map];
    // (the maps' pointers come from a descriptor: tied to a kernel argument they are global to the compiler, dev.h: rebase)
    line_features_body<LPQ>(blockIdx.x, stack_all + d.slot_off, d.Mc, d.slot_off, tp, Vec3<float>(d.pz[0], d.pz[1], d.pz[2]), min_match_sq_dis,
                       rebase(stack_all, m.corner_sorted), rebase(stack_all, m.corner_cells), m.corner_grid, valid, coef, order ? order + d.slot_off : nullptr);
  } else {
    if (int(blockIdx.x) * 128 >= d.Ms * LPQ) return;
    const KfMapDesc &m = md[d.map];
    const FeatFrame fr{stack_all + d.slot_off + d.Mc, d.Ms, d.slot_off + d.Mc, 0, order ? order + d.slot_off + d.Mc : nullptr};
    const FeatScalars fs{min_match_sq_dis, min_plane_dis, mapping_mode, {d.pz[0], d.pz[1], d.pz[2]}};
    features_body<true, LPQ>(fr, fs, blockIdx.x, tp, rebase(stack_all, m.surf_sorted), rebase(stack_all, m.surf_cells), m.surf_grid, valid, coef, nullptr, nullptr);
  }
}

template <int LPQ>
__global__ void __launch_bounds__(128) k_kf_round(const KfDesc *__restrict__ kd, const KfMapDesc *__restrict__ md, const OdomState *__restrict__ st,
                                                 const float4 *__restrict__ stack_all, const uint32_t *__restrict__ order, float min_match_sq_dis, float min_plane_dis,
                                                 int mapping_mode, uint8_t *__restrict__ valid, float4 *__restrict__ coef) {
  kf_round_body<LPQ>(kd, md, st, stack_all, order, min_match_sq_dis, min_plane_dis, mapping_mode, valid, coef);
}
#define KF_OCC_VARIANT(W)                                                                                                                      \
  __global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(W, W)))                                                            \
  k_kf_round1_w##W(const KfDesc *__restrict__ kd, const KfMapDesc *__restrict__ md, const OdomState *__restrict__ st,                          \
                   const float4 *__restrict__ stack_all, const uint32_t *__restrict__ order, float min_match_sq_dis, float min_plane_dis,      \
                   int mapping_mode, uint8_t *__restrict__ valid, float4 *__restrict__ coef) {                                                 \
    kf_round_body<1>(kd, md, st, stack_all, order, min_match_sq_dis, min_plane_dis, mapping_mode, valid, coef);                                \
  }
// the one-lane-per-query form is bound by gather latency: 8 waves per SIMD (64 VGPRs, the fit phase spills a little) beats
// the 5 waves the default allocation gives by 15% (54.4 -> 46.1 ms at 1000 HDL-64 keyframes)
KF_OCC_VARIANT(8)
KF_OCC_VARIANT(6)

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

// the map cell of every query under its keyframe's current transform: the key the queries' processing order is sorted by (all ones: outside
// the map's grid — such a query finds nothing and costs nothing wherever it sits)
__global__ void __launch_bounds__(256) k_kf_query_keys(const KfDesc *__restrict__ kd, const KfMapDesc *__restrict__ md, const OdomState *__restrict__ st,
                                                      const float4 *__restrict__ stack_all, uint32_t *__restrict__ keys) {
  const int k = blockIdx.z;
  const KfDesc d = kd[k];
  const bool surf = blockIdx.y == 1;
  const int M = surf ? d.Ms : d.Mc, i = int(blockIdx.x) * 256 + threadIdx.x;
  if (i >= M) return;
  const int slot = d.slot_off + (surf ? d.Mc : 0) + i;
  const KfMapDesc &m = md[d.map];
  const GridDesc &g = surf ? m.surf_grid : m.corner_grid;
  const float *tp = st[k].T;
  const Quat<float> q(tp[3], tp[0], tp[1], tp[2]);
  const float4 po = stack_all[slot];
  const Vec3<float> r = rotate(q, Vec3<float>(po.x, po.y, po.z));
  const int cx = cell_coord(r.x + tp[4], g.inv_cell) - g.origin[0], cy = cell_coord(r.y + tp[5], g.inv_cell) - g.origin[1],
            cz = cell_coord(r.z + tp[6], g.inv_cell) - g.origin[2];
  const bool in = cx >= 0 && cy >= 0 && cz >= 0 && cx < g.dims[0] && cy < g.dims[1] && cz < g.dims[2];
  keys[slot] = in ? uint32_t(cx + g.dims[0] * (cy + g.dims[1] * cz)) : 0xFFFFFFFFu;
}
void launch_kf_query_keys(const KfDesc *kd, const KfMapDesc *md, const OdomState *st, int n_keyframes, int max_Mc, int max_Ms, const float4 *stack_all, uint32_t *keys,
                          hipStream_t s) {
  if (n_keyframes <= 0) return;
  hipLaunchKernelGGL(k_kf_query_keys, dim3(std::max(1, cdiv(std::max(max_Mc, max_Ms), 256)), 2, n_keyframes), dim3(256), 0, s, kd, md, st, stack_all, keys);
  LIO_HIP(hipGetLastError());
}

void launch_kf_round(const KfDesc *kd, const KfMapDesc *md, const OdomState *st, int n_keyframes, int max_Mc, int max_Ms, long long total_queries,
                     const float4 *stack_all, const uint32_t *order, float min_match_sq_dis, float min_plane_dis, int mapping_mode, uint8_t *valid, float4 *coef,
                     hipStream_t s) {
  // lanes per query: a small batch is latency-bound (8 lanes shorten each query's dependent candidate walk); once the batch
  // fills the GPU many times over, one lane per query wins 1.8x (no merge rounds, no idle lanes in the fit): measured
  // 96 / 69 / 59 / 55 ms for 8 / 4 / 2 / 1 lanes at 1000 HDL-64 keyframes.  The result does not depend on the split.
  static const int lpq_env = [] { const char *e = std::getenv("LIO_KF_LPQ"); return e ? std::atoi(e) : 0; }();
  const int lpq = lpq_env ? lpq_env : (total_queries >= 250000 ? 1 : total_queries >= 60000 ? 4 : 8);   // (round 6, flat candidate lists: 16 keyframes = 377 k queries 0.89 ms at one lane against 0.96 at four; 8 keyframes 0.62 against 0.55)
  const int bx = std::max(1, cdiv((long long)std::max(max_Mc, max_Ms) * lpq, 128));
  const dim3 grid(bx, 2, n_keyframes);
#define KF_ROUND(L) hipLaunchKernelGGL(k_kf_round<L>, grid, dim3(128), 0, s, kd, md, st, stack_all, order, min_match_sq_dis, min_plane_dis, mapping_mode, valid, coef)
  static const int occ_env = [] { const char *e = std::getenv("LIO_KF_OCC"); return e ? std::atoi(e) : 8; }();   // A/B: 0 as compiled, 6, 8 waves per SIMD
  if (lpq == 1 && occ_env == 8) hipLaunchKernelGGL(k_kf_round1_w8, grid, dim3(128), 0, s, kd, md, st, stack_all, order, min_match_sq_dis, min_plane_dis, mapping_mode, valid, coef);
  else if (lpq == 1 && occ_env == 6) hipLaunchKernelGGL(k_kf_round1_w6, grid, dim3(128), 0, s, kd, md, st, stack_all, order, min_match_sq_dis, min_plane_dis, mapping_mode, valid, coef);
  else if (lpq == 1) KF_ROUND(1);
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
