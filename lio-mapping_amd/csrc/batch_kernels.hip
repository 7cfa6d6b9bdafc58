// batch_kernels.hip — see batch_kernels.h.  gfx950 kernels; blockIdx.y (or .z) = the window, blockIdx.x = a block of that window's
// share, blocks past a window's own size leave at once.  Clouds are float4 AoS as everywhere (cloud_kernels.hip); the batch's
// arrays are the windows' ranges laid end to end, so a wave's 64 lanes always read one window's consecutive points.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "batch_kernels.h"
#include "seg_sort.h"
#include "cloud_device.h"

namespace lio {

#define BW_THREADS 256

int bw_round_blocks(int M) { return std::max(1, cdiv(M, 64)); }

// ------------------------------------------------------------------------------------------------
// BuildLocalMap, first half: pcl::transformPointCloud + `+=` (Estimator.cc:1480-1507) and, from the point still in registers, the
// filter's sort key — the window above PCL's voxel index in absolute cells (cloud_kernels.hip: k_vox_keys_abs; the window in the
// high word keeps every window's points together through ONE sort of the whole batch) — and the block's share of the bounds.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BW_THREADS) k_bw_concat_keys(const BatchWin *__restrict__ win, float4 *__restrict__ local_all, uint32_t *__restrict__ keys,
                                                              float *__restrict__ partial, int *__restrict__ range_overflow) {
  const int w = blockIdx.y;
  const BatchWin &W = win[w];
  const int base = int(blockIdx.x) * BW_THREADS;
  if (base >= W.loc_cap) return;
  const int gid = base + threadIdx.x;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  float cnt = 0;
  uint32_t key = BW_KEY_NONE;
  if (gid < W.n_local) {
    int sidx = 0;
    for (int k = 1; k < W.nseg; ++k)
      if (gid >= W.seg[k].dst_off) sidx = k;
    const BwSeg &sg = W.seg[sidx];
    const float4 p = rebase(local_all, sg.src)[gid - sg.dst_off];
    float4 o;
    if (sg.identity) {
      o = p;
    } else {
      const float *m = sg.tf.m;
      o.x = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
      o.y = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
      o.z = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
      o.w = p.w;
    }
    if (sg.set_intensity) o.w = sg.intensity;
    local_all[W.loc_off + gid] = o;
    if (finite3(o)) {
      cnt = 1.f;
      mn[0] = mx[0] = o.x; mn[1] = mx[1] = o.y; mn[2] = mx[2] = o.z;
      const float cx = floorf(o.x * W.inv_leaf), cy = floorf(o.y * W.inv_leaf), cz = floorf(o.z * W.inv_leaf);
      // PCL's voxel index in ABSOLUTE cells, z 9 bits | y 11 | x 11 (+-102 m of height at a 0.4 m leaf; a window beyond that takes the
      // single-window path); the sort runs on the key relative to the window's bounds (seg_sort.h: KeyLayout)
      if (fabsf(cx) < 1024.f && fabsf(cy) < 1024.f && fabsf(cz) < 255.f) key = (uint32_t(int(cz) + 256) << 22) | (uint32_t(int(cy) + 1024) << 11) | uint32_t(int(cx) + 1024);
      else range_overflow[w] = 1;
    }
    keys[W.loc_off + gid] = key;
  }
  __shared__ float sm[7][BW_THREADS / 64];
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
    for (int q = 1; q < BW_THREADS / 64; ++q) v = t < 3 ? fminf(v, sm[t][q]) : (t < 6 ? fmaxf(v, sm[t][q]) : v + sm[t][q]);
    partial[(size_t(W.loc_off / BW_THREADS) + blockIdx.x) * 8 + t] = v;
  }
}

// the window's bounds folded into VoxParams exactly as the single-window filter does, and from them how the sort sees the keys: one block
// per window
__device__ __forceinline__ int bits_for(int extent) { int b = 0; while ((1 << b) < extent) ++b; return b; }
__global__ void __launch_bounds__(BW_THREADS) k_bw_key_layout(const BatchWin *__restrict__ win, const float *__restrict__ partial, VoxParams *__restrict__ params,
                                                             KeyLayout *__restrict__ layout, int *__restrict__ range_overflow, int max_bits) {
  const int w = blockIdx.x;
  const BatchWin &W = win[w];
  const int ntiles = W.loc_cap / BW_THREADS;
  __shared__ float sm[7][BW_THREADS];
  const int t = threadIdx.x;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  float cnt = 0;
  const float *pw = partial + size_t(W.loc_off / BW_THREADS) * 8;
  for (int b = t; b < ntiles; b += BW_THREADS) {
    for (int d = 0; d < 3; ++d) { mn[d] = fminf(mn[d], pw[size_t(b) * 8 + d]); mx[d] = fmaxf(mx[d], pw[size_t(b) * 8 + 3 + d]); }
    cnt += pw[size_t(b) * 8 + 6];   // integers below 2^24: exact in any order
  }
  for (int d = 0; d < 3; ++d) { sm[d][t] = mn[d]; sm[3 + d][t] = mx[d]; }
  sm[6][t] = cnt;
  __syncthreads();
  for (int st = BW_THREADS / 2; st > 0; st >>= 1) {
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
    dd[d] = (long long)((v.mx[d] - v.mn[d]) * W.inv_leaf) + 1;
    v.minb[d] = int(floorf(v.mn[d] * W.inv_leaf));
    const int maxb = int(floorf(v.mx[d] * W.inv_leaf));
    v.divb[d] = maxb - v.minb[d] + 1;
  }
  v.overflow = (sm[6][0] > 0 && dd[0] * dd[1] * dd[2] > (long long)INT_MAX) ? 1 : 0;
  v.n_valid = int(sm[6][0]);
  params[w] = v;
  KeyLayout L{0, 0, 0, 0, 0, 0};
  if (v.n_valid > 0 && !range_overflow[w]) {
    // floor(x inv_leaf) is monotone in x: the smallest stored cell coordinate of the window is the bound's
    L.mx = v.minb[0] + 1024; L.my = v.minb[1] + 1024; L.mz = v.minb[2] + 256;
    const int bx = bits_for(v.divb[0]), by = bits_for(v.divb[1]), bz = bits_for(v.divb[2]);
    L.bx = bx; L.by = bx + by; L.bits = bx + by + bz;
    if (L.bits > max_bits) range_overflow[w] = 1;   // the launch's passes order max_bits + 1 bits ("no point" sits one above the real keys): single-window path
  }
  layout[w] = L;
}

void launch_bw_concat_keys(const BatchWin *win, int B, int max_local, float4 *local_all, uint32_t *keys, float *partial, VoxParams *params, KeyLayout *layout,
                           int *range_overflow, int max_bits, hipStream_t s) {
  if (B <= 0 || max_local <= 0) return;
  hipLaunchKernelGGL(k_bw_concat_keys, dim3(cdiv(max_local, BW_THREADS), B), dim3(BW_THREADS), 0, s, win, local_all, keys, partial, range_overflow);
  hipLaunchKernelGGL(k_bw_key_layout, dim3(B), dim3(BW_THREADS), 0, s, win, partial, params, layout, range_overflow, max_bits);
  LIO_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// pcl::VoxelGrid (Estimator.cc:1518-1519), behind the sort: heads per 256-entry tile of a window's sorted range, then the centroids — a
// run is added up in sorted order, i.e. in ascending original index (the sort is stable), the order the oracle fixes.  Keys: the sort's
// relative keys, all ones = no point (sorted last); positions from n_local on hold nothing.
// ------------------------------------------------------------------------------------------------
#define BW_SORTED_NONE 0xFFFFFFFFu
__device__ __forceinline__ bool bw_is_head(const uint32_t *__restrict__ wkeys, int i, uint32_t k) {   // wkeys: the window's sorted keys, i < n_local
  return k != BW_SORTED_NONE && (i == 0 || wkeys[i - 1] != k);
}
__global__ void __launch_bounds__(BW_THREADS) k_bw_vox_heads(const BatchWin *__restrict__ win, const uint32_t *__restrict__ keys, int *__restrict__ tile_heads) {
  const int w = blockIdx.y;
  const BatchWin &W = win[w];
  const int ntiles = W.loc_cap / BW_THREADS;
  if (int(blockIdx.x) >= ntiles) return;
  __shared__ int swave[BW_THREADS / 64];
  const int i = int(blockIdx.x) * BW_THREADS + threadIdx.x;
  const uint32_t *wk = keys + W.loc_off;
  const uint32_t k = i < W.n_local ? wk[i] : BW_SORTED_NONE;
  const unsigned long long b = __ballot(bw_is_head(wk, i, k));
  if ((threadIdx.x & 63) == 0) swave[threadIdx.x >> 6] = __popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) tile_heads[W.loc_off / BW_THREADS + blockIdx.x] = (swave[0] + swave[1]) + (swave[2] + swave[3]);
}

__global__ void __launch_bounds__(BW_THREADS) k_bw_vox_centroids(const BatchWin *__restrict__ win, const float4 *__restrict__ pts, const uint32_t *__restrict__ keys,
                                                                const uint32_t *__restrict__ vals, const int *__restrict__ tile_heads, float4 *__restrict__ out,
                                                                const VoxParams *__restrict__ params, int *__restrict__ range_overflow,
                                                                BwVoxOut *__restrict__ vout) {
  const int w = blockIdx.y;
  const BatchWin &W = win[w];
  const int ntiles = W.loc_cap / BW_THREADS;
  if (int(blockIdx.x) >= ntiles) return;
  __shared__ float4 sp[BW_THREADS];
  __shared__ uint32_t sk[BW_THREADS];
  __shared__ int swave[BW_THREADS / 64], sbase[BW_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int base_i = int(blockIdx.x) * BW_THREADS, i = base_i + tid;
  const uint32_t *wk = keys + W.loc_off;
  const uint32_t *wvals = vals + W.loc_off;
  const int *th = tile_heads + W.loc_off / BW_THREADS;
  int before = 0;
  for (int b = tid; b < int(blockIdx.x); b += BW_THREADS) before += th[b];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
  const uint32_t k = i < W.n_local ? wk[i] : BW_SORTED_NONE;
  const bool real = k != BW_SORTED_NONE;
  sk[tid] = k;
  if (real) sp[tid] = pts[wvals[i]];
  const bool head = bw_is_head(wk, i, k);
  const unsigned long long hb = __ballot(head);
  if (lane == 0) { swave[wv] = __popcll(hb); sbase[wv] = before; }
  __syncthreads();
  int pos = (sbase[0] + sbase[1]) + (sbase[2] + sbase[3]);
  for (int q = 0; q < wv; ++q) pos += swave[q];
  pos += __popcll(hb & ((1ull << lane) - 1ull));
  if (int(blockIdx.x) == ntiles - 1 && tid == BW_THREADS - 1) {   // the window's last tile knows the total
    BwVoxOut o;
    o.count = pos + (head ? 1 : 0);
    o.params = params[w];
    o.range_overflow = range_overflow[w];
    range_overflow[w] = 0;   // clear for the next solve (no fill command in front of it)
    vout[w] = o;
  }
  if (!head) return;
  float ax = 0, ay = 0, az = 0, ai = 0;
  int e = tid;
  while (e < BW_THREADS && sk[e] == k) { const float4 p = sp[e]; ax += p.x; ay += p.y; az += p.z; ai += p.w; ++e; }
  int cnt = e - tid;
  if (e == BW_THREADS) {
    int g = base_i + BW_THREADS;
    while (g < W.n_local && wk[g] == k) { const float4 p = pts[wvals[g]]; ax += p.x; ay += p.y; az += p.z; ai += p.w; ++g; ++cnt; }
  }
  const float c = float(cnt);
  out[W.loc_off + pos] = make_float4(ax / c, ay / c, az / c, ai / c);
}

void launch_bw_vox_finish(const BatchWin *win, int B, int max_cap, const float4 *local_all, const uint32_t *keys_sorted, const uint32_t *vals_sorted, int *tile_heads,
                          float4 *filtered_all, const VoxParams *params, int *range_overflow, BwVoxOut *out, hipStream_t s) {
  if (B <= 0 || max_cap <= 0) return;
  const int ntiles = max_cap / BW_THREADS;
  hipLaunchKernelGGL(k_bw_vox_heads, dim3(ntiles, B), dim3(BW_THREADS), 0, s, win, keys_sorted, tile_heads);
  hipLaunchKernelGGL(k_bw_vox_centroids, dim3(ntiles, B), dim3(BW_THREADS), 0, s, win, local_all, keys_sorted, vals_sorted, tile_heads, filtered_all, params,
                     range_overflow, out);
  LIO_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// what a solve's feature stage starts from (cloud_kernels.hip: k_solve_setup, per window): feature flags cleared, the newest
// frame's Gauss-Newton state = its local transform; a window without a newest frame counts as converged from the start
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BW_THREADS) k_bw_setup(const BatchWin *__restrict__ win, uint8_t *__restrict__ valid_all, OdomState *__restrict__ odom,
                                                        int *__restrict__ n_converged) {
  const int w = blockIdx.y;
  const BatchWin &W = win[w];
  const size_t i = (size_t(blockIdx.x) * BW_THREADS + threadIdx.x) * 16;
  uint8_t *valid = valid_all + W.slot_base;   // slot_base is a multiple of 16
  const size_t n = size_t(W.n_slots);
  if (i + 16 <= n) *reinterpret_cast<uint4 *>(valid + i) = make_uint4(0, 0, 0, 0);
  else for (size_t k = i; k < n; ++k) valid[k] = 0;
  if (blockIdx.x == 0) {
    unsigned *o = reinterpret_cast<unsigned *>(odom + w);
    const int nw = int(sizeof(OdomState) / 4);
    const float *T = W.tf[LIO_BW_MAX_STATIC];
    const bool none = W.newest.M <= 0;
    if (int(threadIdx.x) < nw) {
      unsigned v = threadIdx.x < 8 ? __float_as_uint(T[threadIdx.x]) : 0u;
      if (none && threadIdx.x == offsetof(OdomState, converged) / 4) v = 1u;
      o[threadIdx.x] = v;
    }
    if (none && threadIdx.x == 0) atomicAdd(n_converged, 1);
  }
}
void launch_bw_setup(const BatchWin *win, int B, int max_slots, uint8_t *valid_all, OdomState *odom, int *n_converged, hipStream_t s) {
  if (B <= 0) return;
  const int nb = std::max(1, cdiv((long long)max_slots, BW_THREADS * 16));
  hipLaunchKernelGGL(k_bw_setup, dim3(nb, B), dim3(BW_THREADS), 0, s, win, valid_all, odom, n_converged);
  LIO_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// K-NN grid (KdTreeFLANN::setInputCloud, Estimator.cc:1544-1545): a window's filtered points ordered by cell (seg_sort.h: the same
// segmented sort as the filter's, so the order inside a cell is the filtered cloud's — defined, unlike a histogram's atomics) and a
// dense table of run starts per window; the tables of the batch are laid end to end, an entry is a position in the batch's
// cell-sorted point array and the search kernels take that array's base as their map.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BW_THREADS) k_bw_cell_keys(const BatchWin *__restrict__ win, const BatchGrid *__restrict__ grid,
                                                            const float4 *__restrict__ filtered_all, uint32_t *__restrict__ keys) {
  const int w = blockIdx.y;
  const BatchGrid &G = grid[w];
  const int i = int(blockIdx.x) * BW_THREADS + threadIdx.x;
  if (i >= G.n_filtered) return;
  const int gi = win[w].loc_off + i;
  const float4 p = filtered_all[gi];
  const GridDesc &g = G.g;
  int cx = cell_coord(p.x, g.inv_cell) - g.origin[0];
  int cy = cell_coord(p.y, g.inv_cell) - g.origin[1];
  int cz = cell_coord(p.z, g.inv_cell) - g.origin[2];
  cx = min(max(cx, 0), g.dims[0] - 1); cy = min(max(cy, 0), g.dims[1] - 1); cz = min(max(cz, 0), g.dims[2] - 1);
  keys[gi] = uint32_t(cx + g.dims[0] * (cy + g.dims[1] * cz));
}
// behind the sort by cell: the cell-sorted points, and the table — entry c = position of the first point whose cell is >= c (an empty cell
// holds the start of the next occupied one, entry n_cells the end), every entry written here: no histogram, no scan, no clearing.
// A point whose cell differs from its predecessor's owns the entries (predecessor's cell, its own]; the wave writes those ranges one
// after the other, 64 entries per store.
__global__ void __launch_bounds__(BW_THREADS) k_bw_cell_table(const BatchWin *__restrict__ win, const BatchGrid *__restrict__ grid,
                                                             const float4 *__restrict__ filtered_all, const uint32_t *__restrict__ keys,
                                                             const uint32_t *__restrict__ vals, int *__restrict__ cells_all, float4 *__restrict__ sorted_all) {
  const int w = blockIdx.y;
  const BatchGrid &G = grid[w];
  const int n = G.n_filtered;
  const int i0 = int(blockIdx.x) * BW_THREADS;
  if (i0 >= max(n, 1)) return;
  const int off = win[w].loc_off, i = i0 + threadIdx.x, lane = threadIdx.x & 63;
  const int ncells = G.g.dims[0] * G.g.dims[1] * G.g.dims[2];
  int *cells = cells_all + G.cell_off;
  if (n == 0) {   // (block 0 only)
    for (int c = threadIdx.x; c <= ncells; c += BW_THREADS) cells[c] = off;
    return;
  }
  int lo = 0, hi = -1;   // entries [lo, hi] get the value off + i
  if (i < n) {
    const uint32_t gsrc = vals[off + i];
    float4 p = filtered_all[gsrc];
    p.w = __int_as_float(int(gsrc) - off);   // the point's index in the window's filtered cloud: what the search orders ties by
    sorted_all[off + i] = p;
    const int k = int(keys[off + i]), prev = i ? int(keys[off + i - 1]) : -1;
    lo = prev + 1; hi = k;
  }
  // most ranges are one to a few entries (neighbouring occupied cells): the owning lane writes up to four itself; only what is left of the
  // long ones (a jump to the next row or layer) goes through the wave, 64 entries per store
#pragma unroll
  for (int q = 0; q < 4; ++q) if (lo + q <= hi) cells[lo + q] = off + i;
  lo += 4;
  unsigned long long heads = __ballot(hi >= lo);
  while (heads) {
    const int src = __ffsll((long long)heads) - 1;
    heads &= heads - 1;
    const int a = __shfl(lo, src, 64), b = __shfl(hi, src, 64), v = off + (i - lane) + src;
    for (int c = a + lane; c <= b; c += 64) cells[c] = v;
  }
  // entries behind the last occupied cell: the end (the whole block of the last point writes them)
  __shared__ int s_last;
  if (threadIdx.x == 0) s_last = -2;
  __syncthreads();
  if (i == n - 1) s_last = int(keys[off + i]);
  __syncthreads();
  if (s_last != -2)
    for (int c = s_last + 1 + int(threadIdx.x); c <= ncells; c += BW_THREADS) cells[c] = off + n;
}
void launch_bw_cell_keys(const BatchWin *win, const BatchGrid *grid, int B, int max_filtered, const float4 *filtered_all, uint32_t *keys, hipStream_t s) {
  if (B <= 0 || max_filtered <= 0) return;
  hipLaunchKernelGGL(k_bw_cell_keys, dim3(cdiv(max_filtered, BW_THREADS), B), dim3(BW_THREADS), 0, s, win, grid, filtered_all, keys);
  LIO_HIP(hipGetLastError());
}
void launch_bw_cell_table(const BatchWin *win, const BatchGrid *grid, int B, int max_filtered, const float4 *filtered_all, const uint32_t *keys_sorted,
                          const uint32_t *vals_sorted, int *cells_all, float4 *sorted_all, hipStream_t s) {
  if (B <= 0) return;
  hipLaunchKernelGGL(k_bw_cell_table, dim3(std::max(1, cdiv(max_filtered, BW_THREADS)), B), dim3(BW_THREADS), 0, s, win, grid, filtered_all, keys_sorted, vals_sorted,
                     cells_all, sorted_all);
  LIO_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// CalculateFeatures (Estimator.cc:1014-1097) of the frames behind the pivot: grid (64-query blocks of the largest frame, frames,
// windows).  A block always owns 64 queries; LPQ lanes work on each, so the block is 64 LPQ threads wide.  The result does not
// depend on LPQ (the candidate set and the total order (distance, index) do not: cloud_device.h), which therefore follows the
// size of the launch: eight lanes shorten a query's dependent candidate walk when the launch is small, one lane per query wins
// once the launch fills the chip many times over (no merge shuffles, no idle lanes in the fit; DESIGN.md: keyframe batch).
// ------------------------------------------------------------------------------------------------
template <int LPQ>
__device__ __forceinline__ void bw_features_body(const BatchWin *__restrict__ win, const BatchGrid *__restrict__ grid, const float4 *__restrict__ sorted_all,
                                                 const int *__restrict__ cells_all, uint8_t *__restrict__ valid_all, float4 *__restrict__ coef_all,
                                                 float *__restrict__ score_all) {
  const int w = blockIdx.z;
  const BatchWin &W = win[w];
  if (int(blockIdx.y) >= W.nstatic) return;
  const BatchGrid &G = grid[w];
  const FeatScalars fs{W.min_match_sq_dis, W.min_plane_dis, 0, {0.f, 0.f, 0.f}};
  FeatFrame frm = W.fr[blockIdx.y];
  frm.stack = rebase(sorted_all, frm.stack);   // (dev.h: a pointer read from a descriptor is generic until it is tied to a kernel argument)
  features_block<false, LPQ, 64 * LPQ>(frm, fs, int(blockIdx.x), &W.tf[0][0], sorted_all, cells_all + G.cell_off, G.g, valid_all, coef_all, score_all,
                                       nullptr);
}
template <int LPQ>
__global__ void __launch_bounds__(64 * LPQ) k_bw_features(const BatchWin *__restrict__ win, const BatchGrid *__restrict__ grid,
                                                         const float4 *__restrict__ sorted_all, const int *__restrict__ cells_all,
                                                         uint8_t *__restrict__ valid_all, float4 *__restrict__ coef_all, float *__restrict__ score_all) {
  bw_features_body<LPQ>(win, grid, sorted_all, cells_all, valid_all, coef_all, score_all);
}
// the one-lane-per-query form at a fixed occupancy (the keyframe batch's k_kf_round1_w8 gained 15 % from eight waves per SIMD at 64
// VGPRs): LIO_BW_OCC = 6 / 8 selects these for A/B runs
#define BW_FEAT_OCC(W)                                                                                                                            \
  __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(W, W)))                                                                \
  k_bw_features1_w##W(const BatchWin *__restrict__ win, const BatchGrid *__restrict__ grid, const float4 *__restrict__ sorted_all,                \
                      const int *__restrict__ cells_all, uint8_t *__restrict__ valid_all, float4 *__restrict__ coef_all, float *__restrict__ score_all) { \
    bw_features_body<1>(win, grid, sorted_all, cells_all, valid_all, coef_all, score_all);                                                        \
  }
BW_FEAT_OCC(6)
BW_FEAT_OCC(8)
// Measured at 64 / 512 windows (profiles/r5_m_occupancy.txt): features 0.747 / 5.69 ms by the compiler's choice (84 VGPRs, 5 waves),
// 0.709 / 5.25 at 6 waves, 0.699 / 5.02 at 8 (64 VGPRs, 84 B of spills); the rounds lose at both (0.835 / 5.98 -> 0.897 / 6.72 ->
// 1.137 / 8.68: their fit + row phase spills 176 - 240 B).  Default: features at 8, rounds as compiled; BatchKnobs::occupancy = 0 / 6 / 8 forces both.
static int env_int(const char *name, int dflt) { const char *e = std::getenv(name); return e ? std::atoi(e) : dflt; }
BatchKnobs batch_knobs_from_env() {
  BatchKnobs k;
  { const int v = env_int("LIO_BW_LPQ", 0); if (v == 1 || v == 2 || v == 4 || v == 8) k.lanes_per_query = v; }
  { const int v = env_int("LIO_BW_OCC", -1); if (v == 0 || v == 6 || v == 8) k.occupancy = v; }
  { const int v = env_int("LIO_BW_GROUPS", 0); if (v >= 1 && v <= 4) k.loop_groups = v; }
  { const int v = env_int("LIO_BW_AUX_THREADS", 0); if (v == 64 || v == 128 || v == 256) k.aux_threads = v; }
  k.aux_stream = env_int("LIO_BW_AUX_STREAM", 0) != 0 ? 1 : 0;
  { const int v = env_int("LIO_BW_FINISH_THREADS", 0); if (v >= 1 && v <= 8) k.finish_threads = v; }
  return k;
}
static int bw_occ(const BatchKnobs &k, bool features) { return k.occupancy >= 0 ? k.occupancy : (features ? 6 : 0); }   // (round 6, flat candidate lists: features 4.09 ms at 6 waves, 4.29 at 8, 4.31 as compiled; rounds 4.84 as compiled, 6.39 at 6 — 512 windows)
static int bw_lanes_per_query(const BatchKnobs &k, long long total_queries) {
  if (k.lanes_per_query) return k.lanes_per_query;
  // measured with the flat candidate lists (round 6, tools/r6 logs): one lane per query wins from ~100 k queries per launch (8 windows' newest
  // frames: rounds 0.27 -> 0.21 ms), four lanes from ~15 k (one window's newest frame: 0.148 -> 0.127 ms)
  return total_queries >= 100000 ? 1 : (total_queries >= 15000 ? 4 : 8);
}
void launch_bw_features(const BatchWin *win, const BatchGrid *grid, int B, int max_M, int max_static, long long total_queries, const BatchKnobs &knobs,
                        const float4 *sorted_all, const int *cells_all, uint8_t *valid_all, float4 *coef_all, float *score_all, hipStream_t s) {
  if (B <= 0 || max_M <= 0 || max_static <= 0) return;
  const dim3 g(cdiv(max_M, 64), max_static, B);
  switch (bw_lanes_per_query(knobs, total_queries)) {
    case 1:
      if (bw_occ(knobs, true) == 8) hipLaunchKernelGGL(k_bw_features1_w8, g, dim3(64), 0, s, win, grid, sorted_all, cells_all, valid_all, coef_all, score_all);
      else if (bw_occ(knobs, true) == 6) hipLaunchKernelGGL(k_bw_features1_w6, g, dim3(64), 0, s, win, grid, sorted_all, cells_all, valid_all, coef_all, score_all);
      else hipLaunchKernelGGL(k_bw_features<1>, g, dim3(64), 0, s, win, grid, sorted_all, cells_all, valid_all, coef_all, score_all);
      break;
    case 2: hipLaunchKernelGGL(k_bw_features<2>, g, dim3(128), 0, s, win, grid, sorted_all, cells_all, valid_all, coef_all, score_all); break;
    case 4: hipLaunchKernelGGL(k_bw_features<4>, g, dim3(256), 0, s, win, grid, sorted_all, cells_all, valid_all, coef_all, score_all); break;
    default: hipLaunchKernelGGL(k_bw_features<8>, g, dim3(512), 0, s, win, grid, sorted_all, cells_all, valid_all, coef_all, score_all); break;
  }
  LIO_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// One round of CalculateLaserOdom (Estimator.cc:1242-1359) for every window whose newest frame has not converged: the search /
// fit / rows launch and the fold + 6x6 step launch of cloud_kernels.hip's launch_odom_round, windows in blockIdx.y / .x.
// A search block owns 64 queries whatever the lanes per query, and leaves ONE row of partial sums: the fold — and with it
// every bit of the step — does not depend on how wide the launch made its blocks.
// ------------------------------------------------------------------------------------------------
template <int LPQ>
__device__ __forceinline__ void bw_odom_round_body(const BatchWin *__restrict__ win, const BatchGrid *__restrict__ grid, const OdomState *__restrict__ odom,
                                                   const float4 *__restrict__ sorted_all, const int *__restrict__ cells_all, uint8_t *__restrict__ valid_all,
                                                   float4 *__restrict__ coef_all, float *__restrict__ score_all, double *__restrict__ partials, int round) {
  const int w = blockIdx.y;
  const OdomState &st = odom[w];
  if (st.converged) return;
  const BatchWin &W = win[w];
  if (int(blockIdx.x) >= W.nb_round) return;
  const BatchGrid &G = grid[w];
  const float *tp = st.T;
  const Quat<float> q(tp[3], tp[0], tp[1], tp[2]);
  const Vec3<float> t(tp[4], tp[5], tp[6]);
  const FeatScalars fs{W.min_match_sq_dis, W.min_plane_dis, 0, {0.f, 0.f, 0.f}};
  FeatFrame frm = W.newest;
  frm.stack = rebase(sorted_all, frm.stack);
  const double v = odom_round_block<LPQ, 64 * LPQ>(fs, frm, q, t, sorted_all, cells_all + G.cell_off, G.g, valid_all, coef_all, score_all, W.newest.slot_off,
                                                   round, W.keep, int(blockIdx.x));
  if (threadIdx.x < 28) partials[(size_t(W.part_off) + blockIdx.x) * 28 + threadIdx.x] = v;
}
template <int LPQ>
__global__ void __launch_bounds__(64 * LPQ) k_bw_odom_round(const BatchWin *__restrict__ win, const BatchGrid *__restrict__ grid,
                                                           const OdomState *__restrict__ odom, const float4 *__restrict__ sorted_all,
                                                           const int *__restrict__ cells_all, uint8_t *__restrict__ valid_all,
                                                           float4 *__restrict__ coef_all, float *__restrict__ score_all,
                                                           double *__restrict__ partials, int round) {
  bw_odom_round_body<LPQ>(win, grid, odom, sorted_all, cells_all, valid_all, coef_all, score_all, partials, round);
}
#define BW_ROUND_OCC(W)                                                                                                                           \
  __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(W, W)))                                                                \
  k_bw_odom_round1_w##W(const BatchWin *__restrict__ win, const BatchGrid *__restrict__ grid, const OdomState *__restrict__ odom,                 \
                        const float4 *__restrict__ sorted_all, const int *__restrict__ cells_all, uint8_t *__restrict__ valid_all,                \
                        float4 *__restrict__ coef_all, float *__restrict__ score_all, double *__restrict__ partials, int round) {                 \
    bw_odom_round_body<1>(win, grid, odom, sorted_all, cells_all, valid_all, coef_all, score_all, partials, round);                               \
  }
BW_ROUND_OCC(6)
BW_ROUND_OCC(8)
__global__ void __launch_bounds__(1024) k_bw_odom_update(const BatchWin *__restrict__ win, OdomState *__restrict__ odom, const double *__restrict__ partials,
                                                        int round, int *__restrict__ n_converged) {
  const int w = blockIdx.x;
  OdomState *st = odom + w;
  if (st->converged) return;
  const BatchWin &W = win[w];
  odom_update_wide_block(partials + size_t(W.part_off) * 28, W.nb_round, st, round, 0, 0, nullptr, HostSignal());
  if (threadIdx.x == 0 && st->converged) atomicAdd(n_converged, 1);   // (thread 0 wrote the flag itself)
}
void launch_bw_odom_round(const BatchWin *win, const BatchGrid *grid, int B, int max_nb, long long total_queries, const BatchKnobs &knobs, int round, OdomState *odom,
                          const float4 *sorted_all, const int *cells_all, uint8_t *valid_all, float4 *coef_all, float *score_all, double *partials,
                          int *n_converged, hipStream_t s) {
  if (B <= 0 || max_nb <= 0) return;
  const dim3 g(max_nb, B);
  switch (bw_lanes_per_query(knobs, total_queries)) {
    case 1:
      if (bw_occ(knobs, false) == 8) hipLaunchKernelGGL(k_bw_odom_round1_w8, g, dim3(64), 0, s, win, grid, odom, sorted_all, cells_all, valid_all, coef_all, score_all, partials, round);
      else if (bw_occ(knobs, false) == 6) hipLaunchKernelGGL(k_bw_odom_round1_w6, g, dim3(64), 0, s, win, grid, odom, sorted_all, cells_all, valid_all, coef_all, score_all, partials, round);
      else hipLaunchKernelGGL(k_bw_odom_round<1>, g, dim3(64), 0, s, win, grid, odom, sorted_all, cells_all, valid_all, coef_all, score_all, partials, round);
      break;
    case 2: hipLaunchKernelGGL(k_bw_odom_round<2>, g, dim3(128), 0, s, win, grid, odom, sorted_all, cells_all, valid_all, coef_all, score_all, partials, round); break;
    case 4: hipLaunchKernelGGL(k_bw_odom_round<4>, g, dim3(256), 0, s, win, grid, odom, sorted_all, cells_all, valid_all, coef_all, score_all, partials, round); break;
    default: hipLaunchKernelGGL(k_bw_odom_round<8>, g, dim3(512), 0, s, win, grid, odom, sorted_all, cells_all, valid_all, coef_all, score_all, partials, round); break;
  }
  hipLaunchKernelGGL(k_bw_odom_update, dim3(B), dim3(1024), 0, s, win, odom, partials, round, n_converged);
  LIO_HIP(hipGetLastError());
}

}  // namespace lio
