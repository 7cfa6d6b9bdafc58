// pointproc.hip — gfx950 kernels of the PointProcessor path (see pointproc.h).
//
// HBM traffic per input point: 16 B read + 16 B ring-ordered write + 4 B curvature +
// 4 B label/mask = 40 B.  The ring stage keeps the whole ring (xyz SoA, mask, labels, curvature and the
// 8 x 512 sort slots) in LDS — one workgroup of 8 waves per ring, wave w sorting subregion w — because
// picks are serially dependent through the ring mask and everything they touch must be
// a few cycles away.  fp32 arithmetic is the reference's, operation for operation (A.1); index lists are
// bit-exact by construction, not by tolerance.
#include <cstring>
#include <hip/hip_runtime.h>

#include <cfloat>
#include <climits>

#include "pointproc.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace lio {

#define PP_PICK_THREADS 512
#define PP_SORT_SLOTS 512
#define PP_WMASK 576     // bytes of a private subregion mask: PP_SORT_SLOTS + 2 x nc (nc <= 15), rounded up to a multiple of 64
#define PP_STAMP_RING 20   // the ring whose block stamps its phases (an HDL-64 ring that looks at the scene, not the sky)
#define PP_WSEL 64       // ints per wave for its speculative picks: max_corner_less_sharp + max_surf_flat <= 64

// B sweeps per launch chain (lio_pp_process_batch): every per-sweep array of the handle is B segments of a fixed stride laid end to
// end, and every kernel of the chain takes the sweep from blockIdx.z and shifts its (kernel-argument, hence global) pointers by
// sweep x stride before doing what it does for one sweep.  A single sweep is the B = 1 case of the same kernels: same code, same bits.
struct PPStrides {
  int pts;      // elements per sweep of the point-indexed arrays (input, keys, azimuths, ring cloud, curvature, mask, labels, staging)
  int table;    // ints per sweep of the (ring, block) count table
  int state;    // ints per sweep of the small state record (counts | ring offsets | first_valid[2] | end_ori)
  int picks;    // ints per sweep of the per-ring pick lists
  int cls;      // elements per sweep of one packed class list / class cloud
};
#define PP_SWEEP(ptr, stride) ptr += size_t(blockIdx.z) * size_t(stride)
#define PP_SWEEP_OPT(ptr, stride) do { if (ptr) ptr += size_t(blockIdx.z) * size_t(stride); } while (0)

// ------------------------------------------------------------------------------------------------
// ring binning
// ------------------------------------------------------------------------------------------------
__device__ inline float azimuth_of(float x, float y) {
  float az = float(2 * M_PI - double(atan2f(y, x)));
  if (double(az) >= 2 * M_PI) az = float(double(az) - 2 * M_PI);
  return az;
}

// Ring binning is a STABLE multi-split of the sweep into <= 128 rings (laser_scans[ring] keeps input order, :193-201): a
// device-wide radix sort of (ring, index) pairs costs ten launches here (55 us at 133 k points).  Instead: per-block ring
// histograms (this kernel), one scan of the ring-major (ring, block) count table, and a scatter that ranks each point
// among the points of its ring inside its block.
#define PP_BIN_THREADS 256
__global__ void __launch_bounds__(PP_BIN_THREADS) k_ring_bin(const float4 *__restrict__ in, const uint16_t *__restrict__ ring_in, int n, float lower,
                                                             float factor, int rings, uint32_t *__restrict__ keys, float *__restrict__ azi,
                                                             int *__restrict__ block_hist, int nblocks, int *first_valid, PPStrides st,
                                                             const int *__restrict__ n_arr, const float4 *const *__restrict__ in_table) {
  if (n_arr) n = n_arr[blockIdx.z];
  if (in_table) in = in_table[blockIdx.z]; else   // (sweeps that already live in device memory are read where they are)
  PP_SWEEP(in, st.pts); PP_SWEEP_OPT(ring_in, st.pts); PP_SWEEP(keys, st.pts); PP_SWEEP(azi, st.pts); PP_SWEEP(block_hist, st.table); PP_SWEEP(first_valid, st.state);
  __shared__ int hist[LIO_PP_MAX_RINGS];
  __shared__ int s_first, s_first0;   // first kept point; first kept point of ring 0 (ring_out[0]->front(), :383)
  for (int r = threadIdx.x; r < rings; r += PP_BIN_THREADS) hist[r] = 0;
  if (threadIdx.x == 0) { s_first = INT_MAX; s_first0 = INT_MAX; }
  __syncthreads();
  const int i = blockIdx.x * PP_BIN_THREADS + threadIdx.x;
  if (i < n) {
    float4 p = in[i];
    uint32_t key = uint32_t(rings);
    float az = 0.f;
    if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
      float dis = sqrtf(p.x * p.x + p.y * p.y);
      float ele = atan2f(p.z, dis);
      az = azimuth_of(p.x, p.y);
      float deg = float(double(ele) * 180.0 / M_PI);  // RadToDeg<float>
      int scan_id = int(double((deg - lower) * factor) + 0.5);
      if (ring_in) scan_id = int(ring_in[i]);  // PointIR variant: the ring comes with the point (PointProcessor.cc:468)
      if (scan_id >= 0 && scan_id < rings) {
        key = uint32_t(scan_id);
        atomicAdd(&hist[scan_id], 1);
        atomicMin(&s_first, i);
        if (scan_id == 0) atomicMin(&s_first0, i);
      }
    }
    keys[i] = key; azi[i] = az;
  }
  __syncthreads();
  for (int r = threadIdx.x; r < rings; r += PP_BIN_THREADS) block_hist[r * nblocks + blockIdx.x] = hist[r];
  if (threadIdx.x == 0 && s_first != INT_MAX) atomicMin(first_valid, s_first);
  if (threadIdx.x == 0 && s_first0 != INT_MAX) atomicMin(first_valid + 1, s_first0);
}

// start_ori_ of the sweep: the azimuth of the first kept point (:261-264), or the value the host's ten-sweep filter
// decided on (infer_start_ori, :348-387).  Kernels read it through this.
__device__ __forceinline__ float sweep_start_ori(const float *__restrict__ azi, const int *__restrict__ first_valid, const float *__restrict__ start_ori_override) {
  return start_ori_override ? *start_ori_override : azi[*first_valid];
}
// infer_start_ori only: the two azimuths the host filter needs
__global__ void k_start_ori_probe(const float *__restrict__ azi, const int *__restrict__ first_valid, float *__restrict__ out, PPStrides st) {
  PP_SWEEP(azi, st.pts); PP_SWEEP(first_valid, st.state); PP_SWEEP(out, 2);
  out[0] = first_valid[0] != INT_MAX ? azi[first_valid[0]] : 0.f;
  out[1] = first_valid[1] != INT_MAX ? azi[first_valid[1]] : __int_as_float(0x7fc00000);
}

// per-ring exclusive scan of the count table in place (block r owns ring r's nblocks counts) + the ring totals
__global__ void __launch_bounds__(256) k_ring_scan(int *__restrict__ table, int nblocks, int *__restrict__ ring_total, PPStrides st) {
  PP_SWEEP(table, st.table); PP_SWEEP(ring_total, LIO_PP_MAX_RINGS);
  __shared__ int swave[4];
  __shared__ int s_carry;
  int *row = table + size_t(blockIdx.x) * nblocks;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < nblocks; base += 256) {
    const int k = base + tid;
    const int c = k < nblocks ? row[k] : 0;
    int incl = c;
    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off, 64); if (lane >= off) incl += v; }
    if (lane == 63) swave[wv] = incl;
    __syncthreads();
    int pre = s_carry;
    for (int w = 0; w < wv; ++w) pre += swave[w];
    if (k < nblocks) row[k] = pre + incl - c;
    __syncthreads();
    if (tid == 255) s_carry = pre + incl;
    __syncthreads();
  }
  if (tid == 0) ring_total[blockIdx.x] = s_carry;
}

// PointIR variant (:481-499): an azimuth behind the first one is unwrapped by 2 pi (half_passed can never be set: its
// condition asks for i > 3 * cloud_size / 2), end_ori_ = the largest unwrapped azimuth, at least 0.
__device__ __forceinline__ float unwrap_azimuth(float az, float start_ori) {
  const float rel = az - start_ori;
  return rel < 0 ? float(double(az) + 2 * M_PI) : az;
}
__global__ void k_ring_end_ori(const uint32_t *__restrict__ keys, const float *__restrict__ azi, int n, int rings, const int *__restrict__ first_valid,
                               int *end_ori_bits, PPStrides st, const int *__restrict__ n_arr) {
  if (n_arr) n = n_arr[blockIdx.z];
  PP_SWEEP(keys, st.pts); PP_SWEEP(azi, st.pts); PP_SWEEP(first_valid, st.state); PP_SWEEP(end_ori_bits, st.state);
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  float v = 0.f;
  if (i < n && int(keys[i]) < rings) v = unwrap_azimuth(azi[i], azi[*first_valid]);
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  if ((threadIdx.x & 63) == 0 && v > 0.f) atomicMax(end_ori_bits, __float_as_int(v));  // positive floats order like their bits
}

__global__ void __launch_bounds__(PP_BIN_THREADS) k_ring_scatter(const float4 *__restrict__ in, const uint32_t *__restrict__ keys,
                                                                 const float *__restrict__ azi, const int *__restrict__ table, int nblocks, int n,
                                                                 const int *__restrict__ ring_total, int *__restrict__ offsets,
                                                                 const int *__restrict__ first_valid, const float *__restrict__ start_ori_override, int rings, double scan_period,
                                                                 float4 *__restrict__ ring_cloud, float *__restrict__ ring_intensity,
                                                                 const int *__restrict__ end_ori_bits, PPStrides st, const int *__restrict__ n_arr,
                                                                 const float4 *const *__restrict__ in_table) {
  if (n_arr) n = n_arr[blockIdx.z];
  if (in_table) in = in_table[blockIdx.z]; else
  PP_SWEEP(in, st.pts); PP_SWEEP(keys, st.pts); PP_SWEEP(azi, st.pts); PP_SWEEP(table, st.table); PP_SWEEP(ring_total, LIO_PP_MAX_RINGS);
  PP_SWEEP(offsets, st.state); PP_SWEEP(first_valid, st.state); PP_SWEEP_OPT(start_ori_override, 1); PP_SWEEP(ring_cloud, st.pts);
  PP_SWEEP(ring_intensity, st.pts); PP_SWEEP_OPT(end_ori_bits, st.state);
  __shared__ int wave_cnt[PP_BIN_THREADS / 64][LIO_PP_MAX_RINGS];
  __shared__ int ring_base[LIO_PP_MAX_RINGS + 1];
  for (int k = threadIdx.x; k < (PP_BIN_THREADS / 64) * LIO_PP_MAX_RINGS; k += PP_BIN_THREADS) (&wave_cnt[0][0])[k] = 0;
  // start of every ring = exclusive scan of the <= 128 ring totals (every block redoes it: 128 values)
  if (threadIdx.x < LIO_PP_MAX_RINGS) ring_base[threadIdx.x + 1] = int(threadIdx.x) < rings ? ring_total[threadIdx.x] : 0;
  if (threadIdx.x == 0) ring_base[0] = 0;
  __syncthreads();
  for (int off = 1; off < LIO_PP_MAX_RINGS; off <<= 1) {
    int v = 0;
    if (threadIdx.x < LIO_PP_MAX_RINGS && int(threadIdx.x) + 1 > off) v = ring_base[threadIdx.x + 1 - off];
    __syncthreads();
    if (threadIdx.x < LIO_PP_MAX_RINGS) ring_base[threadIdx.x + 1] += v;
    __syncthreads();
  }
  if (blockIdx.x == 0)
    for (int r = threadIdx.x; r <= rings; r += PP_BIN_THREADS) offsets[r] = ring_base[r];
  const int i = blockIdx.x * PP_BIN_THREADS + threadIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t key = i < n ? keys[i] : uint32_t(rings);
  const bool valid = key < uint32_t(rings);
  // rank among the earlier lanes of this wave with the same ring: one pass per distinct ring present in the wave
  int rank = 0;
  unsigned long long todo = __ballot(valid);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t k = __shfl(key, leader, 64);
    const unsigned long long same = __ballot(valid && key == k);
    if (valid && key == k) rank = __popcll(same & ((1ull << lane) - 1ull));
    if (lane == leader) wave_cnt[wv][k] = __popcll(same);
    todo &= ~same;
  }
  __syncthreads();
  if (!valid) return;
  int dst = ring_base[key] + table[int(key) * nblocks + blockIdx.x] + rank;
  for (int w = 0; w < wv; ++w) dst += wave_cnt[w][key];
  const float start_ori = sweep_start_ori(azi, first_valid, start_ori_override);
  float rel = azi[i] - start_ori;
  if (rel < 0) rel = float(double(rel) + 2 * M_PI);
  float rel_time = float(scan_period * double(rel) / (2 * M_PI));
  if (end_ori_bits) {  // :507-524: no wrap of the difference, divided by range_ori = end_ori_ - start_ori_
    const float range_ori = __int_as_float(*end_ori_bits) - start_ori;
    const float rel_u = unwrap_azimuth(azi[i], start_ori) - start_ori;
    rel_time = float(scan_period * double(rel_u) / double(range_ori));
  }
  float4 p = in[i];
  ring_intensity[dst] = float(int(p.w)) + rel_time;   // intensity_scans (:413, :524): int(input intensity) + rel_time
  p.w = float(int(key)) + rel_time;
  ring_cloud[dst] = p;
}

// value of lane ^ m (m a compile-time power of two).  m = 1, 2, 8 are DPP moves (quad permutes, a rotation by 8 inside the row of
// 16): no trip through the LDS crossbar, which the eight sorting waves of a block would otherwise saturate; the rest is a shuffle
__device__ __forceinline__ unsigned long long lane_xor_u64(unsigned long long v, int m) {
  int lo = int(unsigned(v)), hi = int(unsigned(v >> 32));
  if (m == 1) { lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true); }        // quad_perm [1,0,3,2]
  else if (m == 2) { lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true); }   // quad_perm [2,3,0,1]
  else if (m == 8) { lo = __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xF, 0xF, true); } // row_ror:8
  else { lo = __shfl_xor(lo, m, 64); hi = __shfl_xor(hi, m, 64); }
  return (static_cast<unsigned long long>(unsigned(hi)) << 32) | unsigned(lo);
}

// counts <- 0, first_valid <- INT_MAX (the atomicMin targets of k_ring_bin), end_ori <- 0: the state a sweep starts from
__global__ void k_pp_init(int *__restrict__ state, int n_count_ints, int *__restrict__ first_valid, int *__restrict__ end_ori, PPStrides st) {
  PP_SWEEP(state, st.state); PP_SWEEP(first_valid, st.state); PP_SWEEP(end_ori, st.state);
  for (int k = threadIdx.x; k < n_count_ints; k += blockDim.x) state[k] = 0;
  if (threadIdx.x < 2) first_valid[threadIdx.x] = INT_MAX;
  if (threadIdx.x == 2) *end_ori = 0;
}

// ------------------------------------------------------------------------------------------------
// per-ring feature picking
// ------------------------------------------------------------------------------------------------
struct PickCfg {
  int rings, nc, ns, max_sharp, max_less_sharp, max_flat;
  float curv_th;
};

__device__ inline float sqdiff(float ax, float ay, float az, float bx, float by, float bz) {
  float dx = ax - bx, dy = ay - by, dz = az - bz;
  return dx * dx + dy * dy + dz * dz;
}

__global__ void __launch_bounds__(PP_PICK_THREADS) k_ring_pick(const float4 *__restrict__ ring_cloud, const int *__restrict__ offsets, PickCfg c,
                                                               float *__restrict__ g_curv, int *__restrict__ g_mask, int8_t *__restrict__ g_label,
                                                               int *__restrict__ pick_idx, int *__restrict__ pick_cnt,
                                                               PPDeviceCounts *counts, PPStrides st, int ring_cap) {
  PP_SWEEP(ring_cloud, st.pts); PP_SWEEP(offsets, st.state); PP_SWEEP(g_curv, st.pts); PP_SWEEP(g_mask, st.pts); PP_SWEEP(g_label, st.pts);
  PP_SWEEP(pick_idx, st.picks); PP_SWEEP(pick_cnt, 3 * LIO_PP_MAX_RINGS);
  counts = reinterpret_cast<PPDeviceCounts *>(reinterpret_cast<int *>(counts) + size_t(blockIdx.z) * size_t(st.state));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int r = blockIdx.x;
  const int base = offsets[r];
  const int n = offsets[r + 1] - base;
  const int tid = threadIdx.x;
  const int cap_sharp = c.ns * c.max_sharp, cap_less = c.ns * c.max_less_sharp, cap_flat = c.ns * c.max_flat;
  const int cap_all = cap_sharp + cap_less + cap_flat;
  int *my_pick = pick_idx + size_t(r) * cap_all;
  if (tid < 3) pick_cnt[r * 3 + tid] = 0;
  // every ring point gets defaults in the global arrays
  for (int i = tid; i < n; i += PP_PICK_THREADS) { g_curv[base + i] = 0.f; g_mask[base + i] = 0; g_label[base + i] = 127; }
  if (n <= 2 * c.nc + 1) return;                         // PointProcessor.cc:660-662
  if (n > ring_cap) { if (tid == 0) atomicExch(&counts->overflow, 1); return; }   // ring_cap: the points per ring this launch's LDS holds (<= LIO_PP_MAX_RING_POINTS)
  // LDS carve (all offsets multiples of 16).  The curvature is NOT kept here: it is written once per point, straight to global memory
  // (18 instead of 22 bytes per point: two workgroups per compute unit at HDL-64E ring lengths)
  const int NP = (n + 63) & ~63;
  unsigned long long *skey = reinterpret_cast<unsigned long long *>(smem);               // 8 * 512 * 8 B
  float *sx = reinterpret_cast<float *>(smem + 8 * PP_SORT_SLOTS * 8);
  float *sy = sx + NP, *sz = sy + NP;
  signed char *smask = reinterpret_cast<signed char *>(sz + NP);
  signed char *slabel = smask + NP;
  unsigned char *snfb = reinterpret_cast<unsigned char *>(slabel + NP);  // MaskPickedInRing reach of every point: nf | nb << 4
  unsigned char *sgap = snfb + NP;
  signed char *macc = reinterpret_cast<signed char *>(sgap + NP);        // the ring's mask incl. the reach of every final pick
  signed char *zfwd = macc + NP;                                         // forward reach of finished subregions beyond their end
  signed char *wmask = zfwd + NP;                                        // 8 private masks, a subregion's range +- nc
  int *wsel = reinterpret_cast<int *>(wmask + 8 * PP_WMASK);             // 8 x (corner picks, flat picks) of the speculative pass
  int *wcnt = wsel + 8 * PP_WSEL;                                        // 8 x (number of corner picks, number of flat picks)
  volatile int *wdone = wcnt + 16;                                       // 8 flags: the wave's picks (and its zfwd entries) are final
  int *lpick = const_cast<int *>(wdone) + 8;                                                // this ring's pick lists, flushed to global memory at the end
#define PICK_STAMP(k) do { if (r == PP_STAMP_RING && tid == 0) counts->pick_stamps[k] = wall_clock64(); } while (0)
  PICK_STAMP(0);
  for (int i = tid; i < n; i += PP_PICK_THREADS) {
    float4 p = ring_cloud[base + i];
    sx[i] = p.x; sy[i] = p.y; sz[i] = p.z; smask[i] = 0; slabel[i] = 127;
  }
  __syncthreads();
  PICK_STAMP(1);
  // ---- PrepareRing (:542-585): writes are idempotent stores of 1, so the i-loop parallelises as is
  for (int i = c.nc + tid; i < n - c.nc; i += PP_PICK_THREADS) {
    float cx = sx[i], cy = sy[i], cz = sz[i];
    float nx = sx[i + 1], ny = sy[i + 1], nz = sz[i + 1];
    float diff_next2 = sqdiff(cx, cy, cz, nx, ny, nz);
    bool done = false;
    if (double(diff_next2) > 0.1) {
      float depth = sqrtf(cx * cx + cy * cy + cz * cz);
      float depth_next = sqrtf(nx * nx + ny * ny + nz * nz);
      if (depth > depth_next) {
        float wb = depth_next / depth;
        float dx = nx - cx * wb, dy = ny - cy * wb, dz = nz - cz * wb;
        float wd = sqrtf(dx * dx + dy * dy + dz * dz) / depth_next;
        if (double(wd) < 0.1) {
          for (int k = 0; k <= c.nc; ++k) smask[i - c.nc + k] = 1;
          done = true;
        }
      } else {
        float wb = depth / depth_next;
        float dx = cx - nx * wb, dy = cy - ny * wb, dz = cz - nz * wb;
        float wd = sqrtf(dx * dx + dy * dy + dz * dz) / depth;
        if (double(wd) < 0.1) {
          for (int k = 0; k <= c.nc; ++k) if (i + 1 + k < n) smask[i + 1 + k] = 1;
          done = true;
        }
      }
    }
    if (!done) {
      float diff_prev2 = sqdiff(cx, cy, cz, sx[i - 1], sy[i - 1], sz[i - 1]);
      float dis2 = cx * cx + cy * cy + cz * cz;
      if (double(diff_next2) > 0.0002 * double(dis2) && double(diff_prev2) > 0.0002 * double(dis2)) smask[i] = 1;
    }
  }
  // ---- reach of MaskPickedInRing (:624-645) for every point, in parallel: a pick at i masks i+1..i+nf and i-1..i-nb, the
  // walks stopping at the first consecutive gap above 0.05 m^2.  With this table the serial pick chain below carries no
  // geometry at all.
  // gap[i] = the step i -> i+1 exceeds 0.05 m^2: one squared distance per point instead of ten per point
  for (int i = tid; i < n - 1; i += PP_PICK_THREADS)
    sgap[i] = double(sqdiff(sx[i + 1], sy[i + 1], sz[i + 1], sx[i], sy[i], sz[i])) > 0.05 ? 1 : 0;
  __syncthreads();
  for (int i = c.nc + tid; i < n - c.nc; i += PP_PICK_THREADS) {
    int nf = c.nc, nb = c.nc;
    for (int q = 1; q <= c.nc; ++q) if (sgap[i + q - 1]) { nf = q - 1; break; }   // step (i+q-1) -> (i+q)
    for (int q = 1; q <= c.nc; ++q) if (sgap[i - q]) { nb = q - 1; break; }       // step (i-q) -> (i-q+1)
    snfb[i] = static_cast<unsigned char>(nf | (nb << 4));
  }
  __syncthreads();
  const int wv = tid >> 6, lane = tid & 63;
  int n_sharp = 0, n_less = 0, n_flat = 0;  // wave 0 only
  // smask is PrepareRing's mask from here on (read-only); macc accumulates it + the reach of every FINAL pick (the ring's mask)
  for (int i = tid; i < NP; i += PP_PICK_THREADS) {
    if (i < n) macc[i] = smask[i];
    zfwd[i] = 0;
  }
  __syncthreads();
  PICK_STAMP(2);
  for (int jg = 0; jg < c.ns; jg += 8) {
    // ---- PrepareSubregion for subregion j = jg + wave: curvature + sort slots
    const int j = jg + wv;
    unsigned long long *wk = skey + wv * PP_SORT_SLOTS;
    int sp = 0, ep = -1;
    bool active = j < c.ns;
    if (active) {
      sp = int((size_t(c.nc) * size_t(c.ns - j) + size_t(n - c.nc) * size_t(j)) / size_t(c.ns));
      ep = int((size_t(c.nc) * size_t(c.ns - 1 - j) + size_t(n - c.nc) * size_t(j + 1)) / size_t(c.ns)) - 1;
      if (ep <= sp) active = false;
    }
    const int region = active ? ep - sp + 1 : 0;
    if (region > PP_SORT_SLOTS) { if (lane == 0) atomicExch(&counts->overflow, 1); active = false; }
    // the wave's 512 keys stay in REGISTERS through the sort (eight per lane): (curvature bits << 32 | ring index) orders like
    // std::sort on pair<float, size_t> for the non-negative curvatures; empty slots sort last.  Which key starts in which slot
    // does not matter to a sort, so lane l computes the points l, l + 64, ... (neighbouring lanes read neighbouring LDS words)
    unsigned long long key[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = q * 64 + lane;
      key[q] = ~0ull;
      if (active && k < region) {
        const int i = sp + k;
        const int npn = 2 * c.nc;
        float dx = float(-npn) * sx[i], dy = float(-npn) * sy[i], dz = float(-npn) * sz[i];
        for (int t = 1; t <= c.nc; ++t) {
          dx += sx[i + t] + sx[i - t];
          dy += sy[i + t] + sy[i - t];
          dz += sz[i + t] + sz[i - t];
        }
        float cv = dx * dx + dy * dy + dz * dz;
        g_curv[base + i] = cv;   // (the zero of the defaults loop above lies behind several block barriers)
        slabel[i] = 0;
        key[q] = (static_cast<unsigned long long>(__float_as_uint(cv)) << 32) | static_cast<unsigned int>(i);
      }
    }
    if (jg == 0) PICK_STAMP(3);
    // ---- bitonic sort of the wave's 512 slots, ascending.  Strides below 8 exchange registers of one lane, strides of 8 and more
    // exchange whole registers with lane ^ (stride / 8): 21 shuffle steps and 24 register steps instead of 45 passes through LDS
    // with a wave fence each (20 us of the kernel's 66 in round 2).
    if (active) {
#pragma unroll
      for (int size = 2; size <= PP_SORT_SLOTS; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
          if (stride >= 8) {
            const int m = stride >> 3;
            const bool up = (lane & (size >> 3)) == 0;
            const bool keep_min = ((lane & m) == 0) == up;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const unsigned long long mine_k = key[q];
              const unsigned long long other = lane_xor_u64(mine_k, m);
              const bool take_other = keep_min ? (other < mine_k) : (other > mine_k);
              key[q] = take_other ? other : mine_k;
            }
          } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              if ((q & stride) == 0) {
                // slot g = 8 lane + q: the direction bit (g & size) is a register bit for size <= 4, a lane bit above
                const bool up = size >= 8 ? ((lane & (size >> 3)) == 0) : ((q & size) == 0);
                const unsigned long long a = key[q], b = key[q | stride];
                const bool sw = (a > b) == up;
                key[q] = sw ? b : a;
                key[q | stride] = sw ? a : b;
              }
            }
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) wk[lane * 8 + q] = key[q];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (jg == 0) PICK_STAMP(4);
    signed char *wm = wmask + wv * PP_WMASK;
    const int off = sp - c.nc;
    // ---- picks (:685-725).  The mask is shared by the ring's subregions (A.4): a pick masks up to nc points on either side, so a
    // subregion sees its predecessors' picks — but only through the <= nc points behind its start (its "zone").  The eight waves
    // therefore pick their subregions at the same time, each on a private copy of the mask (its range +- nc) that starts as
    // PrepareRing's; what finished predecessors reach forward is published in zfwd.  A wave runs freely until the candidate it would
    // pick next lies in its zone: whether the serial loop picks or skips that one depends on the predecessors, so the wave then (and
    // only then) waits for all of them, folds zfwd into its zone and retakes the decision.  Everything it did before is what the
    // serial loop does: up to that candidate the two runs see the same mask on every candidate they actually pick or skip for
    // being masked ... except zone points, none of which was reached.  No pick is ever undone.
    // Inside a subregion the picks are sequentially dependent through the mask only: a wave examines 64 sorted candidates at a time,
    // a ballot finds the first still-eligible one (the one the serial loop would reach next), every lane keeps the masked state of
    // ITS candidate in a register and updates it from the picked index and its reach, so the dependent chain never waits on LDS.
    // (plain pointer: the wave's own DS operations execute in order, and the wave fences at the chunk boundaries keep the compiler
    // from carrying mask bytes across them — a volatile pointer costs a wait per store inside the dependent pick chain)
    int *sel = wsel + wv * PP_WSEL;
    const int zone_end = sp + c.nc;   // zone = [sp, zone_end)
    if (lane == 0) wdone[wv] = 0;
    if (active)
      for (int li = lane; li < region + 2 * c.nc; li += 64) { const int i = off + li; wm[li] = (i >= 0 && i < n) ? smask[i] : 0; }
    __syncthreads();
    if (active) {
      bool resolved = false;
      auto resolve_zone = [&](int idx, bool cand, bool &masked) {
        // every predecessor of this group has published its forward reach (earlier groups did before the last block barrier)
        for (int w2 = 0; w2 < wv; ++w2)
          while (wdone[w2] == 0) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (lane < c.nc) { const int i = sp + lane; if (i <= ep && zfwd[i]) wm[i - off] = 1; }
        masked = masked || (cand && idx < zone_end && zfwd[idx] != 0);
        resolved = true;
      };
      if (wv == 0) { bool dummy = false; resolve_zone(0, false, dummy); }   // nothing to wait for: zfwd already holds the earlier groups
      auto apply_pick = [&](int pidx, int reach, int idx, bool &masked) {
        const int nf = reach & 15, nb = reach >> 4;
        masked = masked || (idx >= pidx - nb && idx <= pidx + nf);
        // one store per lane, no branches: lane 0 -> the pick, lanes 1..nf -> forward reach, lanes nc+1..nc+nb -> backward reach
        int o = 0;
        bool wr = lane == 0;
        if (lane >= 1 && lane <= nf) { o = lane; wr = true; }
        if (lane > c.nc && lane - c.nc <= nb) { o = -(lane - c.nc); wr = true; }
        if (wr) wm[pidx + o - off] = 1;
      };
      // corners: descending curvature
      int num_largest = 0;
      int mine = -1;   // lane q keeps the q-th pick of this subregion; labels and lists are written once, after the chain
      bool stop = false;
      for (int pos = region; pos > 0 && num_largest < c.max_less_sharp && !stop; pos -= 64) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const int k = pos - 1 - lane;
        const bool in = k >= 0;
        const unsigned long long e = in ? wk[k] : 0ull;
        const float cv = __uint_as_float(static_cast<unsigned int>(e >> 32));
        const int idx = int(static_cast<unsigned int>(e));
        const bool above = in && (cv > c.curv_th);
        bool masked = !above || wm[idx - off] != 0;
        const int reach = above ? int(snfb[idx]) : 0;
        int consumed = -1;
        while (num_largest < c.max_less_sharp) {
          const bool elig = above && lane > consumed && !masked;
          const unsigned long long bm = __ballot(elig);
          if (!bm) break;
          const int L = __ffsll((long long)bm) - 1;
          const int pidx = __builtin_amdgcn_readlane(idx, L);      // L is wave-uniform: v_readlane, no LDS round trip
          if (!resolved && pidx < zone_end) { resolve_zone(idx, above, masked); continue; }
          const int preach = __builtin_amdgcn_readlane(reach, L);
          mine = (lane == num_largest) ? pidx : mine;
          ++num_largest;
          apply_pick(pidx, preach, idx, masked);
          consumed = L;
        }
        if (__ballot(in && !above)) stop = true;  // sorted: nothing further down exceeds the threshold
      }
      if (lane < num_largest) { slabel[mine] = lane < c.max_sharp ? 2 : 1; sel[lane] = mine; }
      // flats: ascending curvature
      int num_smallest = 0;
      mine = -1;
      stop = false;
      for (int pos = 0; pos < region && num_smallest < c.max_flat && !stop; pos += 64) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const int k = pos + lane;
        const bool in = k < region;
        const unsigned long long e = in ? wk[k] : 0ull;
        const float cv = __uint_as_float(static_cast<unsigned int>(e >> 32));
        const int idx = int(static_cast<unsigned int>(e));
        const bool below = in && (cv < c.curv_th);
        bool masked = !below || wm[idx - off] != 0;
        const int reach = below ? int(snfb[idx]) : 0;
        int consumed = -1;
        while (num_smallest < c.max_flat) {
          const bool elig = below && lane > consumed && !masked;
          const unsigned long long bm = __ballot(elig);
          if (!bm) break;
          const int L = __ffsll((long long)bm) - 1;
          const int pidx = __builtin_amdgcn_readlane(idx, L);
          if (!resolved && pidx < zone_end) { resolve_zone(idx, below, masked); continue; }
          const int preach = __builtin_amdgcn_readlane(reach, L);
          mine = (lane == num_smallest) ? pidx : mine;
          ++num_smallest;
          apply_pick(pidx, preach, idx, masked);
          consumed = L;
        }
        if (__ballot(in && !below)) stop = true;
      }
      if (lane < num_smallest) { slabel[mine] = -1; sel[c.max_less_sharp + lane] = mine; }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      // publish what this subregion's picks reach beyond its end — all a later subregion can see of it — then the flag
      if (lane < c.nc) { const int i = ep + 1 + lane; if (i < n && wm[i - off]) zfwd[i] = 1; }
      if (lane == 0) { wcnt[2 * wv] = num_largest; wcnt[2 * wv + 1] = num_smallest; }
    } else if (lane == 0) { wcnt[2 * wv] = 0; wcnt[2 * wv + 1] = 0; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) wdone[wv] = 1;
    __syncthreads();
    if (jg == 0) PICK_STAMP(5);
    // the ring's lists in subregion order; every wave folds its private mask into the ring's mask (stores of 1 only, so the
    // overlapping margins need no ordering)
    if (wv == 0) {
      for (int w = 0; w < 8 && jg + w < c.ns; ++w) {
        const int nl = wcnt[2 * w], nsm = wcnt[2 * w + 1];
        const int *sel2 = wsel + w * PP_WSEL;
        if (lane < nl) {
          const int m = sel2[lane];
          lpick[cap_sharp + n_less + lane] = m;
          if (lane < c.max_sharp) lpick[n_sharp + lane] = m;
        }
        if (lane < nsm) lpick[cap_sharp + cap_less + n_flat + lane] = sel2[c.max_less_sharp + lane];
        n_sharp += min(nl, c.max_sharp);
        n_less += nl;
        n_flat += nsm;
      }
    }
    if (active)
      for (int li = lane; li < region + 2 * c.nc; li += 64) { const int i = off + li; if (i >= 0 && i < n && wm[li]) macc[i] = 1; }
    __syncthreads();
  }
  PICK_STAMP(6);
  if (tid == 0) { pick_cnt[r * 3 + 0] = n_sharp; pick_cnt[r * 3 + 1] = n_less; pick_cnt[r * 3 + 2] = n_flat; }
  for (int k = tid; k < cap_all; k += PP_PICK_THREADS) my_pick[k] = lpick[k];
  for (int i = tid; i < n; i += PP_PICK_THREADS) { g_mask[base + i] = int(macc[i]); g_label[base + i] = slabel[i]; }
  __syncthreads();
  PICK_STAMP(7);
}

// prefix of the per-ring pick counts -> compact, ring-major lists (the order the reference pushes them)
// Packing of the per-ring results in ring order, one launch: blockIdx.y = 0 gathers the picked classes of ring r behind those
// of the rings before it (the reference appends ring by ring, :647-735), blockIdx.y = 1 packs the ring's less-flat segment.
// Every block sums the counts of the rings before it itself (<= 128 values per class, one wave each): no offsets kernel.
__global__ void __launch_bounds__(256) k_pp_pack(const float4 *__restrict__ ring_cloud, const int *__restrict__ offsets, const int *__restrict__ pick_idx,
                                                 const int *__restrict__ pick_cnt, PickCfg c, int *__restrict__ class_ring, int *__restrict__ class_idx,
                                                 float4 *__restrict__ cloud1, float4 *__restrict__ cloud2, float4 *__restrict__ cloud3, int cap_total,
                                                 const float4 *__restrict__ lf_staged, const int *__restrict__ lf_ring_count,
                                                 float4 *__restrict__ less_flat, PPDeviceCounts *counts, PPStrides st) {
  PP_SWEEP(ring_cloud, st.pts); PP_SWEEP(offsets, st.state); PP_SWEEP(pick_idx, st.picks); PP_SWEEP(pick_cnt, 3 * LIO_PP_MAX_RINGS);
  PP_SWEEP(class_ring, 3 * size_t(st.cls)); PP_SWEEP(class_idx, 3 * size_t(st.cls)); PP_SWEEP(cloud1, st.cls); PP_SWEEP(cloud2, st.cls); PP_SWEEP(cloud3, st.cls);
  PP_SWEEP(lf_staged, st.pts); PP_SWEEP(lf_ring_count, LIO_PP_MAX_RINGS); PP_SWEEP(less_flat, st.pts);
  counts = reinterpret_cast<PPDeviceCounts *>(reinterpret_cast<int *>(counts) + size_t(blockIdx.z) * size_t(st.state));
  __shared__ int s_dst[4];
  const int r = blockIdx.x, rings = c.rings, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (blockIdx.y == 0) {
    if (wv < 3) {
      int acc = 0;
      for (int q = lane; q < r; q += 64) acc += pick_cnt[q * 3 + wv];
      for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
      if (lane == 0) s_dst[wv] = acc;
    }
    __syncthreads();
    const int cap_sharp = c.ns * c.max_sharp, cap_less = c.ns * c.max_less_sharp, cap_flat = c.ns * c.max_flat;
    const int cap_all = cap_sharp + cap_less + cap_flat;
    const int *mp = pick_idx + size_t(r) * cap_all;
    const int src_off[3] = {0, cap_sharp, cap_sharp + cap_less};
    float4 *clouds[3] = {cloud1, cloud2, cloud3};
    for (int cls = 0; cls < 3; ++cls) {
      const int cnt = pick_cnt[r * 3 + cls], dst = s_dst[cls];
      for (int k = threadIdx.x; k < cnt; k += blockDim.x) {
        const int idx = mp[src_off[cls] + k];
        class_ring[cls * cap_total + dst + k] = r;
        class_idx[cls * cap_total + dst + k] = idx;
        clouds[cls][dst + k] = ring_cloud[offsets[r] + idx];
      }
      if (r == rings - 1 && threadIdx.x == 0) counts->n_class[cls + 1] = dst + cnt;
    }
  } else {
    if (wv == 0) {
      int acc = 0;
      for (int q = lane; q < r; q += 64) acc += lf_ring_count[q];
      for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
      if (lane == 0) s_dst[0] = acc;
    }
    __syncthreads();
    const int dst = s_dst[0], cnt = lf_ring_count[r], src = offsets[r];
    for (int k = threadIdx.x; k < cnt; k += blockDim.x) less_flat[dst + k] = lf_staged[src + k];
    if (r == rings - 1 && threadIdx.x == 0) counts->n_less_flat = dst + cnt;
  }
}

// One workgroup per ring does the whole per-ring VoxelGrid in LDS: bounds, (voxel, index) keys, a bitonic sort sized to
// the ring, run heads, a block scan and the centroids (+ the rel-time recompute of :755-778), written to the ring's own
// segment of a staging cloud; k_lf_compact then packs the segments in ring order.  Two launches instead of bounds + keys +
// a 64-bit device-wide sort (block sort and eight merge passes at this size) + heads + scan + centroids.
#define PP_LF_THREADS 512
__global__ void __launch_bounds__(PP_LF_THREADS) k_lf_ring(const float4 *__restrict__ ring_cloud, const int *__restrict__ offsets,
                                                           const int8_t *__restrict__ label, float inv_leaf, const float *__restrict__ azi,
                                                           const int *__restrict__ first_valid, const float *__restrict__ start_ori_override, double scan_period, float4 *__restrict__ staged,
                                                           int *__restrict__ ring_count, PPStrides st, int pts_cap, int sort_cap) {
  PP_SWEEP(ring_cloud, st.pts); PP_SWEEP(offsets, st.state); PP_SWEEP(label, st.pts); PP_SWEEP(azi, st.pts); PP_SWEEP(first_valid, st.state);
  PP_SWEEP_OPT(start_ori_override, 1); PP_SWEEP(staged, st.pts); PP_SWEEP(ring_count, LIO_PP_MAX_RINGS);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int r = blockIdx.x, tid = threadIdx.x;
  const int base = offsets[r], n = offsets[r + 1] - base;
  // pts_cap points and sort_cap (a power of two >= pts_cap) sort slots fit this launch's LDS; a longer ring was flagged by k_ring_pick
  if (n <= 0 || n > pts_cap) { if (tid == 0) ring_count[r] = 0; return; }
  unsigned long long *skey = reinterpret_cast<unsigned long long *>(smem);                          // sort_cap keys
  float4 *spt = reinterpret_cast<float4 *>(smem + size_t(sort_cap) * 8);                             // n points
  short *spos = reinterpret_cast<short *>(smem + size_t(sort_cap) * 8 + size_t(pts_cap) * 16);       // flags -> positions (< 4096: 16-bit words)
  __shared__ float sb[PP_LF_THREADS / 64][6];
  __shared__ int swave[PP_LF_THREADS / 64];
  __shared__ int s_members;
  const int lane = tid & 63, wv = tid >> 6;
  if (tid == 0) s_members = 0;
  // ---- load + bounds of the less-flat members (label <= 0, A.5)
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i = tid; i < n; i += PP_LF_THREADS) {
    const float4 p = ring_cloud[base + i];
    spt[i] = p;
    if (label[base + i] <= 0) {
      mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
      mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
    }
  }
  for (int off = 32; off > 0; off >>= 1)
    for (int d = 0; d < 3; ++d) { mn[d] = fminf(mn[d], __shfl_xor(mn[d], off, 64)); mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], off, 64)); }
  if (lane == 0) for (int d = 0; d < 3; ++d) { sb[wv][d] = mn[d]; sb[wv][3 + d] = mx[d]; }
  __syncthreads();
  for (int w = 0; w < PP_LF_THREADS / 64; ++w)
    for (int d = 0; d < 3; ++d) { mn[d] = fminf(mn[d], sb[w][d]); mx[d] = fmaxf(mx[d], sb[w][3 + d]); }
  const int minb0 = int(floorf(mn[0] * inv_leaf)), minb1 = int(floorf(mn[1] * inv_leaf)), minb2 = int(floorf(mn[2] * inv_leaf));
  const int div0 = int(floorf(mx[0] * inv_leaf)) - minb0 + 1, div1 = int(floorf(mx[1] * inv_leaf)) - minb1 + 1;
  // ---- keys of the members only, in any order: voxel index (pcl::VoxelGrid, B.1) in the high word, index in the ring in the
  // low word => unique keys, and the members of a voxel come out in ascending index (the summation order the oracle fixes)
  for (int i = tid; i < n; i += PP_LF_THREADS) {
    if (label[base + i] <= 0) {
      const float4 p = spt[i];
      const int i0 = int(floorf(p.x * inv_leaf) - float(minb0));
      const int i1 = int(floorf(p.y * inv_leaf) - float(minb1));
      const int i2 = int(floorf(p.z * inv_leaf) - float(minb2));
      const unsigned int vk = static_cast<unsigned int>(i0 + i1 * div0 + i2 * div0 * div1);
      skey[atomicAdd(&s_members, 1)] = (static_cast<unsigned long long>(vk) << 32) | static_cast<unsigned int>(i);
    }
  }
  __syncthreads();
  const int members = s_members;
  int NS = 128;
  while (NS < members) NS <<= 1;                             // sort size: next power of two
  for (int i = members + tid; i < NS; i += PP_LF_THREADS) skey[i] = ~0ull;
  __syncthreads();
  // bitonic sort.  Strides <= 64 only move keys inside 128-key chunks, and a chunk is handled by one wave: those stages are
  // ordered by a wave-level fence; only the strides >= 128 need the block barrier (10 of the 66 stages at 2048 keys).
  auto exchange = [&](int t, int size, int stride) {
    const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
    const bool up = ((lo & size) == 0);
    const unsigned long long a = skey[lo], b = skey[hi];
    if ((a > b) == up) { skey[lo] = b; skey[hi] = a; }
  };
  auto wave_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  const int nchunks = NS / 128;
  for (int c = wv; c < nchunks; c += PP_LF_THREADS / 64)
    for (int size = 2; size <= 128; size <<= 1)
      for (int stride = size >> 1; stride > 0; stride >>= 1) { exchange(c * 64 + lane, size, stride); wave_sync(); }
  __syncthreads();
  for (int size = 256; size <= NS; size <<= 1) {
    for (int stride = size >> 1; stride >= 128; stride >>= 1) {
      for (int t = tid; t < NS / 2; t += PP_LF_THREADS) exchange(t, size, stride);
      __syncthreads();
    }
    for (int c = wv; c < nchunks; c += PP_LF_THREADS / 64)
      for (int stride = 64; stride > 0; stride >>= 1) { exchange(c * 64 + lane, size, stride); wave_sync(); }
    __syncthreads();
  }
  // ---- run heads and their exclusive positions (block scan: per-thread serial over a contiguous slice, then across threads)
  const int per = NS / PP_LF_THREADS > 0 ? NS / PP_LF_THREADS : 1;
  int local = 0;
  for (int q = 0; q < per; ++q) {
    const int i = tid * per + q;
    if (i < NS) {
      const unsigned long long k = skey[i];
      const bool head = k != ~0ull && (i == 0 || (skey[i - 1] >> 32) != (k >> 32));
      spos[i] = short(head ? 1 : 0);
      local += head ? 1 : 0;
    }
  }
  // inclusive scan of `local` over the block (wave shuffles + one LDS hop)
  int incl = local;
  for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off, 64); if (lane >= off) incl += v; }
  if (lane == 63) swave[wv] = incl;
  __syncthreads();
  int wave_off = 0;
  for (int w = 0; w < wv; ++w) wave_off += swave[w];
  int run = wave_off + incl - local;   // exclusive prefix of this thread's slice
  int total = 0;
  for (int w = 0; w < PP_LF_THREADS / 64; ++w) total += swave[w];
  for (int q = 0; q < per; ++q) {
    const int i = tid * per + q;
    if (i < NS) { const int f = spos[i]; spos[i] = short(f ? run : -1); run += f; }
  }
  __syncthreads();
  // ---- centroids
  const float start_ori = sweep_start_ori(azi, first_valid, start_ori_override);
  for (int i = tid; i < NS; i += PP_LF_THREADS) {
    const int pos = int(spos[i]);
    if (pos < 0) continue;
    const unsigned int vk = static_cast<unsigned int>(skey[i] >> 32);
    float ax = 0, ay = 0, az = 0, ai = 0;
    int e = i;
    while (e < NS && skey[e] != ~0ull && static_cast<unsigned int>(skey[e] >> 32) == vk) {
      const float4 p = spt[static_cast<unsigned int>(skey[e])];
      ax += p.x; ay += p.y; az += p.z; ai += p.w;
      ++e;
    }
    const float cnt = float(e - i);
    float4 o = make_float4(ax / cnt, ay / cnt, az / cnt, ai / cnt);
    const float a = azimuth_of(o.x, o.y);
    float rel = a - start_ori;
    if (rel < 0) rel = float(double(rel) + 2 * M_PI);
    const float rel_time = float(scan_period * double(rel) / (2 * M_PI));
    o.w = float(int(o.w)) + rel_time;
    staged[base + pos] = o;
  }
  if (tid == 0) ring_count[r] = total;
}

// ------------------------------------------------------------------------------------------------
// NormalizeRad<float> / AbsRadDistance (math_utils.h:44-51, PointProcessor.cc:70-72) with the reference's float/double mix
static float normalize_rad_f(float rad) {
  rad = float(std::fmod(double(rad) + M_PI, 2 * M_PI));
  if (rad < 0) rad = float(double(rad) + 2 * M_PI);
  return float(double(rad) - M_PI);
}
static double abs_rad_distance(double a, double b) {
  double rad = std::fmod(a - b + M_PI, 2 * M_PI);
  if (rad < 0) rad += 2 * M_PI;
  return std::fabs(rad - M_PI);
}
float StartOriFilter::Update(float measured, float ring0_front, double rad_diff) {
  float s = measured;
  Push(seen_, n2_, h2_, s);
  if (n1_ >= kDepth) {
    auto used = [&](int i) { return used_[(h1_ + i) % kDepth]; };
    auto seen = [&](int i) { return seen_[(h2_ + i) % kDepth]; };
    const float step_used = normalize_rad_f(used(9) - used(0)) / 9;
    const float step_seen = normalize_rad_f(seen(9) - seen(0)) / 9;
    if (double(std::fabs(normalize_rad_f(s - used(9)))) > rad_diff) {   // a jump: extrapolate the used history (:362-368)
      s = normalize_rad_f(used(9) + step_used);
      if (s < 0) s = float(double(s) + 2 * M_PI);
    }
    bool even = abs_rad_distance(step_used, step_seen) < 0.05;          // :371-383
    for (int k = 9; k >= 1 && even; --k) even = abs_rad_distance(seen(k) - seen(k - 1), step_used) < 0.05;
    if (even && ring0_front == ring0_front) s = ring0_front;
  }
  Push(used_, n1_, h1_, s);
  return s;
}

// ints of a sweep's state record in front of first_valid: PPDeviceCounts, then the ring offsets
static constexpr int kCountInts = int(sizeof(PPDeviceCounts) / sizeof(int));
static constexpr int kOffFirstValid = kCountInts + LIO_PP_MAX_RINGS + 1;
static constexpr int kSizedLdsFromSweeps = 4;   // batches of at least this many sweeps size the per-ring kernels' LDS by the rings they hold
static_assert(sizeof(PPDeviceCounts) % sizeof(int) == 0 && sizeof(PPDeviceCounts) % 8 == 0, "the state record starts with the counts");

float PointProcessorDev::StartOri() {
  if (!processed_ || sel_ >= nsw_ || n_sw_[size_t(sel_)] == 0) return std::nanf("");   // (an empty sweep of a batch has no first point)
  if (!start_ori_known_) {
    const PPStrides st{int(pts_stride_), 0, state_stride_, 0, 0};
    hipLaunchKernelGGL(k_start_ori_probe, dim3(1, 1, nsw_), dim3(1), 0, stream_, azi_.p, d_state_.p + kOffFirstValid, start_ori_dev_.p, st);
    LIO_HIP(hipMemcpyAsync(h_ori_, start_ori_dev_.p, size_t(2) * nsw_ * sizeof(float), hipMemcpyDeviceToHost, stream_));
    LIO_HIP(hipStreamSynchronize(stream_));
    for (int k = nsw_ - 1; k >= 0; --k) { const float v = h_ori_[2 * k]; h_ori_[3 * k + 2] = v; }   // (probe pairs -> triples: the last first)
    start_ori_known_ = true;
  }
  return h_ori_[3 * sel_ + 2];
}

PointProcessorDev::PointProcessorDev(float lower, float upper, int rings, const lio_pp_config &cfg)
    : lower_(lower), upper_(upper), rings_(rings), cfg_(cfg) {
  factor_ = (rings - 1) / (upper - lower);
  int nd = 0;
  LIO_HIP(hipGetDeviceCount(&nd));
  if (nd <= 0) throw DeviceError("no HIP device: the product has no CPU path");
  LIO_HIP(hipStreamCreate(&stream_));
  ring_offsets_.assign(rings + 1, 0);
  state_stride_ = (kOffFirstValid + 4 + 3) & ~3;
  ReserveHost(1);
  // the pick kernel needs up to ~104 KB of dynamic LDS
  LIO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ring_pick), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  LIO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lf_ring), hipFuncAttributeMaxDynamicSharedMemorySize, 4096 * 28));
}
PointProcessorDev::~PointProcessorDev() {
  if (stream_) (void)hipStreamSynchronize(stream_);
  if (h_state_) (void)hipHostFree(h_state_);
  if (stream_) (void)hipStreamDestroy(stream_);
}
void PointProcessorDev::ReserveHost(int B) {
  if (B <= h_cap_sweeps_) return;
  if (h_state_) { LIO_HIP(hipStreamSynchronize(stream_)); (void)hipHostFree(h_state_); h_state_ = nullptr; }
  const size_t ints = size_t(B) * state_stride_;
  LIO_HIP(hipHostMalloc(reinterpret_cast<void **>(&h_state_), ints * sizeof(int) + size_t(4) * B * sizeof(float) + size_t(B) * sizeof(const float4 *)));
  h_ori_ = reinterpret_cast<float *>(h_state_ + ints);
  h_ptr_ = reinterpret_cast<const float4 **>(h_ori_ + size_t(4) * B);   // (4 B floats behind a multiple-of-4 int count: 8-byte aligned)
  h_cap_sweeps_ = B;
}
bool PointProcessorDev::SameSensor(const PointProcessorDev &o) const {
  return lower_ == o.lower_ && upper_ == o.upper_ && rings_ == o.rings_ && std::memcmp(&cfg_, &o.cfg_, sizeof(cfg_)) == 0;
}

void PointProcessorDev::Process(const float *xyzi, size_t n, const uint16_t *ring) {
  ProcessLaunch(xyzi, n, ring);
  ProcessFinish();
}
void PointProcessorDev::ProcessLaunch(const float *xyzi, size_t n, const uint16_t *ring) {
  StartOriFilter *f = &start_ori_filter_;
  ProcessLaunchBatch(&xyzi, ring ? &ring : nullptr, &n, 1, false, &f);
}

// Everything B sweeps need, enqueued on the handle's stream: uploads, ring split, picks, less-flat filter, packing, and ONE copy of the
// sweeps' state records into pinned memory.  Host buffers are read by the uploads: they must stay untouched until ProcessFinish().
void PointProcessorDev::ProcessLaunchBatch(const float *const *xyzi, const uint16_t *const *ring, const size_t *n, int B, bool on_device,
                                           StartOriFilter *const *filters) {
  if (in_flight_) ProcessFinish();
  std::memset(&counts_, 0, sizeof(counts_));
  std::fill(ring_offsets_.begin(), ring_offsets_.end(), 0);
  size_t n_max = 0;
  for (int k = 0; k < B; ++k) n_max = std::max(n_max, n[k]);
  // nothing to do: every count reads zero; the start azimuth stays the last sweep's, as the reference's member does (PointProcessor.cc:261-264)
  last_empty_ = (B <= 0 || n_max == 0);
  if (last_empty_) return;
  bool any_ring = false;
  for (int k = 0; k < B; ++k) any_ring = any_ring || (ring && ring[k] && n[k]);
  for (int k = 0; k < B && any_ring; ++k)
    if (n[k] && !ring[k]) throw std::runtime_error("PointProcessor: a batch mixes sweeps with and without a ring field");
  in_flight_ = true;
  nsw_ = B; sel_ = 0;
  n_sw_.assign(n, n + B);
  hipStream_t s = stream_;
  ReserveHost(B);
  const size_t stride = (n_max + PP_BIN_THREADS - 1) / PP_BIN_THREADS * PP_BIN_THREADS, tot = stride * size_t(B);
  pts_stride_ = stride;
  if (!on_device) in_.reserve(tot);   // (sweeps that already lie in device memory are read where they are)
  ring_cloud_.reserve(tot); ring_intensity_.reserve(tot); azi_.reserve(tot); curv_.reserve(tot); mask_.reserve(tot); label_.reserve(tot);
  keys_.reserve(tot); less_flat_.reserve(tot); lf_tmp_.reserve(tot);
  const int nblocks = int(stride / PP_BIN_THREADS);
  ring_table_.reserve(size_t(B) * rings_ * nblocks); ring_total_.reserve(size_t(B) * LIO_PP_MAX_RINGS); lf_ring_count_.reserve(size_t(B) * LIO_PP_MAX_RINGS);
  d_state_.reserve(size_t(B) * state_stride_);
  PPDeviceCounts *d_counts = reinterpret_cast<PPDeviceCounts *>(d_state_.p);
  int *d_ring_offsets = d_state_.p + kCountInts, *first_valid = d_state_.p + kOffFirstValid, *end_ori = first_valid + 2;
  PickCfg pc{rings_, cfg_.num_curvature_regions, cfg_.num_scan_subregions, cfg_.max_corner_sharp, cfg_.max_corner_less_sharp,
             cfg_.max_surf_flat, cfg_.surf_curv_th};
  const int cap_sharp = pc.ns * pc.max_sharp, cap_less = pc.ns * pc.max_less_sharp, cap_flat = pc.ns * pc.max_flat;
  const int cap_all = cap_sharp + cap_less + cap_flat;
  const int cap_total = rings_ * std::max(cap_less, std::max(cap_sharp, cap_flat));
  cls_stride_ = size_t(cap_total);
  pick_idx_.reserve(size_t(B) * rings_ * cap_all); pick_cnt_.reserve(size_t(B) * 3 * LIO_PP_MAX_RINGS);
  class_ring_.reserve(size_t(B) * 3 * cap_total); class_idx_.reserve(size_t(B) * 3 * cap_total);
  for (int cidx = 1; cidx <= 3; ++cidx) class_cloud_[cidx].reserve(size_t(B) * cap_total);
  const PPStrides st{int(stride), rings_ * nblocks, state_stride_, rings_ * cap_all, cap_total};

  static const bool dbg = std::getenv("LIO_DEBUG_TIMING") != nullptr;
  const auto t_begin = std::chrono::steady_clock::now();
  const hipMemcpyKind up = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  const float4 *const *d_in_table = nullptr;
  if (on_device) {
    // the two kernels that read the input take it where it lies: one table of B pointers goes up instead of B device-to-device copies
    d_in_table_.reserve(B);
    for (int k = 0; k < B; ++k) h_ptr_[k] = reinterpret_cast<const float4 *>(xyzi[k]);
    LIO_HIP(hipMemcpyAsync(d_in_table_.p, h_ptr_, size_t(B) * sizeof(const float4 *), hipMemcpyHostToDevice, s));
    d_in_table = d_in_table_.p;
  }
  for (int k = 0; k < B; ++k) {
    if (n[k] && !on_device) LIO_HIP(hipMemcpyAsync(in_.p + size_t(k) * stride, xyzi[k], n[k] * sizeof(float4), up, s));
  }
  if (dbg) {
    LIO_HIP(hipStreamSynchronize(s));
    std::fprintf(stderr, "[lio_hip pp timing] H2D of %d sweep(s), %zu points the largest, %.1f us\n", B, n_max,
                 std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count());
  }
  // points per sweep: a kernel argument for one sweep, a small device array for several
  const int *d_n = nullptr;
  if (B > 1) {
    d_n_.reserve(B);
    int *h_n = h_state_;   // (staged in the landing zone, which nothing else touches before this upload has run: the chain's later copies follow it on the stream)
    for (int k = 0; k < B; ++k) h_n[k] = int(n[k]);
    LIO_HIP(hipMemcpyAsync(d_n_.p, h_n, size_t(B) * sizeof(int), hipMemcpyHostToDevice, s));
    d_n = d_n_.p;
  }
  const int ni = int(n_max);
  hipLaunchKernelGGL(k_pp_init, dim3(1, 1, B), dim3(64), 0, s, d_state_.p, kCountInts, first_valid, end_ori, st);
  const uint16_t *d_ring = nullptr;
  if (any_ring) {
    ring_in_.reserve(tot);
    for (int k = 0; k < B; ++k)
      if (n[k]) LIO_HIP(hipMemcpyAsync(ring_in_.p + size_t(k) * stride, ring[k], n[k] * sizeof(uint16_t), up, s));
    // end_ori_ = 0 (:439): k_pp_init
    d_ring = ring_in_.p;
  }
  hipLaunchKernelGGL(k_ring_bin, dim3(nblocks, 1, B), dim3(PP_BIN_THREADS), 0, s, in_.p, d_ring, ni, lower_, factor_, rings_, keys_.p, azi_.p, ring_table_.p,
                     nblocks, first_valid, st, d_n, d_in_table);
  if (d_ring) hipLaunchKernelGGL(k_ring_end_ori, dim3(cdiv(ni, 256), 1, B), dim3(256), 0, s, keys_.p, azi_.p, ni, rings_, first_valid, end_ori, st, d_n);
  hipLaunchKernelGGL(k_ring_scan, dim3(rings_, 1, B), dim3(256), 0, s, ring_table_.p, nblocks, ring_total_.p, st);
  const float *d_override = nullptr;
  processed_ = true; start_ori_known_ = false;
  if (cfg_.infer_start_ori && !d_ring) {
    // :348-387 — ten lines of host state between the two passes of PointToRing; costs one round trip, only when enabled.  Sweep k's
    // probe goes through the history it belongs to (filters[k]: the handle it came in through)
    start_ori_dev_.reserve(size_t(3) * B);
    hipLaunchKernelGGL(k_start_ori_probe, dim3(1, 1, B), dim3(1), 0, s, azi_.p, first_valid, start_ori_dev_.p, st);
    LIO_HIP(hipMemcpyAsync(h_ori_, start_ori_dev_.p, size_t(2) * B * sizeof(float), hipMemcpyDeviceToHost, s));
    LIO_HIP(hipStreamSynchronize(s));
    for (int k = B - 1; k >= 0; --k) {   // (probe pairs -> (probe, probe, used) triples in place: the last sweep first)
      const float measured = h_ori_[2 * k], ring0 = h_ori_[2 * k + 1];
      h_ori_[3 * k] = measured; h_ori_[3 * k + 1] = ring0;
    }
    float *h_used = reinterpret_cast<float *>(h_state_);   // (staging of the B values on their way up; the landing zone is idle)
    for (int k = 0; k < B; ++k) {
      StartOriFilter *f = (filters && filters[k]) ? filters[k] : &start_ori_filter_;
      h_ori_[3 * k + 2] = n[k] ? f->Update(h_ori_[3 * k], h_ori_[3 * k + 1], cfg_.rad_diff) : std::nanf("");
      h_used[B + k] = h_ori_[3 * k + 2];
    }
    LIO_HIP(hipMemcpyAsync(start_ori_dev_.p + size_t(2) * B, h_used + B, size_t(B) * sizeof(float), hipMemcpyHostToDevice, s));
    d_override = start_ori_dev_.p + size_t(2) * B;
    start_ori_known_ = true;
  } else {
    start_ori_dev_.reserve(size_t(3) * B);
  }
  hipLaunchKernelGGL(k_ring_scatter, dim3(nblocks, 1, B), dim3(PP_BIN_THREADS), 0, s, in_.p, keys_.p, azi_.p, ring_table_.p, nblocks, ni, ring_total_.p, d_ring_offsets,
                     first_valid, d_override, rings_, cfg_.scan_period, ring_cloud_.p, ring_intensity_.p, d_ring ? end_ori : nullptr, st, d_n, d_in_table);
  // The two per-ring kernels keep a ring in LDS.  One sweep: sized for the longest ring the handle takes (no host round trip in front of
  // the launch).  A batch: the host reads the rings' lengths first (one small copy + wait per batch, nothing against B sweeps of work)
  // and asks for what the longest ring of THIS batch needs — at HDL-64E ring lengths two workgroups per compute unit instead of one.
  int ring_cap = LIO_PP_MAX_RING_POINTS;
  if (B >= kSizedLdsFromSweeps) {
    int *h_tot = h_state_;   // (the landing zone is idle until the chain's last copy)
    LIO_HIP(hipMemcpyAsync(h_tot, ring_total_.p, size_t(B) * LIO_PP_MAX_RINGS * sizeof(int), hipMemcpyDeviceToHost, s));
    LIO_HIP(hipStreamSynchronize(s));
    int mx = 0;
    for (int k = 0; k < B; ++k)
      for (int r = 0; r < rings_; ++r) mx = std::max(mx, h_tot[size_t(k) * LIO_PP_MAX_RINGS + r]);
    ring_cap = std::min(LIO_PP_MAX_RING_POINTS, (std::max(mx, 64) + 63) & ~63);   // (a longer ring raises the overflow flag, as ever)
  }
  const size_t lds = size_t(8) * PP_SORT_SLOTS * 8 + size_t(ring_cap + 64) * (3 * sizeof(float) + 6) + size_t(8) * PP_WMASK +
                     size_t(8) * PP_WSEL * sizeof(int) + 24 * sizeof(int) + size_t(cap_all) * sizeof(int) + 64;
  hipLaunchKernelGGL(k_ring_pick, dim3(rings_, 1, B), dim3(PP_PICK_THREADS), lds, s, ring_cloud_.p, d_ring_offsets, pc, curv_.p, mask_.p, label_.p,
                     pick_idx_.p, pick_cnt_.p, d_counts, st, ring_cap);
  // less-flat
  const float inv_leaf = 1.0f / cfg_.less_flat_filter_size;
  int sort_cap = 128;
  while (sort_cap < ring_cap) sort_cap <<= 1;
  const size_t lf_lds = size_t(sort_cap) * 8 + size_t(ring_cap) * 16 + size_t(sort_cap) * 2;
  hipLaunchKernelGGL(k_lf_ring, dim3(rings_, 1, B), dim3(PP_LF_THREADS), lf_lds, s, ring_cloud_.p, d_ring_offsets, label_.p, inv_leaf, azi_.p,
                     first_valid, d_override, cfg_.scan_period, lf_tmp_.p, lf_ring_count_.p, st, ring_cap, sort_cap);
  hipLaunchKernelGGL(k_pp_pack, dim3(rings_, 2, B), dim3(256), 0, s, ring_cloud_.p, d_ring_offsets, pick_idx_.p, pick_cnt_.p, pc, class_ring_.p, class_idx_.p,
                     class_cloud_[1].p, class_cloud_[2].p, class_cloud_[3].p, cap_total, lf_tmp_.p, lf_ring_count_.p, less_flat_.p, d_counts, st);
  LIO_HIP(hipGetLastError());
  // results come back through pinned memory: a D2H into pageable memory blocks the host per copy (20 us between the two)
  LIO_HIP(hipMemcpyAsync(h_state_, d_state_.p, size_t(B) * state_stride_ * sizeof(int), hipMemcpyDeviceToHost, s));   // every sweep's counts + ring offsets
  t_begin_ = t_begin;
}

void PointProcessorDev::ProcessFinish() {
  if (!in_flight_) return;
  in_flight_ = false;
  static const bool dbg = std::getenv("LIO_DEBUG_TIMING") != nullptr;
  const auto t_begin = t_begin_;
  LIO_HIP(hipStreamSynchronize(stream_));
  if (dbg) {
    std::fprintf(stderr, "[lio_hip pp timing] process total %.1f us (%d sweep(s))\n",
                 std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count(), nsw_);
    const long long *st = reinterpret_cast<const PPDeviceCounts *>(h_record(0))->pick_stamps;
    std::fprintf(stderr, "[lio_hip pp timing] k_ring_pick ring %d, 10 ns ticks: load %lld, PrepareRing+reach %lld, curvature+keys %lld, sort %lld, picks %lld, lists + mask merge %lld, write-back %lld\n",
                 PP_STAMP_RING, st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4], st[6] - st[5], st[7] - st[6]);
  }
  for (int k = 0; k < nsw_; ++k)
    if (reinterpret_cast<const PPDeviceCounts *>(h_record(k))->overflow) { SelectSweep(0); throw std::runtime_error("PointProcessor: a ring exceeds LIO_PP_MAX_RING_POINTS"); }
  SelectSweep(0);
}

void PointProcessorDev::SelectSweep(int k) {
  if (last_empty_) {   // (a batch of empty sweeps launched nothing)
    std::memset(&counts_, 0, sizeof(counts_));
    std::fill(ring_offsets_.begin(), ring_offsets_.end(), 0);
    return;
  }
  if (k < 0 || k >= nsw_) throw std::runtime_error("PointProcessor: no such sweep in the last batch");
  sel_ = k;
  std::memcpy(&counts_, h_record(k), sizeof(counts_));
  const int *off = h_record(k) + kCountInts;
  std::copy(off, off + rings_ + 1, ring_offsets_.begin());
  counts_.n_ring_points = ring_offsets_[rings_];
}

size_t PointProcessorDev::Count(int which) const {
  switch (which) {
    case LIO_PP_RINGS: return size_t(counts_.n_ring_points);
    case LIO_PP_LESS_FLAT: return size_t(counts_.n_less_flat);
    default: return size_t(counts_.n_class[which]);
  }
}
void PointProcessorDev::GetCloud(int which, float *out) {
  size_t n = Count(which);
  if (!n) return;
  const float4 *src = which == LIO_PP_RINGS ? ring_cloud_.p + sel_ * pts_stride_
                                            : (which == LIO_PP_LESS_FLAT ? less_flat_.p + sel_ * pts_stride_ : class_cloud_[which].p + sel_ * cls_stride_);
  LIO_HIP(hipMemcpyAsync(out, src, n * sizeof(float4), hipMemcpyDeviceToHost, stream_));
  LIO_HIP(hipStreamSynchronize(stream_));
}
void PointProcessorDev::GetIndices(int which, int32_t *ring, int32_t *idx) {
  size_t n = Count(which);
  if (!n) return;
  const size_t base = (size_t(sel_) * 3 + size_t(which - 1)) * cls_stride_;
  LIO_HIP(hipMemcpyAsync(ring, class_ring_.p + base, n * sizeof(int), hipMemcpyDeviceToHost, stream_));
  LIO_HIP(hipMemcpyAsync(idx, class_idx_.p + base, n * sizeof(int), hipMemcpyDeviceToHost, stream_));
  LIO_HIP(hipStreamSynchronize(stream_));
}
void PointProcessorDev::GetRingOffsets(int32_t *out) {
  for (int r = 0; r <= rings_; ++r) out[r] = ring_offsets_[r];
}
void PointProcessorDev::GetRingIntensity(float *out) {
  const size_t n = size_t(counts_.n_ring_points);
  if (!n || !out) return;
  LIO_HIP(hipMemcpyAsync(out, ring_intensity_.p + sel_ * pts_stride_, n * sizeof(float), hipMemcpyDeviceToHost, stream_));
  LIO_HIP(hipStreamSynchronize(stream_));
}
void PointProcessorDev::GetCurvature(float *curv, int32_t *mask) {
  size_t n = size_t(counts_.n_ring_points);
  if (!n) return;
  if (curv) LIO_HIP(hipMemcpyAsync(curv, curv_.p + sel_ * pts_stride_, n * sizeof(float), hipMemcpyDeviceToHost, stream_));
  if (mask) LIO_HIP(hipMemcpyAsync(mask, mask_.p + sel_ * pts_stride_, n * sizeof(int), hipMemcpyDeviceToHost, stream_));
  LIO_HIP(hipStreamSynchronize(stream_));
}

}  // namespace lio
