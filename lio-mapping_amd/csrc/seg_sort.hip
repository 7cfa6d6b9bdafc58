// seg_sort.hip — see seg_sort.h.  gfx950 kernels; blockIdx.y = the segment, blockIdx.x = a tile of it (blocks past a segment's own
// tiles leave at once).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "dev.h"
#include "seg_sort.h"

namespace lio {

#define SS_KEY_NONE_STORED 0x7FFFFFFFu   // batch_kernels.h: BW_KEY_NONE (no point) in the stored absolute key
#define SS_KEY_NONE 0xFFFFFFFFu          // what it sorts as: behind every real key (the passes cover one bit more than the real keys use)

__device__ __forceinline__ uint32_t ss_key(uint32_t k, const KeyLayout *__restrict__ L) {
  if (!L) return k;
  if (k == SS_KEY_NONE_STORED) return SS_KEY_NONE;
  const uint32_t x = (k & 2047u) - uint32_t(L->mx), y = ((k >> 11) & 2047u) - uint32_t(L->my), z = (k >> 22) - uint32_t(L->mz);
  return ((z << L->by) | (y << L->bx) | x) & 0x7FFFFFFFu;   // (at most 31 bits; a window whose extent needs more than the launch's passes order is flagged and re-done by the single-window path)
}

__device__ __forceinline__ void ss_wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// element of (wave, round, lane) inside a tile: a wave owns SS_ITEMS rounds of 64 consecutive elements
template <int THREADS>
__device__ __forceinline__ int ss_index(int tile, int wave, int round, int lane) { return tile * (THREADS * SS_ITEMS) + wave * (64 * SS_ITEMS) + round * 64 + lane; }

template <int THREADS>
__global__ void __launch_bounds__(THREADS) k_ss_hist(const SegDesc *__restrict__ desc, const uint32_t *__restrict__ keys, uint32_t *__restrict__ hist, int shift, int bits,
                                                     const KeyLayout *__restrict__ layout) {
  const SegDesc sg = desc[blockIdx.y];
  constexpr int TILE = THREADS * SS_ITEMS;
  const int ntiles = (sg.n + TILE - 1) / TILE, tile = blockIdx.x;
  if (tile >= ntiles) return;
  __shared__ uint32_t h[1 << SS_MAX_BITS];
  const int nb = 1 << bits;
  for (int d = threadIdx.x; d < nb; d += THREADS) h[d] = 0;
  __syncthreads();
  const KeyLayout *L = layout ? layout + blockIdx.y : nullptr;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t k[SS_ITEMS];
#pragma unroll
  for (int r = 0; r < SS_ITEMS; ++r) {
    const int i = ss_index<THREADS>(tile, wave, r, lane);
    k[r] = i < sg.n ? keys[sg.off + i] : 0u;
  }
#pragma unroll
  for (int r = 0; r < SS_ITEMS; ++r) {
    const int i = ss_index<THREADS>(tile, wave, r, lane);
    if (i < sg.n) atomicAdd(&h[(ss_key(k[r], L) >> shift) & uint32_t(nb - 1)], 1u);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < nb; d += THREADS) hist[size_t(sg.hist_off) + size_t(d) * ntiles + tile] = h[d];
}

// exclusive scan of a segment's (digit, tile) counts in place: one block per segment
#define SS_SCAN_THREADS 1024
__global__ void __launch_bounds__(SS_SCAN_THREADS) k_ss_scan(const SegDesc *__restrict__ desc, uint32_t *__restrict__ hist, int bits, int tile_elems) {
  const SegDesc sg = desc[blockIdx.x];
  const int ntiles = (sg.n + tile_elems - 1) / tile_elems;
  const int E = ntiles << bits;
  if (E <= 0) return;
  uint32_t *h = hist + sg.hist_off;
  const int c = (E + SS_SCAN_THREADS - 1) / SS_SCAN_THREADS;
  const int b0 = threadIdx.x * c, b1 = min(b0 + c, E);
  uint32_t sum = 0;
  for (int i = b0; i < b1; ++i) sum += h[i];
  __shared__ uint32_t ws[SS_SCAN_THREADS / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
  if (lane == 63) ws[wave] = incl;
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < wave; ++w) base += ws[w];
  uint32_t run = base + incl - sum;
  for (int i = b0; i < b1; ++i) { const uint32_t t = h[i]; h[i] = run; run += t; }
}

// LDS of the scatter kernel: the waves' digit counts (NW x bins), the tile's digit starts (bins) and — the staging form — the tile's pairs
// in sorted order.  Staged: the tile is first ordered inside LDS, then written out digit run by digit run with consecutive lanes on
// consecutive addresses; unstaged (1024-thread tiles do not leave room): every lane writes its own element to its final position — 64
// different lines per store.
template <int THREADS, bool STAGED>
__global__ void __launch_bounds__(THREADS) k_ss_scatter(const SegDesc *__restrict__ desc, const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                        uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, const uint32_t *__restrict__ hist, int shift,
                                                        int bits, const KeyLayout *__restrict__ layout) {
  const SegDesc sg = desc[blockIdx.y];
  constexpr int TILE = THREADS * SS_ITEMS, NW = THREADS / 64, NB = 1 << SS_MAX_BITS;
  const int ntiles = (sg.n + TILE - 1) / TILE, tile = blockIdx.x;
  if (tile >= ntiles) return;
  extern __shared__ uint32_t ss_lds[];
  // per wave: elements of each digit seen so far; later the wave's first position of the digit (inside the tile when staged: 16 bits do)
  using WH = typename std::conditional<STAGED, uint16_t, uint32_t>::type;
  WH (*wh)[NB] = reinterpret_cast<WH (*)[NB]>(ss_lds);
  uint32_t *delta = ss_lds + NW * NB * sizeof(WH) / 4;               // STAGED: digit's first output position for this tile minus its first position inside the tile
  uint32_t *sk = delta + NB, *sv = sk + TILE;                        // STAGED: the tile's pairs in sorted order
  const int nb = 1 << bits;
  for (int e = threadIdx.x; e < int(NW * NB * sizeof(WH) / 4); e += THREADS) ss_lds[e] = 0;
  const KeyLayout *L = layout ? layout + blockIdx.y : nullptr;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t key[SS_ITEMS], val[SS_ITEMS];
#pragma unroll
  for (int r = 0; r < SS_ITEMS; ++r) {
    const int i = ss_index<THREADS>(tile, wave, r, lane);
    const bool in = i < sg.n;
    key[r] = in ? keys_in[sg.off + i] : 0u;
    val[r] = in ? (vals_in ? vals_in[sg.off + i] : uint32_t(sg.off + i)) : 0u;
  }
  __syncthreads();
  WH *mine = wh[wave];
  const unsigned long long lt = (1ull << lane) - 1ull;
  uint32_t loc[SS_ITEMS];
#pragma unroll
  for (int r = 0; r < SS_ITEMS; ++r) {
    const int i = ss_index<THREADS>(tile, wave, r, lane);
    const bool in = i < sg.n;
    key[r] = ss_key(key[r], L);
    const uint32_t d = (key[r] >> shift) & uint32_t(nb - 1);
    unsigned long long m = __ballot(in);
    for (int b = 0; b < bits; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long bal = __ballot(bit);
      m &= bit ? bal : ~bal;
    }
    const uint32_t old = in ? mine[d] : 0u;
    ss_wave_lds_sync();                                      // (every lane has read the digit's count before its first lane moves it on)
    if (in && (m & lt) == 0ull) mine[d] = WH(old + uint32_t(__popcll(m)));
    ss_wave_lds_sync();
    loc[r] = old + uint32_t(__popcll(m & lt));
  }
  __syncthreads();
  if constexpr (!STAGED) {
    // a digit's positions: the tile's first (scanned histogram), then wave after wave
    for (int d = threadIdx.x; d < nb; d += THREADS) {
      uint32_t g = hist[size_t(sg.hist_off) + size_t(d) * ntiles + tile];
#pragma unroll
      for (int w = 0; w < NW; ++w) { const uint32_t t = wh[w][d]; wh[w][d] = WH(g); g += t; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SS_ITEMS; ++r) {
      const int i = ss_index<THREADS>(tile, wave, r, lane);
      if (i < sg.n) {
        const uint32_t d = (key[r] >> shift) & uint32_t(nb - 1);
        const size_t pos = size_t(sg.off) + mine[d] + loc[r];
        keys_out[pos] = key[r];
        vals_out[pos] = val[r];
      }
    }
  } else {
    // the tile's digit totals -> exclusive scan over the digits (thread d owns digit d: THREADS >= nb is not required, two per thread at 256)
    constexpr int DPT = (NB + THREADS - 1) / THREADS;
    uint32_t tot[DPT], pre[DPT];
    uint32_t tsum = 0;
#pragma unroll
    for (int q = 0; q < DPT; ++q) {
      const int d = threadIdx.x * DPT + q;
      uint32_t t = 0;
      if (d < nb) for (int w = 0; w < NW; ++w) t += wh[w][d];
      tot[q] = t; pre[q] = tsum; tsum += t;
    }
    uint32_t incl = tsum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    __shared__ uint32_t wsum[NW];
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = incl - tsum;
    for (int w = 0; w < wave; ++w) base += wsum[w];
#pragma unroll
    for (int q = 0; q < DPT; ++q) {
      const int d = threadIdx.x * DPT + q;
      if (d < nb) {
        uint32_t l = base + pre[q];
        delta[d] = hist[size_t(sg.hist_off) + size_t(d) * ntiles + tile] - l;   // (modulo 2^32: position = delta + place inside the tile)
#pragma unroll
        for (int w = 0; w < NW; ++w) { const uint32_t t = wh[w][d]; wh[w][d] = WH(l); l += t; }   // the wave's first position of the digit inside the tile
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SS_ITEMS; ++r) {
      const int i = ss_index<THREADS>(tile, wave, r, lane);
      if (i < sg.n) {
        const uint32_t d = (key[r] >> shift) & uint32_t(nb - 1);
        const uint32_t lp = mine[d] + loc[r];
        sk[lp] = key[r]; sv[lp] = val[r];
      }
    }
    __syncthreads();
    const int tile_n = min(TILE, sg.n - tile * TILE);
    for (int j = threadIdx.x; j < tile_n; j += THREADS) {
      const uint32_t k = sk[j];
      const uint32_t d = (k >> shift) & uint32_t(nb - 1);
      const size_t pos = size_t(sg.off) + uint32_t(delta[d] + uint32_t(j));
      keys_out[pos] = k;
      vals_out[pos] = sv[j];
    }
  }
}

SegSortPlan seg_sort_plan(SegDesc *desc, int nseg, int bits) {
  SegSortPlan p{};
  long long total = 0;
  int max_n = 0;
  for (int k = 0; k < nseg; ++k) { total += desc[k].n; max_n = std::max(max_n, desc[k].n); }
  // bigger tiles (longer runs per digit in the scatter) once the launch fills the chip with them; LIO_SS_THREADS for A/B runs
  p.threads = (total >= (long long)(512 * SS_ITEMS) * 1024) ? 512 : 256;
  { static const int forced = [] { const char *e = std::getenv("LIO_SS_THREADS"); const int v = e ? std::atoi(e) : 0; return (v == 256 || v == 512 || v == 1024) ? v : 0; }(); if (forced) p.threads = forced; }
  const int tile = p.threads * SS_ITEMS;
  size_t h = 0;
  for (int k = 0; k < nseg; ++k) {
    desc[k].hist_off = int(h);
    h += size_t((desc[k].n + tile - 1) / tile) << bits;
  }
  if (h > size_t(INT32_MAX)) throw DeviceError("seg_sort_plan: histogram table beyond 2^31 entries");
  p.hist_entries = h;
  p.max_tiles = (max_n + tile - 1) / tile;
  return p;
}

void seg_sort_pass(const SegDesc *d_desc, int nseg, const SegSortPlan &plan, const uint32_t *keys_in, const uint32_t *vals_in, uint32_t *keys_out, uint32_t *vals_out,
                   uint32_t *hist, int shift, int bits, const KeyLayout *layout, hipStream_t s) {
  if (nseg <= 0 || plan.max_tiles <= 0) return;
  if (bits < 1 || bits > SS_MAX_BITS) throw DeviceError("seg_sort_pass: digit width out of range");
  const dim3 grid(plan.max_tiles, nseg);
  constexpr int NB = 1 << SS_MAX_BITS;
  auto lds_bytes = [&](int threads, bool staged) { return size_t(threads / 64) * NB * (staged ? 2 : 4) + (staged ? size_t(NB) * 4 + size_t(2) * threads * SS_ITEMS * 4 : 0); };
  static const bool attrs = [&] {   // (more than 64 KB of dynamic LDS needs the attribute; per device — one device per process here)
    LIO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ss_scatter<512, true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes(512, true))));
    LIO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ss_scatter<1024, false>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes(1024, false))));
    return true;
  }();
  (void)attrs;
  static const bool unstaged = std::getenv("LIO_SS_UNSTAGED") != nullptr;   // A/B: every lane writes its element to its final position
  if (plan.threads == 1024) {
    hipLaunchKernelGGL(k_ss_hist<1024>, grid, dim3(1024), 0, s, d_desc, keys_in, hist, shift, bits, layout);
    hipLaunchKernelGGL(k_ss_scan, dim3(nseg), dim3(SS_SCAN_THREADS), 0, s, d_desc, hist, bits, 1024 * SS_ITEMS);
    hipLaunchKernelGGL((k_ss_scatter<1024, false>), grid, dim3(1024), lds_bytes(1024, false), s, d_desc, keys_in, vals_in, keys_out, vals_out, hist, shift, bits, layout);
  } else if (plan.threads == 512) {
    hipLaunchKernelGGL(k_ss_hist<512>, grid, dim3(512), 0, s, d_desc, keys_in, hist, shift, bits, layout);
    hipLaunchKernelGGL(k_ss_scan, dim3(nseg), dim3(SS_SCAN_THREADS), 0, s, d_desc, hist, bits, 512 * SS_ITEMS);
    if (unstaged) hipLaunchKernelGGL((k_ss_scatter<512, false>), grid, dim3(512), lds_bytes(512, false), s, d_desc, keys_in, vals_in, keys_out, vals_out, hist, shift, bits, layout);
    else hipLaunchKernelGGL((k_ss_scatter<512, true>), grid, dim3(512), lds_bytes(512, true), s, d_desc, keys_in, vals_in, keys_out, vals_out, hist, shift, bits, layout);
  } else {
    hipLaunchKernelGGL(k_ss_hist<256>, grid, dim3(256), 0, s, d_desc, keys_in, hist, shift, bits, layout);
    hipLaunchKernelGGL(k_ss_scan, dim3(nseg), dim3(SS_SCAN_THREADS), 0, s, d_desc, hist, bits, 256 * SS_ITEMS);
    if (unstaged) hipLaunchKernelGGL((k_ss_scatter<256, false>), grid, dim3(256), lds_bytes(256, false), s, d_desc, keys_in, vals_in, keys_out, vals_out, hist, shift, bits, layout);
    else hipLaunchKernelGGL((k_ss_scatter<256, true>), grid, dim3(256), lds_bytes(256, true), s, d_desc, keys_in, vals_in, keys_out, vals_out, hist, shift, bits, layout);
  }
  LIO_HIP(hipGetLastError());
}

bool seg_sort_host_test(const uint32_t *keys, const uint32_t *vals, size_t n_total, const int *seg_off, const int *seg_n, int nseg, int bits, int passes,
                        uint32_t *keys_out, uint32_t *vals_out) {
  if (!keys || !seg_off || !seg_n || nseg < 1 || bits < 1 || bits > SS_MAX_BITS || passes < 1 || passes * bits > 32 || !keys_out || !vals_out) return false;
  std::vector<SegDesc> desc(static_cast<size_t>(nseg));
  for (int k = 0; k < nseg; ++k) {
    if (seg_off[k] < 0 || seg_n[k] < 0 || size_t(seg_off[k]) + size_t(seg_n[k]) > n_total) return false;
    desc[size_t(k)] = SegDesc{seg_off[k], seg_n[k], 0};
  }
  const SegSortPlan plan = seg_sort_plan(desc.data(), nseg, bits);
  hipStream_t s = nullptr;
  DBuf<uint32_t> k0, k1, v0, v1, hist;
  DBuf<SegDesc> dd;
  const size_t n = std::max<size_t>(n_total, 1);
  k0.reserve(n); k1.reserve(n); v0.reserve(n); v1.reserve(n); hist.reserve(std::max<size_t>(plan.hist_entries, 1)); dd.reserve(size_t(nseg));
  LIO_HIP(hipMemcpy(k0.p, keys, n_total * sizeof(uint32_t), hipMemcpyHostToDevice));
  LIO_HIP(hipMemcpy(k1.p, keys, n_total * sizeof(uint32_t), hipMemcpyHostToDevice));   // (elements outside every segment keep their values in both buffers)
  if (vals) { LIO_HIP(hipMemcpy(v0.p, vals, n_total * sizeof(uint32_t), hipMemcpyHostToDevice)); LIO_HIP(hipMemcpy(v1.p, vals, n_total * sizeof(uint32_t), hipMemcpyHostToDevice)); }
  else { LIO_HIP(hipMemset(v0.p, 0, n * sizeof(uint32_t))); LIO_HIP(hipMemset(v1.p, 0, n * sizeof(uint32_t))); }
  LIO_HIP(hipMemcpy(dd.p, desc.data(), sizeof(SegDesc) * size_t(nseg), hipMemcpyHostToDevice));
  uint32_t *ki = k0.p, *ko = k1.p, *vi = v0.p, *vo = v1.p;
  for (int p = 0; p < passes; ++p) {
    seg_sort_pass(dd.p, nseg, plan, ki, (p == 0 && !vals) ? nullptr : vi, ko, vo, hist.p, p * bits, bits, nullptr, s);
    std::swap(ki, ko); std::swap(vi, vo);
  }
  LIO_HIP(hipDeviceSynchronize());
  LIO_HIP(hipMemcpy(keys_out, ki, n_total * sizeof(uint32_t), hipMemcpyDeviceToHost));
  LIO_HIP(hipMemcpy(vals_out, vi, n_total * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return true;
}

}  // namespace lio
