// solve_kernels.hip — gfx950 kernels: S_i = sum_k rho'_k z_k z_k^T on v_mfma_f64_16x16x4_f64.
//
// One wave consumes 4 residuals per MFMA: lane l supplies element (l & 15) of residual (l >> 4) as BOTH
// the A operand (A[i = l&15][k = l>>4]) and the B operand (B[k = l>>4][j = l&15]) — the fragment layouts
// of the f64 16x16x4 form coincide for a symmetric rank-4 update, so no LDS transpose is needed.
// The accumulator is 4 f64 per lane: D[row = (l>>4) + 4 r][col = l & 15]  (cdna_hip_programming.md §3:
// the f64 MFMA does NOT use the f32 C/D map).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <mutex>
#include <cstdlib>
#include <string>

#include "solve_kernels.h"
#include "solve_step.h"
#include "solve_device.h"

namespace lio {

#define MOMENT_THREADS 256

// Two kernels compute the same moments: the fp64-MFMA form (default) and a structured fp64-VALU form (73 sums per lane,
// lio_est_config.moments_form = 2 / LIO_MOMENTS=valu).  Measured on the MI355X (tools/batched_moments.py, B windows of the bench
// workload in one launch): round 2 chose the VALU form from four chunks per wave on (4.5 vs 3.4 TB/s algorithmic at B = 512) because
// the MFMA kernel paid the HBM latency once per chunk; with its loads issued three chunks ahead and four blocks per CU the MFMA form
// is level at B = 64 (3.57 vs 3.70 TB/s) and ahead at B = 512 (4.01 vs 3.82), so it is used at every size.
static bool use_mfma(int max_slots, int blocks_per_frame, int form) {
  (void)max_slots; (void)blocks_per_frame;
  return form != 2;
}

int moment_blocks_per_frame(int max_slots) {
  // one residual per lane, 256 residuals per block-iteration; aim for ~2 iterations per block
  int b = cdiv(max_slots, 256 * 2);
  return b < 1 ? 1 : (b > 64 ? 64 : b);
}

// Cauchy weight sqrt(rho') = (1 + s)^-1/2 from the hardware reciprocal-square-root estimate and two Newton steps (<= 2 ulp),
// instead of a correctly rounded divide followed by a correctly rounded sqrt (~100 fp64 instructions at 4 cycles each).
__device__ __forceinline__ double rsqrt_1p(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  y = y * __builtin_fma(-hx * y, y, 1.5);
  y = y * __builtin_fma(-hx * y, y, 1.5);
  return y;
}
// sum of log(1 + s_k) as the log of a running product (renormalised by 2^-400 before it can overflow): one multiply per
// residual and ONE log per lane per launch instead of a ~150-instruction log per residual
struct LogProduct {
  double prod = 1.0, exp2 = 0.0;
  __device__ __forceinline__ void mul(double f) {
    prod *= f;
    const bool big = prod > 0x1p+400;
    prod = big ? prod * 0x1p-400 : prod;
    exp2 += big ? 400.0 : 0.0;
  }
  __device__ __forceinline__ double log_value() const { return log(prod) + exp2 * 0.69314718055994530942; }
};


#define ZROW 17  // 16 doubles per residual + 1 pad: conflict-free ds_write_b64 (lane stride 136 B)

__device__ __forceinline__ void lidar_moments_body(const MomentFrame &fr, const uint8_t *__restrict__ valid, const float4 *__restrict__ coef,
                                                   double *__restrict__ partials, int nblk) {
  // nblk = blocks per frame (gridDim.x of the plain launches; the device-solver launch has a wider grid: k_lidar_moments_dev)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int e = lane & 15, grp = lane >> 4;
  const int waves_total = nblk * (MOMENT_THREADS / 64);
  const int wid = blockIdx.x * (MOMENT_THREADS / 64) + wv;
  // LDS: one transpose buffer per wave; after the chunk loop the same memory carries the wave's 16x16 tile to the block fold
  // (a wave touches only its own slice until the block barrier), so a block needs 34 KB and four of them fit a CU
  __shared__ double zbuf[MOMENT_THREADS / 64][64 * ZROW];
  static_assert(64 * ZROW >= LIO_MOMENT_OUT, "the fold's slice fits the transpose buffer");
  double (*sm)[64 * ZROW] = zbuf;
  double *zb = zbuf[wv];
  v4f64 acc = {0.0, 0.0, 0.0, 0.0};
  double cost = 0.0, cnt = 0.0;
  LogProduct lp;
  // ---- a chunk in two steps.  fetch(): the three loads of a lane's residual, issued THREE chunks ahead of their use — a wave
  // that only starts loading chunk i + 1 when it computes chunk i pays the HBM latency (1-2 us) once per chunk, and with three
  // or four waves per SIMD nothing hides it (the batched launch ran at a quarter of what its vector and matrix issue rate allow).
  // weigh(): residual at the current T_{pivot<-i}, Cauchy weight, scaled z (13 values), branch-free so that it sits in the same
  // basic block as the 16 dependent MFMAs of the chunk before and fills their issue slots.  Arithmetic and order are unchanged.
  struct Raw { float4 po, c; bool ok; };
  auto fetch = [&](int base) {
    Raw r;
    const int sidx = base + lane;
    const bool in = sidx < fr.slot_end;
    const int si = in ? sidx : fr.slot_begin;  // a safe slot to load from when this lane has no residual
    r.ok = in && valid[fr.slot_off + si] != 0;
    r.po = fr.stack[si % fr.M];
    r.c = coef[fr.slot_off + si];
    return r;
  };
  auto weigh = [&](const Raw &rw, double (&z)[16], double &c_add, double &n_add) {
    const bool ok = rw.ok;
    const float4 po = rw.po, c = rw.c;
    const double px = po.x, py = po.y, pz = po.z;
    const double w0 = ok ? double(c.x) : 0.0, w1 = ok ? double(c.y) : 0.0, w2 = ok ? double(c.z) : 0.0, d = ok ? double(c.w) : 0.0;
    const double qx = fr.R[0] * px + fr.R[1] * py + fr.R[2] * pz + fr.t[0];
    const double qy = fr.R[3] * px + fr.R[4] * py + fr.R[5] * pz + fr.t[1];
    const double qz = fr.R[6] * px + fr.R[7] * py + fr.R[8] * pz + fr.t[2];
    const double r = w0 * qx + w1 * qy + w2 * qz + d;
    const double sq = r * r;
    // CauchyLoss(1): rho' = 1/(1+s); rho'' < 0 => alpha = 0 (only sqrt(rho') scaling)
    const double sw = ok ? rsqrt_1p(1.0 + sq) : 0.0;
    const double s0 = sw * w0, s1 = sw * w1, s2 = sw * w2;
    z[0] = s0 * px; z[1] = s0 * py; z[2] = s0 * pz; z[3] = s0;
    z[4] = s1 * px; z[5] = s1 * py; z[6] = s1 * pz; z[7] = s1;
    z[8] = s2 * px; z[9] = s2 * py; z[10] = s2 * pz; z[11] = s2;
    z[12] = sw * d; z[13] = 0.0; z[14] = 0.0; z[15] = 0.0;
    c_add = ok ? 1.0 + sq : 1.0;   // factor of the running product; the log is taken once per lane at the end
    n_add = ok ? 1.0 : 0.0;
  };
  const int stride = waves_total * 64;
  int base = fr.slot_begin + wid * 64;
  if (base < fr.slot_end) {
    Raw r1 = fetch(base + stride), r2 = fetch(base + 2 * stride), r3 = fetch(base + 3 * stride);   // (past the end: the safe slot, ok = false)
    double z[16], c_add, n_add;
    { const Raw r0 = fetch(base); weigh(r0, z, c_add, n_add); }
    for (; base < fr.slot_end; base += stride) {
      lp.mul(c_add); cnt += n_add;
      // ---- transpose through LDS (wave-private rows; LDS executes a wave's DS ops in order)
#pragma unroll
      for (int k = 0; k < 16; ++k) zb[lane * ZROW + k] = z[k];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // ---- next chunk's weights (all lanes idle past the end: zeros) overlap the MFMA chain below; the loads for the chunk
      // four ahead go out now
      weigh(r1, z, c_add, n_add);
      r1 = r2; r2 = r3; r3 = fetch(base + 4 * stride);
      // ---- 16 MFMAs consume the 64 residuals: lane supplies element e of residual 4t+grp as A and B operand
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        double op = zb[(4 * t + grp) * ZROW + e];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(op, op, acc, 0, 0, 0);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  cost = 0.5 * lp.log_value();
#pragma unroll
  for (int r = 0; r < 4; ++r) sm[wv][(grp + 4 * r) * 16 + e] = acc[r];
  for (int off = 32; off > 0; off >>= 1) { cost += __shfl_down(cost, off, 64); cnt += __shfl_down(cnt, off, 64); }
  if (lane == 0) { sm[wv][256] = cost; sm[wv][257] = cnt; }
  __syncthreads();
  double *dst = partials + (size_t(blockIdx.y) * nblk + blockIdx.x) * LIO_MOMENT_OUT;
  for (int k = threadIdx.x; k < 258; k += MOMENT_THREADS) {
    double v = 0;
    for (int w = 0; w < MOMENT_THREADS / 64; ++w) v += sm[w][k];
    dst[k] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// Structured form of the same moments, on the fp64 vector units.  z z^T has only 73 distinct entries:
//   z = sw [w (x) ph ; d],  ph = (p, 1)   =>   S[(a,i),(b,j)] = (s_a s_b)(ph_i ph_j): 6 pairs (a<=b) x 10 pairs (i<=j) = 60,
//   S[(a,i),12] = s_a (ph_i dd): 12,  S[12,12] = dd^2: 1.
// Every lane keeps the 73 sums of ITS residuals in registers (73 FMAs per residual instead of the 16 MFMAs = 512 flop per
// residual of the padded 16x16 product) and the block folds them once at the end through LDS.  Measured on gfx950 the fp64
// MFMA does not overlap fp64 VALU work (both ~78.6 TF peak, and the pipelined MFMA kernel tops out at a third of it), so the
// 3.5x fewer flops win whenever a wave has more than a couple of chunks.
#define LIO_NACC 73
#define RED_ROW (16 * 17 + 1)

// one residual into the 73 per-lane sums (shared by the launch form and the resident form: identical arithmetic)
__device__ __forceinline__ void sym_accumulate(const double *__restrict__ Rm, const double *__restrict__ tv, float fpx, float fpy, float fpz, float4 c, bool ok,
                                               double (&a)[LIO_NACC], LogProduct &lp, double &cnt) {
  const double px = fpx, py = fpy, pz = fpz;
  const double w0 = ok ? double(c.x) : 0.0, w1 = ok ? double(c.y) : 0.0, w2 = ok ? double(c.z) : 0.0, d = ok ? double(c.w) : 0.0;
  const double qx = Rm[0] * px + Rm[1] * py + Rm[2] * pz + tv[0];
  const double qy = Rm[3] * px + Rm[4] * py + Rm[5] * pz + tv[1];
  const double qz = Rm[6] * px + Rm[7] * py + Rm[8] * pz + tv[2];
  const double r = w0 * qx + w1 * qy + w2 * qz + d;
  const double sq = r * r;
  const double sw = ok ? rsqrt_1p(1.0 + sq) : 0.0;
  const double S[3] = {sw * w0, sw * w1, sw * w2};
  const double dd = sw * d;
  lp.mul(ok ? 1.0 + sq : 1.0);
  cnt += ok ? 1.0 : 0.0;
  const double P[10] = {px * px, px * py, px * pz, px, py * py, py * pz, py, pz * pz, pz, 1.0};
  const double W[6] = {S[0] * S[0], S[0] * S[1], S[0] * S[2], S[1] * S[1], S[1] * S[2], S[2] * S[2]};
  const double Q[4] = {px * dd, py * dd, pz * dd, dd};
#pragma unroll
  for (int ab = 0; ab < 6; ++ab)
#pragma unroll
    for (int ij = 0; ij < 10; ++ij) a[ab * 10 + ij] = __builtin_fma(W[ab], P[ij], a[ab * 10 + ij]);
#pragma unroll
  for (int sa = 0; sa < 3; ++sa)
#pragma unroll
    for (int i = 0; i < 4; ++i) a[60 + sa * 4 + i] = __builtin_fma(S[sa], Q[i], a[60 + sa * 4 + i]);
  a[72] = __builtin_fma(dd, dd, a[72]);
}

// block fold of the 73 per-lane sums (+ cost, count) into uniq[0..74], 16 accumulators at a time:
// [16][16 slices of 16 threads, padded to 17] then [16][16 slices].  Ends with a barrier: uniq is readable by every thread.
struct SymFoldLds {
  double red[16 * RED_ROW];
  double red2[16 * 17];
  double uniq[LIO_NACC + 2];
  double cw[MOMENT_THREADS / 64][2];
};
__device__ __forceinline__ void sym_block_fold(const double (&a)[LIO_NACC], const LogProduct &lp, double cnt, SymFoldLds &L) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int kk = tid & 15, sl = tid >> 4;
#pragma unroll
  for (int g = 0; g < (LIO_NACC + 15) / 16; ++g) {
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (g * 16 + k < LIO_NACC) L.red[k * RED_ROW + sl * 17 + kk] = a[g * 16 + k];
    __syncthreads();
    double v = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) v += L.red[kk * RED_ROW + sl * 17 + j];
    L.red2[kk * 17 + sl] = v;
    __syncthreads();
    if (tid < 16 && g * 16 + tid < LIO_NACC) {
      double w = 0.0;
#pragma unroll
      for (int q = 0; q < 16; ++q) w += L.red2[tid * 17 + q];
      L.uniq[g * 16 + tid] = w;
    }
    __syncthreads();
  }
  double cost = 0.5 * lp.log_value();
  for (int off = 32; off > 0; off >>= 1) { cost += __shfl_down(cost, off, 64); cnt += __shfl_down(cnt, off, 64); }
  if (lane == 0) { L.cw[wv][0] = cost; L.cw[wv][1] = cnt; }
  __syncthreads();
  if (tid < 2) {
    double w = 0.0;
    for (int q = 0; q < MOMENT_THREADS / 64; ++q) w += L.cw[q][tid];
    L.uniq[LIO_NACC + tid] = w;
  }
  __syncthreads();
}
// (row, col) of the 16x16 moment matrix -> index into the 73 sums (-1: structural zero)
__host__ __device__ inline int sym_unique_index(int r, int cidx) {
  if (r >= 13 || cidx >= 13) return -1;
  if (r == 12 && cidx == 12) return 72;
  if (r == 12 || cidx == 12) return 60 + (r == 12 ? cidx : r);
  const int ra = r >> 2, ri = r & 3, ca = cidx >> 2, ci = cidx & 3;
  const int a0 = ra < ca ? ra : ca, a1 = ra < ca ? ca : ra, i0 = ri < ci ? ri : ci, i1 = ri < ci ? ci : ri;
  const int ab = a0 == 0 ? a1 : (a0 == 1 ? 2 + a1 : 5);                          // 00 01 02 11 12 22
  const int ij = i0 == 0 ? i1 : (i0 == 1 ? 3 + i1 : (i0 == 2 ? 5 + i1 : 9));     // 00 01 02 03 11 12 13 22 23 33
  return ab * 10 + ij;
}

__device__ __forceinline__ void lidar_moments_sym_body(const MomentFrame &fr, const uint8_t *__restrict__ valid, const float4 *__restrict__ coef,
                                                       double *__restrict__ partials, int nblk) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int waves_total = nblk * (MOMENT_THREADS / 64);
  const int wid = blockIdx.x * (MOMENT_THREADS / 64) + wv;
  double a[LIO_NACC];
#pragma unroll
  for (int k = 0; k < LIO_NACC; ++k) a[k] = 0.0;
  double cnt = 0.0;
  LogProduct lp;
  const int stride = waves_total * 64;
  for (int base = fr.slot_begin + wid * 64; base < fr.slot_end; base += stride) {
    const int sidx = base + lane;
    const bool in = sidx < fr.slot_end;
    const int si = in ? sidx : fr.slot_begin;
    const bool ok = in && valid[fr.slot_off + si] != 0;
    const float4 po = fr.stack[si % fr.M];
    const float4 c = coef[fr.slot_off + si];
    sym_accumulate(fr.R, fr.t, po.x, po.y, po.z, c, ok, a, lp, cnt);
  }
  __shared__ SymFoldLds L;
  sym_block_fold(a, lp, cnt, L);
  double *dst = partials + (size_t(blockIdx.y) * nblk + blockIdx.x) * LIO_MOMENT_OUT;
  {
    // expand the 73 sums into the row-major 16x16 layout the host expects (13x13 used, rest zero)
    const int u = sym_unique_index(tid >> 4, tid & 15);
    dst[tid] = u >= 0 ? L.uniq[u] : 0.0;
    if (tid < 2) dst[256 + tid] = L.uniq[LIO_NACC + tid];
  }
}

// ------------------------------------------------------------------------------------------------
// Resident form: see solve_kernels.h.  Block b = frame * blocks_per_frame + j serves frame `frame` with the lane -> residual map
// of k_lidar_moments_sym at the same blocks per frame (wave w of the frame takes slots slot_begin + 64 w + lane + it * stride).
// one poll of a frame's doorbell record in HBM by lanes 0..15: returns the sequence number both cache lines agree on (NaN while
// they differ); v keeps the lane's slot
__device__ __forceinline__ double door_poll(const double *door, int lane, double &v) {
  if (lane < LIO_RES_DOOR) v = __hip_atomic_load(door + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const double s0 = ds_bcast_lane(v, 7), s1 = ds_bcast_lane(v, 15);
  return s0 == s1 ? s0 : __builtin_nan("");
}

// The relay block: the only poller of host memory.  It waits until EVERY frame's two lines carry the expected sequence number (a
// line is read whole, and the host stores a line's payload before its sequence slot), republishes what that poll read in HBM —
// payload first, acknowledged, then the sequence slots — and goes back to polling.  STOP and the timeout travel the same way.
// A block parks a COMPACT record: the 91 entries of the 13 x 13 upper triangle (row-major), then cost and count — 93 doubles in
// 12 cache lines instead of the padded 16 x 16 tile's 33: the frame fold is bound by the number of lines its agent-scope loads
// keep in flight (measured: ~50 ns per 256-double block record), and the tile is symmetric bit for bit (A = B in the MFMA,
// the same order of additions for every entry), so the lower triangle is the upper one mirrored.
#define RES_NTRI 91
#define RES_NREC 93   // + cost, count
__device__ __forceinline__ int res_tri_index(int a, int b) { return a * 13 - a * (a - 1) / 2 + (b - a); }   // a <= b < 13
// One batch of the frame fold for one lane: NG groups of four blocks starting at block b0, of which this lane takes the two
// blocks 4 g + 2 h and 4 g + 2 h + 1 of every group (h = 0: chains 0 and 1 of k_moment_reduce's order, h = 1: chains 2 and 3) —
// two lanes per value, so a lane has HALF the loads in flight: the fold's time is proportional to the loads per lane (measured:
// ~50 ns per load instruction whatever their kind), not to the bytes.  The loads are unconditional (a block past the end re-reads
// the last one and is masked out of the sum with +0.0, which leaves a chain unchanged bit for bit: the chains start at +0.0 and can
// never be -0.0).  The blocks beyond the last multiple of four continue chain 0, in block order, after its last full group.
#ifndef RES_FOLD_PLAIN
#define RES_FOLD_PLAIN 0   // 1: one agent-scope acquire fence after the flags, then ordinary loads (measured: no faster); 0: agent-scope loads
#endif
__device__ __forceinline__ double resident_fold_load(const double *p) {
  return RES_FOLD_PLAIN ? __builtin_nontemporal_load(p) : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int NG>
__device__ __forceinline__ void resident_fold_batch(const double *src, int h, int b0, int nblk, int b4, double &va, double &vb) {
  double xa[NG], xb[NG], xr[3];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int b = b0 + 4 * g + 2 * h;
    xa[g] = resident_fold_load(src + size_t(min(b, nblk - 1)) * LIO_MOMENT_OUT);
    xb[g] = resident_fold_load(src + size_t(min(b + 1, nblk - 1)) * LIO_MOMENT_OUT);
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) xr[r] = resident_fold_load(src + size_t(min(b4 + r, nblk - 1)) * LIO_MOMENT_OUT);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const bool full = b0 + 4 * g + 4 <= b4;
    va += full ? xa[g] : 0.0; vb += full ? xb[g] : 0.0;
  }
  if (b0 + 4 * NG >= b4) {   // the last batch: chain 0 takes the remainder
#pragma unroll
    for (int r = 0; r < 3; ++r) va += (h == 0 && b4 + r < nblk) ? xr[r] : 0.0;
  }
}
#define RES_RELAY_SLOTS 8   // doubles per lane: LIO_MAX_FRAMES * LIO_RES_DOOR / 64
__device__ __forceinline__ void resident_relay(const MomentArgs &a, const ResidentArgs &ra) {
  const int lane = threadIdx.x & 63;
  if (threadIdx.x >= 64) return;
  const int nd = a.nframes * LIO_RES_DOOR;
  unsigned seq = ra.first_seq;
  const double stop = LIO_RES_STOP(ra.first_seq);
  for (;;) {
    const long long t0 = wall_clock64();
    double v[RES_RELAY_SLOTS];
    double verdict = 0.0;
    for (;;) {
      bool all_seq = true, all_stop = true;
#pragma unroll
      for (int q = 0; q < RES_RELAY_SLOTS; ++q) {
        const int i = q * 64 + lane;
        v[q] = (i < nd) ? __hip_atomic_load(ra.door + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.0;
      }
#pragma unroll
      for (int q = 0; q < RES_RELAY_SLOTS; ++q) {
        const int i = q * 64 + lane;
        const bool is_slot = i < nd && ((i & 7) == 7);
        all_seq = all_seq && __all(!is_slot || v[q] == double(seq));
        all_stop = all_stop && __all(!is_slot || v[q] == stop);
      }
      if (all_seq) { verdict = double(seq); break; }
      if (all_stop) { verdict = stop; break; }
      if (wall_clock64() - t0 > ra.timeout_ticks) {
        verdict = stop;
        if (lane == 0) host_store(ra.words + LIO_MAX_FRAMES, LIO_RES_EXPIRED);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    const double t_detect = double(wall_clock64());   // diagnostics: rides in the unused payload slot 13 of every frame record
    if (lane == 0 && verdict == double(seq)) host_store(ra.words + LIO_MAX_FRAMES + 1, seq);   // echo: lets the host time the inbound leg (debug)
#pragma unroll
    for (int q = 0; q < RES_RELAY_SLOTS; ++q) {
      const int i = q * 64 + lane;
      if (i < nd && (i & 7) != 7) __hip_atomic_store(ra.relay + i, (i & 15) == 13 ? t_detect : v[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < RES_RELAY_SLOTS; ++q) {
      const int i = q * 64 + lane;
      if (i < nd && (i & 7) == 7) __hip_atomic_store(ra.relay + i, verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (verdict == stop) return;
    ++seq;
  }
}

// z (13 values padded to 16), the factor of the running cost product and the count of ONE residual at the pose (Rm, tv): the
// arithmetic of lidar_moments_body's setup, shared with it
__device__ __forceinline__ void moment_z(const double *__restrict__ Rm, const double *__restrict__ tv, float fpx, float fpy, float fpz, float4 c, bool ok,
                                         double (&z)[16], double &c_add, double &n_add) {
  const double px = fpx, py = fpy, pz = fpz;
  const double w0 = ok ? double(c.x) : 0.0, w1 = ok ? double(c.y) : 0.0, w2 = ok ? double(c.z) : 0.0, d = ok ? double(c.w) : 0.0;
  const double qx = Rm[0] * px + Rm[1] * py + Rm[2] * pz + tv[0];
  const double qy = Rm[3] * px + Rm[4] * py + Rm[5] * pz + tv[1];
  const double qz = Rm[6] * px + Rm[7] * py + Rm[8] * pz + tv[2];
  const double r = w0 * qx + w1 * qy + w2 * qz + d;
  const double sq = r * r;
  const double sw = ok ? rsqrt_1p(1.0 + sq) : 0.0;
  const double s0 = sw * w0, s1 = sw * w1, s2 = sw * w2;
  z[0] = s0 * px; z[1] = s0 * py; z[2] = s0 * pz; z[3] = s0;
  z[4] = s1 * px; z[5] = s1 * py; z[6] = s1 * pz; z[7] = s1;
  z[8] = s2 * px; z[9] = s2 * py; z[10] = s2 * pz; z[11] = s2;
  z[12] = sw * d; z[13] = 0.0; z[14] = 0.0; z[15] = 0.0;
  c_add = ok ? 1.0 + sq : 1.0;
  n_add = ok ? 1.0 : 0.0;
}

// The resident kernel: fp64-MFMA form.  Block b = frame * blocks_per_frame + j serves frame `frame` with the wave -> chunk map of
// k_lidar_moments at the same blocks per frame (wave w of the frame takes the 64-slot chunks slot_begin + 64 w + it * stride), the
// chunks of a wave accumulate into its 16x16 MFMA tile in the same order, the four waves of a block are summed in the same order:
// block records equal that kernel's partials bit for bit.
template <int R>
__global__ void __launch_bounds__(MOMENT_THREADS) k_lidar_moments_resident(MomentArgs a, ResidentArgs ra, const uint8_t *__restrict__ valid,
                                                                           const float4 *__restrict__ coef) {
  static_assert(LIO_MAX_FRAMES * LIO_RES_DOOR <= 64 * RES_RELAY_SLOTS, "relay lane slots");
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int e = lane & 15, grp = lane >> 4;
  const int nblk = a.blocks_per_frame;
  // the relay is block 0: the dispatcher places blocks in index order, so the one block every worker depends on is resident first
  if (blockIdx.x == 0) { resident_relay(a, ra); return; }
  const int wb = int(blockIdx.x) - 1;   // worker index
  const int f = wb / nblk, j = wb - f * nblk;
  const MomentFrame &fr = a.fr[f];
  // ---- the wave's residuals, loaded ONCE (features do not change during a solve); chunks past the frame's end stay empty
  float px[R], py[R], pz[R];
  float4 cf[R];
  bool ok[R];
  int nchunk = 0;
  {
    const int stride = nblk * (MOMENT_THREADS / 64) * 64;
    const int base0 = fr.slot_begin + (j * (MOMENT_THREADS / 64) + wv) * 64;
#pragma unroll
    for (int it = 0; it < R; ++it) {
      const int base = base0 + it * stride;
      if (base < fr.slot_end) nchunk = it + 1;
      const int sidx = base + lane;
      const bool in = sidx < fr.slot_end;
      const int si = in ? sidx : fr.slot_begin;
      ok[it] = in && valid[fr.slot_off + si] != 0;
      const float4 po = fr.stack[si % fr.M];
      px[it] = po.x; py[it] = po.y; pz[it] = po.z;
      cf[it] = coef[fr.slot_off + si];
    }
  }
  __shared__ double zbuf[MOMENT_THREADS / 64][64 * ZROW];
  __shared__ double sm[MOMENT_THREADS / 64][LIO_MOMENT_OUT];
  __shared__ double pose[12];
  __shared__ int cmd, go;
  __shared__ long long t_seen;
  __shared__ int n_polls;
  __shared__ double t_relay;
  double *zb = zbuf[wv];
  const double *door = ra.relay + size_t(f) * LIO_RES_DOOR;
  double *mine = ra.block_part + size_t(wb) * LIO_MOMENT_OUT;
  int rec_src = 0;   // where this lane's entry of the compact record sits in the block's 16 x 16 (+ cost, count) LDS record
  if (tid < RES_NTRI) { int a = 0, c = tid; while (c >= 13 - a) { c -= 13 - a; ++a; } rec_src = a * 16 + a + c; }
  else if (tid < RES_NREC) rec_src = 256 + (tid - RES_NTRI);
  const double *frame_part = ra.block_part + size_t(f) * nblk * LIO_MOMENT_OUT;
  unsigned seq = ra.first_seq;
  for (;; ++seq) {
    // ---- wait for the doorbell copy in HBM
    if (wv == 0) {
      const long long t0 = wall_clock64();
      int state = 0, polls = 0;
      double v = 0.0;
      for (;;) {
        const double sq = door_poll(door, lane, v);
        ++polls;
        if (sq == double(seq)) { state = 1; break; }
        if (sq == LIO_RES_STOP(ra.first_seq)) { state = 2; break; }
        if (wall_clock64() - t0 > 2 * ra.timeout_ticks) { state = 2; break; }   // the relay is gone: leave quietly (it reported the timeout)
        __builtin_amdgcn_s_sleep(1);
      }
      if (state == 1) {
        // lanes 0..6 -> R[0..6], lanes 8, 9 -> R[7], R[8], lanes 10..12 -> t
        if (lane < 7) pose[lane] = v;
        else if (lane == 8 || lane == 9) pose[lane - 1] = v;
        else if (lane >= 10 && lane <= 12) pose[lane - 1] = v;
        else if (lane == 13) t_relay = v;
      }
      if (lane == 0) { cmd = state; t_seen = wall_clock64(); n_polls = polls; }
    }
    __syncthreads();
    if (cmd != 1) return;
    double Rm[9], tv[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) Rm[k] = pose[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) tv[k] = pose[9 + k];
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
    double cnt = 0.0;
    LogProduct lp;
#pragma unroll
    for (int it = 0; it < R; ++it) {
      if (it < nchunk) {   // wave-uniform
        double z[16], c_add, n_add;
        moment_z(Rm, tv, px[it], py[it], pz[it], cf[it], ok[it], z, c_add, n_add);
        lp.mul(c_add); cnt += n_add;
#pragma unroll
        for (int k = 0; k < 16; ++k) zb[lane * ZROW + k] = z[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const double op = zb[(4 * t + grp) * ZROW + e];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(op, op, acc, 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
    const long long t_acc = wall_clock64();
    double cost = 0.5 * lp.log_value();
#pragma unroll
    for (int r = 0; r < 4; ++r) sm[wv][(grp + 4 * r) * 16 + e] = acc[r];
    for (int off = 32; off > 0; off >>= 1) { cost += __shfl_down(cost, off, 64); cnt += __shfl_down(cnt, off, 64); }
    if (lane == 0) { sm[wv][256] = cost; sm[wv][257] = cnt; }
    __syncthreads();
    // ---- park the block's record in HBM (agent-scope stores: written through, visible to every XCD); once every wave's stores
    // are acknowledged, the pass number goes into slot 259: the flag the frame's folding block waits for.  No atomics: a ticket
    // drawn by twenty blocks at once serialises on one address.
    // (the record is compact — RES_NREC doubles: upper triangle, cost, count — so that two waves move it in one round)
    if (tid < RES_NREC) {
      double v = 0;
      for (int w = 0; w < MOMENT_THREADS / 64; ++w) v += sm[w][rec_src];
      __hip_atomic_store(mine + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const long long t_parked = wall_clock64();
    if (tid == 0) __hip_atomic_store(mine + LIO_MOMENT_OUT - 1, double(seq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (j != 0) continue;
    // ---- block 0 of the frame folds: wave 0 waits until every block's flag carries this pass, then the block sums the records in
    // k_moment_reduce's order (chain q takes the blocks q, q + 4, ... below the last multiple of four, chain 0 then the remainder,
    // the chains combine as (v0 + v1) + (v2 + v3)), one lane per value, all loads of a lane in flight together.
    if (wv == 0) {
      const long long t0 = wall_clock64();
      int ready = 0;
      for (;;) {
        bool all = true;
        for (int b0 = 0; b0 < nblk; b0 += 64) {
          const int b = b0 + lane;
          const double fl = b < nblk ? __hip_atomic_load(frame_part + size_t(b) * LIO_MOMENT_OUT + LIO_MOMENT_OUT - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                     : double(seq);
          all = all && __all(fl == double(seq));
        }
        if (all) { ready = 1; break; }
        if (wall_clock64() - t0 > ra.timeout_ticks) break;   // a block died: the host's stream query reports it
      }
      if (lane == 0) go = ready;
    }
    __syncthreads();
    if (!go) return;
    const long long t_ready = wall_clock64();
    const int b4 = nblk & ~3;
    // the records were written through to memory by blocks on other XCDs (agent-scope stores) and their flags have been seen:
    // one acquire fence drops whatever this XCD still caches of them from the previous pass, then the loads are ordinary ones
    if (RES_FOLD_PLAIN) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    {
      const int h = tid >> 7, c = tid & 127;   // waves 0, 1: chains 0 and 1; waves 2, 3: chains 2 and 3
      double va = 0.0, vb = 0.0;
      if (c < RES_NREC) {
        const double *src = frame_part + c;
        if (nblk <= 32) resident_fold_batch<8>(src, h, 0, nblk, b4, va, vb);
        else for (int b0 = 0; b0 < nblk; b0 += 64) resident_fold_batch<16>(src, h, b0, nblk, b4, va, vb);
        if (h == 1) { sm[1][c] = va; sm[2][c] = vb; }
      }
      __syncthreads();
      if (h == 0 && c < RES_NREC) sm[0][c] = (va + vb) + (sm[1][c] + sm[2][c]);
    }
    __syncthreads();
    {
      // the host's record is the compact one too (upper triangle, cost, count; the host mirrors it into the padded 16 x 16 tile):
      // every 8-byte system-scope store is a PCIe write of its own, and the 264-double record took ~6 us from "posted" to the
      // host seeing the word against ~3 us for these 93 (measured with the relay's echo: the inbound leg is 2 us)
      double *o = ra.out + size_t(f) * LIO_RES_OUT;
      if (tid < RES_NREC) host_store(o + tid, sm[0][tid]);
    }
    // diagnostics (wall-clock ticks from the doorbell copy seen): accumulated, parked, every flag in, sums formed; polls —
    // the time to "sums formed" always (the bench's per-pass clock), the rest on request
    if (tid == 0) {
      double *dg = ra.out + size_t(f) * LIO_RES_OUT + 258;
      host_store(dg + 3, double(wall_clock64() - t_seen));
      if (ra.diag) {
        host_store(dg + 0, double(t_acc - t_seen)); host_store(dg + 1, double(t_parked - t_seen)); host_store(dg + 2, double(t_ready - t_seen));
        host_store(dg + 4, double(n_polls)); host_store(dg + 5, double(t_seen) - t_relay);
      }
    }
    host_signal_drain();
    __syncthreads();
    if (tid == 0) host_store(ra.words + f, seq);
  }
}

__global__ void __launch_bounds__(MOMENT_THREADS) k_lidar_moments_sym(MomentArgs a, const uint8_t *__restrict__ valid,
                                                                      const float4 *__restrict__ coef, double *__restrict__ partials) {
  lidar_moments_sym_body(a.fr[blockIdx.y], valid, coef, partials, gridDim.x);
}
__global__ void __launch_bounds__(MOMENT_THREADS) k_lidar_moments_sym_batched(const MomentFrame *__restrict__ frames, const uint8_t *__restrict__ valid,
                                                                              const float4 *__restrict__ coef, double *__restrict__ partials) {
  lidar_moments_sym_body(frames[blockIdx.y], valid, coef, partials, gridDim.x);
}

__global__ void __launch_bounds__(MOMENT_THREADS) k_lidar_moments(MomentArgs a, const uint8_t *__restrict__ valid,
                                                                  const float4 *__restrict__ coef, double *__restrict__ partials) {
  lidar_moments_body(a.fr[blockIdx.y], valid, coef, partials, gridDim.x);
}

// Batched form for B windows in flight: the frame descriptors live in device memory (B x Wo of them), everything else is the
// same code.  Used by the batched roofline measurement (SURVEY.md §8d ii) and by multi-window hosts.
__global__ void __launch_bounds__(MOMENT_THREADS) k_lidar_moments_batched(const MomentFrame *__restrict__ frames, const uint8_t *__restrict__ valid,
                                                                          const float4 *__restrict__ coef, double *__restrict__ partials) {
  lidar_moments_body(frames[blockIdx.y], valid, coef, partials, gridDim.x);
}

// Fold of the per-block partials: out[f][k] = sum_b partials[f][b][k].  Four lanes per value, lane q walks the blocks b = q, q + 4, ...
// with all of its loads in flight (the remainder beyond the last multiple of four goes to lane 0), then (v0 + v1) + (v2 + v3)
// by two xor-shuffles: the same additions in the same order as one lane with four interleaved chains, at a quarter of the
// dependent-load depth (4.9 -> 2.x us at 39 blocks per frame).  Grid (frames, 3), 384 threads: 96 values per block.
#define REDUCE_THREADS 384
__global__ void __launch_bounds__(REDUCE_THREADS) k_moment_reduce(const double *__restrict__ partials, int bpf, double *__restrict__ out, HostSignal sig) {
  const int k = blockIdx.y * (REDUCE_THREADS / 4) + (threadIdx.x >> 2), q = threadIdx.x & 3;
  const bool in = k < 258;
  const double *src = partials + size_t(blockIdx.x) * bpf * LIO_MOMENT_OUT + (in ? k : 0);
  const int b4 = bpf & ~3;
  double v = 0;
  int b = q;
  for (; b + 12 < b4; b += 16) {   // four loads in flight per lane
    const double x0 = src[size_t(b) * LIO_MOMENT_OUT], x1 = src[size_t(b + 4) * LIO_MOMENT_OUT], x2 = src[size_t(b + 8) * LIO_MOMENT_OUT],
                 x3 = src[size_t(b + 12) * LIO_MOMENT_OUT];
    v += x0; v += x1; v += x2; v += x3;
  }
  for (; b < b4; b += 4) v += src[size_t(b) * LIO_MOMENT_OUT];
  if (q == 0) for (int r = b4; r < bpf; ++r) v += src[size_t(r) * LIO_MOMENT_OUT];
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  if (in && q == 0) {
    if (sig.flag) host_store(&out[size_t(blockIdx.x) * LIO_MOMENT_OUT + k], v);   // `out` is coherent host memory then
    else out[size_t(blockIdx.x) * LIO_MOMENT_OUT + k] = v;
  }
  if (sig.flag) {   // this block's values are out: post its completion word (dev.h: HostSignal)
    host_signal_drain();
    __syncthreads();
    if (threadIdx.x == 0) post_host_signal(sig, int(blockIdx.x * gridDim.y + blockIdx.y));
  }
}

// Worker blocks of the resident form that the CURRENT device keeps co-resident beside the relay: occupancy of the kernel x compute
// units - 1, never more than LIO_RES_MAX_BLOCKS (the size of the record arrays).  0 when the device cannot be asked (no GPU) —
// the resident form is then never chosen.  A partitioned (CPX) or smaller part gets a smaller figure and with it the launch path
// for windows that do not fit; nothing is assumed about "256 CUs".
int resident_max_blocks(int per_lane) {
  // cached per (device, residuals per lane): a process may drive devices of different sizes or partition modes, and several
  // estimators may ask at once (-1 = not asked yet; racing first calls compute the same value)
  static std::atomic<int> cache[16][9];
  static std::atomic<bool> init{false};
  if (per_lane < 1 || per_lane > 8) return 0;
  if (!init.load(std::memory_order_acquire)) {
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (!init.load(std::memory_order_relaxed)) {
      for (auto &row : cache) for (auto &v : row) v.store(-1, std::memory_order_relaxed);
      init.store(true, std::memory_order_release);
    }
  }
  int dev = 0, cus = 0, per_cu = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
  const bool cached = dev >= 0 && dev < 16;
  if (cached) { const int v = cache[dev][per_lane].load(std::memory_order_relaxed); if (v >= 0) return v; }
  e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (e == hipSuccess) {
    switch (per_lane) {
      case 1: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_lidar_moments_resident<1>, MOMENT_THREADS, 0); break;
      case 2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_lidar_moments_resident<2>, MOMENT_THREADS, 0); break;
      case 4: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_lidar_moments_resident<4>, MOMENT_THREADS, 0); break;
      case 8: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_lidar_moments_resident<8>, MOMENT_THREADS, 0); break;
      default: per_cu = 0; break;
    }
  }
  int v = 0;
  if (e != hipSuccess) (void)hipGetLastError();
  else v = std::max(0, std::min(LIO_RES_MAX_BLOCKS, cus * per_cu - 1));
  if (cached) cache[dev][per_lane].store(v, std::memory_order_relaxed);
  return v;
}

int resident_blocks_per_frame(int max_slots, int nframes, int per_lane) {
  if (max_slots <= 0 || nframes <= 0 || per_lane < 1) return 0;
  const int b = cdiv(max_slots, MOMENT_THREADS * per_lane);
  return (b * nframes <= resident_max_blocks(per_lane)) ? b : 0;
}

void launch_lidar_moments_resident(const MomentArgs &a, const ResidentArgs &ra, int per_lane, const uint8_t *valid, const float4 *coef, hipStream_t s) {
  const dim3 grid(a.blocks_per_frame * a.nframes + 1), block(MOMENT_THREADS);   // the relay (block 0) + the workers
  switch (per_lane) {
    case 1: hipLaunchKernelGGL(k_lidar_moments_resident<1>, grid, block, 0, s, a, ra, valid, coef); break;
    case 2: hipLaunchKernelGGL(k_lidar_moments_resident<2>, grid, block, 0, s, a, ra, valid, coef); break;
    case 4: hipLaunchKernelGGL(k_lidar_moments_resident<4>, grid, block, 0, s, a, ra, valid, coef); break;
    case 8: hipLaunchKernelGGL(k_lidar_moments_resident<8>, grid, block, 0, s, a, ra, valid, coef); break;
    default: throw DeviceError("resident moments: per_lane must be 1, 2, 4 or 8");
  }
  LIO_HIP(hipGetLastError());
}

// Inside a batch the partition of a window's factor slots must not depend on the batch (a window gives the same bits alone and in
// any company): 2048 slots per block — eight 64-slot chunks per wave, enough for the chunk loop's three-deep prefetch to fill the
// MFMA issue slots — and never more than 64 blocks per frame.
int batch_blocks_per_frame(int max_slots) {
  // slots per block: a function of the window's own size only (bit-identity of a window alone and in a batch); LIO_BW_SLOTS_PER_BLOCK for A/B runs
  static const int per_block = [] { const char *e = std::getenv("LIO_BW_SLOTS_PER_BLOCK"); const int v = e ? std::atoi(e) : 0; return v >= 256 ? v : MOMENT_THREADS * 16; }();   // 4096: measured against 1024 / 2048 / 8192 / 16384 at 64 and 512 windows (profiles/r5_l_*): fewer partials for the step kernel's fold, still 640 blocks per 32 windows
  return std::max(1, std::min(cdiv(max_slots, per_block), 64));
}

int moment_blocks_per_frame_batched(int max_slots, int nframes) {
  // enough waves to fill the chip about four times over, each with as many chunks as possible (the chunk loop is software
  // pipelined: more chunks per wave = more MFMA issue slots filled with the next chunk's setup)
  const int want = std::max(1, 2048 / std::max(nframes, 1));
  return std::max(1, std::min(want, moment_blocks_per_frame(max_slots)));
}

void launch_lidar_moments_batched(const MomentFrame *d_frames, int nframes, int blocks_per_frame, int max_slots, const uint8_t *valid,
                                  const float4 *coef, double *partials, double *out, hipStream_t s, int form) {
  if (nframes <= 0) return;
  if (use_mfma(max_slots, blocks_per_frame, form))
    hipLaunchKernelGGL(k_lidar_moments_batched, dim3(blocks_per_frame, nframes), dim3(MOMENT_THREADS), 0, s, d_frames, valid, coef, partials);
  else
    hipLaunchKernelGGL(k_lidar_moments_sym_batched, dim3(blocks_per_frame, nframes), dim3(MOMENT_THREADS), 0, s, d_frames, valid, coef, partials);
  hipLaunchKernelGGL(k_moment_reduce, dim3(nframes, 3), dim3(REDUCE_THREADS), 0, s, partials, blocks_per_frame, out, HostSignal());
  LIO_HIP(hipGetLastError());
}

void launch_lidar_moments(const MomentArgs &a, const uint8_t *valid, const float4 *coef, double *partials, double *out, hipStream_t s,
                          const HostSignal &sig) {
  if (a.nframes <= 0) return;
  int max_slots = 0;
  for (int k = 0; k < a.nframes; ++k) max_slots = std::max(max_slots, a.fr[k].slot_end - a.fr[k].slot_begin);
  if (use_mfma(max_slots, a.blocks_per_frame, a.form))
    hipLaunchKernelGGL(k_lidar_moments, dim3(a.blocks_per_frame, a.nframes), dim3(MOMENT_THREADS), 0, s, a, valid, coef, partials);
  else
    hipLaunchKernelGGL(k_lidar_moments_sym, dim3(a.blocks_per_frame, a.nframes), dim3(MOMENT_THREADS), 0, s, a, valid, coef, partials);
  hipLaunchKernelGGL(k_moment_reduce, dim3(a.nframes, 3), dim3(REDUCE_THREADS), 0, s, partials, a.blocks_per_frame, out, sig);
  LIO_HIP(hipGetLastError());
}

// ================================================================================================
// Device-resident dogleg (solve_step.h): the two launches of one iteration (the kernel-side executor DevExec: solve_device.h)
// ================================================================================================
// Launch A of an iteration, for every window of a batch, is two kernels (one kernel holding both bodies takes the register count of
// the larger — 256 VGPRs for the factor code against 107 for the moments — and the moments pass then runs at a quarter of its
// occupancy: 252 us against 90 at 64 windows, profiles/r5_a_batch64_first_kernel_stats.md):
//   k_bw_aux       grid (Wo + 1, windows): everything that depends on the candidate but not on the points — block i < Wo the ImuFactor
//                  between optimised frames i and i + 1 and the lidar linear map of frame i + 1, block Wo the marginalization
//                  prior and the extrinsic prior;
//   k_bw_moments   grid (bpf, Wo, windows): the moments of frames 1 .. Wo at the candidate's T_{pivot<-i}, read from the
//                  device-resident state.
// A window that is done (converged, or handed back to the host) costs its blocks one load.
__global__ void __launch_bounds__(MOMENT_THREADS) k_bw_moments(const BatchSolve *__restrict__ bs, BatchBases bb, const uint8_t *__restrict__ valid,
                                                               const float4 *__restrict__ coef) {
  const BatchSolve &S = bs[blockIdx.z];
  const DevState *st = rebase(bb.st, S.st);
  if (!S.active || st->done) return;
  if (int(blockIdx.y) >= S.nframes || int(blockIdx.x) >= S.bpf) return;
  MomentFrame fr = S.fr[blockIdx.y];
  fr.stack = rebase(coef, fr.stack);
  const double *Rt = st->cand_Rt[blockIdx.y];
#pragma unroll
  for (int k = 0; k < 9; ++k) fr.R[k] = Rt[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) fr.t[k] = Rt[9 + k];
  lidar_moments_body(fr, valid, coef, rebase(bb.partials, S.partials), S.bpf);
}
__global__ void __launch_bounds__(MOMENT_THREADS) k_bw_aux(const BatchSolve *__restrict__ bs, BatchBases bb) {
  const BatchSolve &S = bs[blockIdx.y];
  const DevState *st = rebase(bb.st, S.st);
  if (!S.active || st->done) return;
  const int Wo = S.nframes, i = blockIdx.x;
  if (i > Wo) return;
  __shared__ double aux_lds[1536];
  const DevExec x{int(threadIdx.x), int(blockDim.x), int(threadIdx.x & 63), int(threadIdx.x >> 6), int(blockDim.x >> 6)};
  const DevParams &P = st->cand;
  const DevProblem *pb = rebase(bb.pb, S.pb);
  if (i < Wo) {
    long long *prof = i == 0 ? rebase(bb.slab, S.prof) : nullptr;
    x.stamp(prof, 64);
    aux_imu(x, pb->pim[i], P.pose[i], P.sb[i], P.pose[i + 1], P.sb[i + 1], rebase(bb.slab, S.imu_out) + size_t(i) * DS_IMU_OUT, aux_lds, prof);
    __syncthreads();
    x.stamp(prof, 67);
    aux_lmap(x, P.pose[0], P.pose[i + 1], P.ex, rebase(bb.slab, S.lmap) + size_t(i) * DS_LMAP_OUT, aux_lds);
    __syncthreads();
    x.stamp(prof, 68);
  } else {
    if (pb->have_prior) aux_prior(x, *pb, rebase(bb.slab, S.prior_mats), P, rebase(bb.slab, S.prior_out), aux_lds);
    if (pb->use_ex_prior) aux_exprior(x, *pb, P, rebase(bb.slab, S.exprior_out));
  }
}

// Launch B: one workgroup per window (solve_step.h)
__global__ void __launch_bounds__(DS_THREADS) k_bw_solve_step(const BatchSolve *__restrict__ bs, BatchBases bb) {
  extern __shared__ __attribute__((aligned(16))) double ds_lds[];
  const BatchSolve &S = bs[blockIdx.x];
  if (!S.active) return;
  const DevExec x{int(threadIdx.x), int(blockDim.x), int(threadIdx.x & 63), int(threadIdx.x >> 6), int(blockDim.x >> 6)};
  StepBuffers B{rebase(bb.slab, S.prior_mats), rebase(bb.partials, S.partials), rebase(bb.slab, S.imu_out), rebase(bb.slab, S.lmap), rebase(bb.slab, S.prior_out), rebase(bb.slab, S.exprior_out),
                rebase(bb.slab, S.Hcur), rebase(bb.slab, S.S_buf), rebase(bb.slab, S.prof)};
  solve_step(x, *rebase(bb.pb, S.pb), *rebase(bb.st, S.st), B, ds_lds);
}

// Test hook (lio_ldlt_solve): the LDS-resident blocked L D L^T + back-substitution of launch B on its own
__global__ void __launch_bounds__(DS_THREADS) k_ldlt_test(const double *__restrict__ Ain, const double *__restrict__ b, int n, int npad,
                                                         double *__restrict__ xout, int *__restrict__ ok_out) {
  extern __shared__ __attribute__((aligned(16))) double ds_lds[];
  const DevExec x{int(threadIdx.x), int(blockDim.x), int(threadIdx.x & 63), int(threadIdx.x >> 6), int(blockDim.x >> 6)};
  const int ld = npad + 1;
  double *A = ds_lds, *gz = A + size_t(npad) * ld, *invd = gz + npad, *part = invd + npad;
  double *xinv = part + DS_PART;
  int *flag = reinterpret_cast<int *>(xinv + size_t(npad) * DS_NB);
  for (int e = x.tid; e < npad * npad; e += x.nthr) {
    const int r = e / npad, c = e % npad;
    A[size_t(r) * ld + c] = (r < n && c < n) ? Ain[size_t(r) * n + c] : (r == c ? 1.0 : 0.0);
  }
  for (int i = x.tid; i < npad; i += x.nthr) gz[i] = i < n ? b[i] : 0.0;
  x.sync();
  const int ok = ds_ldlt_solve(x, A, ld, npad, gz, invd, part, xinv, flag);
  if (x.tid == 0) *ok_out = ok;
  if (ok) for (int i = x.tid; i < n; i += x.nthr) xout[i] = gz[i];
  // the strict upper triangle must have survived (launch B reads H from it after the factorisation)
  if (ok) for (int e = x.tid; e < n * n; e += x.nthr) { const int r = e / n, c = e % n; if (c > r && A[size_t(r) * ld + c] != Ain[size_t(r) * n + c]) *ok_out = -1; }
}
int ldlt_solve_device(const double *A, const double *b, int n, double *xh, hipStream_t s) {
  const int npad = (n + DS_NB - 1) / DS_NB * DS_NB;
  const size_t lds = (size_t(npad) * (npad + 1) + 2 * size_t(npad) + DS_PART + size_t(npad) * DS_NB + 8) * sizeof(double);
  if (n < 1 || lds > 160 * 1024) return -2;
  static const bool attr_set = [] {
    LIO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ldlt_test), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return true;
  }();
  (void)attr_set;
  DBuf<double> dA, db, dx; DBuf<int> dok;
  dA.reserve(size_t(n) * n); db.reserve(n); dx.reserve(n); dok.reserve(1);
  LIO_HIP(hipMemcpyAsync(dA.p, A, sizeof(double) * n * n, hipMemcpyHostToDevice, s));
  LIO_HIP(hipMemcpyAsync(db.p, b, sizeof(double) * n, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_ldlt_test, dim3(1), dim3(DS_THREADS), lds, s, dA.p, db.p, n, npad, dx.p, dok.p);
  LIO_HIP(hipGetLastError());
  int ok = 0;
  LIO_HIP(hipMemcpyAsync(&ok, dok.p, sizeof(int), hipMemcpyDeviceToHost, s));
  LIO_HIP(hipMemcpyAsync(xh, dx.p, sizeof(double) * n, hipMemcpyDeviceToHost, s));
  LIO_HIP(hipStreamSynchronize(s));
  return ok;
}

void launch_bw_aux(const BatchSolve *bs, const BatchBases &bb, int B, int max_wo, int forced, hipStream_t s) {
  if (B <= 0) return;
  // Threads per block follow the size of the launch (the results do not depend on it: every sum of the aux row has a fixed order).
  // The kernel holds 256 VGPRs, so a CU runs two 256-thread blocks or eight one-wave blocks: at 256 windows per launch the 1536
  // blocks took 334 us in six rounds (profiles/r5_final3_batch512_kernel_stats.md); one wave per block keeps them all resident.
  // A small launch is a latency chain and wants the four serial jobs of an IMU factor on four waves.
  const int threads = (forced == 64 || forced == 128 || forced == 256) ? forced : (B >= 128 ? 64 : MOMENT_THREADS);
  hipLaunchKernelGGL(k_bw_aux, dim3(max_wo + 1, B), dim3(threads), 0, s, bs, bb);
}
void launch_bw_moments(const BatchSolve *bs, const BatchBases &bb, int B, int max_bpf, int max_wo, const uint8_t *valid, const float4 *coef, hipStream_t s) {
  if (B <= 0) return;
  hipLaunchKernelGGL(k_bw_moments, dim3(max_bpf, max_wo, B), dim3(MOMENT_THREADS), 0, s, bs, bb, valid, coef);
}
// the step kernel's dynamic-LDS limit on the CURRENT device (hipFuncSetAttribute is per device): every EstimatorBatch calls it for its own
void prepare_bw_step_kernel() {
  LIO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bw_solve_step), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
}
void launch_bw_step(const BatchSolve *bs, const BatchBases &bb, int B, int max_wo, int max_npad, hipStream_t s) {
  if (B <= 0) return;
  const size_t lds = ds_lds_doubles(max_npad, max_wo) * sizeof(double);   // (above 64 KB: prepare_bw_step_kernel on this device, EstimatorBatch's constructor)
  hipLaunchKernelGGL(k_bw_solve_step, dim3(B), dim3(DS_THREADS), lds, s, bs, bb);
  LIO_HIP(hipGetLastError());
}
void launch_bw_solve_iteration(const BatchSolve *bs, const BatchBases &bb, int B, int max_bpf, int max_wo, int max_npad, int aux_threads, const uint8_t *valid, const float4 *coef,
                               hipStream_t s) {
  launch_bw_aux(bs, bb, B, max_wo, aux_threads, s);
  launch_bw_moments(bs, bb, B, max_bpf, max_wo, valid, coef, s);
  launch_bw_step(bs, bb, B, max_wo, max_npad, s);
}

}  // namespace lio
