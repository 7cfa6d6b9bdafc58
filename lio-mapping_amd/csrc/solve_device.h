// solve_device.h — the kernel-side executor of solve_step.h's templates (512- or 256-thread workgroup, LDS, wave shuffles, fp64 MFMA
// tiles), shared by the solve kernels (solve_kernels.hip) and the marginalization kernels (marg_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "solve_step.h"

#if defined(__HIPCC__)
namespace lio {

typedef double v4f64 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double ds_bcast_lane(double v, int src_lane) {   // src_lane: wave-uniform (a constant after unrolling)
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, src_lane);
  hi = __builtin_amdgcn_readlane(hi, src_lane);
  return __hiloint2double(hi, lo);
}

struct DevExec {
  static constexpr bool kDevice = true;
  static constexpr int WT = 64;
  int tid, nthr, lane, wave, nwave;
  __device__ __forceinline__ void sync() const { __syncthreads(); }
  // barrier for exchanges through LDS only: waits for this wave's LDS operations, not for its global stores (__syncthreads() drains
  // vmcnt too — one HBM write round trip, 1-2 us, at every barrier that follows a burst of global stores; the step kernel has eight
  // such bursts per launch and no reader of them inside the launch)
  __device__ __forceinline__ void sync_lds() const { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
  __device__ __forceinline__ double wsum(double v) const {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  }
  __device__ __forceinline__ double wmax(double v) const {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double u = __shfl_xor(v, o, 64); v = u > v ? u : v; }
    return v;
  }
  // 1 / d from the hardware estimate and two Newton steps (<= 1 ulp): a correctly rounded fp64 divide is ~40 instructions on
  // the pivot chain of the factorisation
  __device__ __forceinline__ double rcp(double d) const {
    double y = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    return y;
  }
  // lanes 4m .. 4m+3 hold v0..v3: every one of them gets (v0 + v1) + (v2 + v3)
  __device__ __forceinline__ void stamp(long long *prof, int k) const { if (prof && tid == 0) prof[k] = clock64(); }
  // quad permutes on the data-parallel-primitives path (no LDS crossbar: __shfl_xor is a ds_bpermute, ~100 clocks per 32-bit half)
  template <int CTRL>
  __device__ __forceinline__ static double dpp_quad(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
  }
  __device__ __forceinline__ double pair_sum4(double v) const {
    v += dpp_quad<0xB1>(v);   // quad_perm [1, 0, 3, 2]: lane ^ 1
    v += dpp_quad<0x4E>(v);   // quad_perm [2, 3, 0, 1]: lane ^ 2
    return v;
  }

  // L D L^T of the 16x16 diagonal block at p by ONE wave: lane r (mod 16) keeps row r in registers; per pivot the block's column j
  // goes through 16 doubles of LDS and comes back to every lane as broadcast reads (the first form moved it with 2 x 15 v_readlane
  // per pivot: 10.6 us per block, profiles/r5_c_step_phases.txt).  Then X = L11^-1 (lane c: column c by forward substitution, L11
  // read back from the block as broadcast LDS reads) is left row-major in Xs (one 256-double slot per panel: the back-substitution uses it again): the rows below the block are solved
  // against it on the matrix cores (panel_trsm_mfma).  scr: >= 320 doubles of LDS (exchange 64 | staging 256).
  __device__ __forceinline__ void wave_lds_sync() const {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  __device__ __forceinline__ int panel_factor_regs(double *A, int ld, int p, double *invd, double *scr, double *Xs) const {
    const int r = lane & 15;
    double a[DS_NB];
    const double *row = A + (p + r) * ld + p;
#pragma unroll
    for (int c = 0; c < DS_NB; ++c) a[c] = row[c];
    int ok = 1;
    double myinv = 0.0;
    // Measured on the MI355X (tools/micro/pivot_chain.hip, diag_block.hip; profiles/r5_pivot_chain.txt, r5_f_*): a dependent fp64 FMA 14
    // clocks, rcp + two Newton steps 64, an LDS write -> wave barrier -> read 92.  A pivot is one LDS round trip (column j of the block
    // out, every lane reads what it needs back as broadcast reads) + the reciprocal + a multiply and a fused multiply-add on the chain:
    // 3.5 k clocks for the sixteen.  (With separate multiplies and adds the block cost 11.6 k, with the inverse built inside this loop
    // 16-20 k.)
#pragma unroll
    for (int j = 0; j < DS_NB; ++j) {
      double *bc = scr + (j & 1) * DS_NB;
      if (lane < DS_NB) bc[lane] = a[j];
      wave_lds_sync();
      const double d = bc[j];
      ok &= (d > 0.0) ? 1 : 0;
      const double inv = rcp(d);
      const double l = a[j] * inv;
#pragma unroll
      for (int c = j + 1; c < DS_NB; ++c) a[c] = __builtin_fma(-l, bc[c], a[c]);
      a[j] = (r > j) ? l : a[j];
      myinv = (r == j) ? inv : myinv;
    }
    // the factored rows go to a dense 16 x 16 staging block (unconditional wide stores); from there the lower triangle is copied into
    // A by all 64 lanes (A's strict upper triangle must survive) and the inverse reads L11 with a leading dimension of 16
    double *Ls = scr + 4 * DS_NB;
    if (lane < DS_NB) {
#pragma unroll
      for (int c = 0; c < DS_NB; ++c) Ls[r * DS_NB + c] = a[c];
      invd[p + r] = myinv;
    }
    wave_lds_sync();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = lane + 64 * q, rr = e >> 4, cc = e & 15;
      if (cc <= rr) A[(p + rr) * ld + p + cc] = Ls[e];
    }
    // X = L11^-1 by forward substitution, FOUR lanes per column (column c = lane >> 2; lane part g = lane & 3 sums the terms
    // k = g, g + 4, ...; the four partial sums meet through two quad permutes): 30 multiply-adds per lane instead of 120
    {
      const int c = lane >> 2, g = lane & 3;
      double xcol[DS_NB];
#pragma unroll
      for (int k = 0; k < DS_NB; ++k) xcol[k] = (k == c) ? 1.0 : 0.0;
#pragma unroll
      for (int rr = 1; rr < DS_NB; ++rr) {
        double acc = 0.0;
#pragma unroll
        for (int k4 = 0; k4 < rr; k4 += 4) {
          // term k = k4 + g of this lane (k < rr): the select below keeps the register index static
          double xk = xcol[k4];
          if (k4 + 1 < DS_NB) xk = (g == 1) ? xcol[k4 + 1] : xk;
          if (k4 + 2 < DS_NB) xk = (g == 2) ? xcol[k4 + 2] : xk;
          if (k4 + 3 < DS_NB) xk = (g == 3) ? xcol[k4 + 3] : xk;
          const int k = k4 + g;
          const double lv = (k < rr) ? Ls[rr * DS_NB + k] : 0.0;
          acc = __builtin_fma(lv, xk, acc);
        }
        acc = pair_sum4(acc);
        xcol[rr] = (rr > c) ? -acc : xcol[rr];
      }
      if (g == 0) {
#pragma unroll
        for (int rr = 0; rr < DS_NB; ++rr) Xs[rr * DS_NB + c] = xcol[rr];
      }
    }
    return ok;
  }

  // rows below the block: L21 = (A21 L11^-T) D^-1 on the fp64 matrix cores, one 16-row tile per wave at a time; the right-hand
  // side (one more row) by the wave after the last tile.  v_mfma_f64_16x16x4: lane l supplies A[i = l & 15][k = l >> 4] and
  // B[k = l >> 4][j = l & 15]; D row = (l >> 4) + 4 reg, col = l & 15.
  __device__ __forceinline__ void panel_trsm_mfma(double *A, int ld, int npad, int p, double *gz, const double *invd, const double *Xs) const {
    const int q0 = p + DS_NB, nt = (npad - q0) / DS_NB;
    const int i = lane & 15, kq = lane >> 4;
    for (int t = wave; t <= nt; t += nwave) {
      if (t == nt) {
        double z = 0.0;
#pragma unroll
        for (int k = 0; k < DS_NB; ++k) z += gz[p + k] * Xs[i * DS_NB + k];
        wave_lds_sync();   // (every lane has read the sixteen inputs before any is overwritten)
        if (lane < DS_NB) gz[p + i] = z * invd[p + i];
        continue;
      }
      const int rb = q0 + DS_NB * t;
      v4f64 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int k = 4 * kk + kq;
        const double aop = A[(rb + i) * ld + p + k];
        const double bop = Xs[i * DS_NB + k];   // B[k][j = i] = X[j][k]
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc, 0, 0, 0);
      }
      wave_lds_sync();   // (the tile's inputs have been read: they are overwritten below)
      const double dinv = invd[p + i];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) A[(rb + kq + 4 * rr) * ld + p + i] = acc[rr] * dinv;
    }
  }

  // A22 -= L21 D L21^T on the fp64 matrix cores, one 16x16 tile of the lower triangle per wave at a time.
  // v_mfma_f64_16x16x4: lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]; D row = (l >> 4) + 4 reg, col = l & 15.
  __device__ __forceinline__ void trailing_update_mfma(double *A, int ld, int npad, int p) const {
    const int q0 = p + DS_NB, nb = (npad - q0) / DS_NB;
    const int i = lane & 15, kq = lane >> 4;
    double dk[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) dk[kk] = A[(p + 4 * kk + kq) * ld + p + 4 * kk + kq];
    int t = 0;
    for (int I = 0; I < nb; ++I)
      for (int J = 0; J <= I; ++J, ++t) {
        if (t % nwave != wave) continue;
        const int rb = q0 + DS_NB * I, cb = q0 + DS_NB * J;
        v4f64 acc;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) acc[rr] = A[(rb + kq + 4 * rr) * ld + cb + i];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int k = 4 * kk + kq;
          const double aop = -A[(rb + i) * ld + p + k];
          const double bop = A[(cb + i) * ld + p + k] * dk[kk];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc, 0, 0, 0);
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int row = kq + 4 * rr;
          if (I != J || row >= i) A[(rb + row) * ld + cb + i] = acc[rr];   // a diagonal tile keeps its strict upper triangle
        }
      }
  }

  // x_blk of L11^T x = y for the 16 unknowns at p, y = z - (the rows below): with X = L11^-1 kept from the factorisation x = X^T y is
  // one LDS exchange and sixteen multiply-adds (the substitution it replaces was a chain of fifteen broadcast + multiply-add steps:
  // 1.4 k clocks per block).  Lane j: y_j from the 32 slices (four interleaved sums), then column j of X against y.
  __device__ __forceinline__ void panel_backsolve_regs(double *A, int ld, int p, double *gz, double *part, const double *Xs) const {
    const int j = lane & 15;
    double s4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int sl = 0; sl < 32; ++sl) s4[sl & 3] += part[sl * 16 + j];
    const double y = gz[p + j] - ((s4[0] + s4[1]) + (s4[2] + s4[3]));
    wave_lds_sync();   // (every lane has read the slices before the first sixteen words are reused)
    if (lane < DS_NB) part[j] = y;
    wave_lds_sync();
    double a4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < DS_NB; ++k) a4[k & 3] = __builtin_fma(Xs[k * DS_NB + j], part[k], a4[k & 3]);   // X is lower triangular with zeros above
    if (lane < DS_NB) gz[p + j] = (a4[0] + a4[1]) + (a4[2] + a4[3]);
  }
};

}  // namespace lio
#endif
