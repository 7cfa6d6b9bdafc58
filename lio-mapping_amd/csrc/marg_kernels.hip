// marg_kernels.hip — see marg_kernels.h.  One workgroup; every matrix lives in LDS.
//
//   Amm   = 1/2 (A[:m,:m] + A[:m,:m]^T)                     (MarginalizationFactor.cc:275)
//   Amm^+ = V diag(1 / s_i if s_i > eps else 0) V^T          (:276-283, SelfAdjointEigenSolver)
//   T     = Arm Amm^+,  S = Arr - T Amr,  bs = br - T bm     (:285-291; the n x m x n product runs on v_mfma_f64_16x16x4)
//   S     = V2 diag(s) V2^T                                  (:293)
//   linearized_jacobians = diag(sqrt(s_k > eps ? s_k : 0)) V2^T,  linearized_residuals = diag(1/sqrt(s_k) or 0) V2^T bs   (:294-302)
//
// The eigensolver is the Householder + implicit-QL pair (tridiag_ql_lds below): the method of the host path and of the reference's
// Eigen::SelfAdjointEigenSolver.  Which eigenvalues of the badly graded S (entries from 1e9 down to 1e-4) fall on which side of the
// absolute 1e-8 cut is rounding noise on every such path (tests/golden/README.md); the tests pin the invariants (J^T J, J^T r, r^T r, the
// kept spectrum), not single eigenvectors.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "dev.h"
#include "marg_kernels.h"
#include "solve_device.h"

namespace lio {

#ifndef MARG_THREADS
#define MARG_THREADS 512       // the Schur / eigen kernels: latency chains through LDS, two waves per SIMD to hide them
#endif
#define MARG_AUX_THREADS 256   // the aux row (256 VGPRs per lane: one wave per SIMD)

// Symmetric eigendecomposition in LDS by one workgroup: Householder reduction to tridiagonal form, then the implicit QL iteration with
// Wilkinson-type shifts — the tred2 / tql2 pair, i.e. the method of the host path (hlinalg.h: sym_eig) and of the reference's
// Eigen::SelfAdjointEigenSolver (MarginalizationFactor.cc:276, :293).  Rounds 5-6 ran a cyclic Jacobi here (n / 2 rotations per step, all
// threads busy): 12 sweeps x 45 steps x 3.1 k clocks = 1.7 M clocks at n = 45 however it was tuned (profiles/r6_b_marg_phases.txt) — the
// work is O(sweeps x n^3) behind 1 080 block barriers.  This form does O(n^3) once:
//   reduction   n - 2 reflectors H = I - u u^T / h, each three barriers: the reflector (one wave: two sums, a square root);  A u and Z u
//               (8 lanes per row);  the rank-2 update of A's leading block and the rank-1 update of Z = H_{n-1} .. H_2 — accumulated as
//               the reduction goes, not in a second pass;
//   QL          a serial recurrence (the rotations of one sweep depend on each other through p, c, s): wave 0 runs it with the diagonal
//               and sub-diagonal in REGISTERS (lane k holds d_k, e_k, d_{k+64}, e_{k+64}; element i is a v_readlane with a uniform
//               index, no LDS round trip on the chain) and leaves the sweep's (c, s) pairs in LDS; the other waves apply the PREVIOUS
//               sweep's rotations to Z (one thread per row, the running column in a register) at the same time.  One barrier per sweep.
// A: n x n (leading dimension ld), symmetric on entry, destroyed; its diagonal holds the eigenvalues on return (not sorted).
// Z: n x n (ld), eigenvectors as COLUMNS (column k belongs to A[k][k]).  work: 9 n + 16 doubles.  n <= 128.  Returns the number of QL sweeps.
// wave-wide sum on DPP moves and four v_readlane (a shuffle through the LDS crossbar costs ~120 clocks per step, and two of these sums
// sit on the critical path of every reflector): quad swaps, two rotations inside the rows of 16, then the four row sums
template <int CTRL> __device__ __forceinline__ double eig_dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double eig_quad_sum(double v) {   // every lane: the sum over its quad
  v += eig_dpp<0xB1>(v);    // quad_perm [1, 0, 3, 2]
  v += eig_dpp<0x4E>(v);    // quad_perm [2, 3, 0, 1]
  return v;
}
__device__ __forceinline__ double eig_wsum(double v) {
  v = eig_quad_sum(v);
  v += eig_dpp<0x124>(v);   // row_ror:4
  v += eig_dpp<0x128>(v);   // row_ror:8
  return (ds_bcast_lane(v, 0) + ds_bcast_lane(v, 16)) + (ds_bcast_lane(v, 32) + ds_bcast_lane(v, 48));
}
__device__ __forceinline__ double eig_rsqrt(double d) {  // 1 / sqrt(d): hardware estimate + two Newton steps (<= 1 ulp) — a correctly rounded divide and square root are ~80 instructions on the serial chain
  double y = __builtin_amdgcn_rsq(d);
  const double h = 0.5 * d;
  y = y * __builtin_fma(-h * y, y, 1.5);
  return y * __builtin_fma(-h * y, y, 1.5);
}
__device__ __forceinline__ double eig_rcp(double d) {   // 1 / d likewise
  double y = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-d, y, 1.0);
  return __builtin_fma(y, e, y);
}
__device__ __forceinline__ void eig_wave_sync() {   // orders this wave's LDS traffic for the compiler (the hardware runs a wave's DS operations in order)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ int tridiag_ql_lds(double *A, double *Z, int n, int ld, double *work, double *prof = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = MARG_THREADS / 64;
  const long long t_in = clock64();
  double *u = work, *praw = work + n, *zv = work + 2 * n, *cs = work + 3 * n /* 2 x 2 n */, *ee = work + 7 * n, *dd = work + 8 * n, *sc = work + 9 * n;
  int *cmd = reinterpret_cast<int *>(sc + 4);   // 2 x (l, m, done, -)
  for (int r = wave; r < n; r += NW)
    for (int c = lane; c < n; c += 64) Z[r * ld + c] = r == c ? 1.0 : 0.0;
  if (tid < n) ee[tid] = 0.0;
  __syncthreads();
  // ---------------- reduction to tridiagonal form, rows n - 1 .. 2
  for (int i = n - 1; i >= 2; --i) {
    if (wave == 0) {
      const double *row = A + i * ld;
      double p_all = 0.0, p_excl = 0.0;
      for (int k = lane; k < i; k += 64) { const double x = row[k]; p_all += x * x; p_excl += (k != i - 1) ? x * x : 0.0; }
      const double sigma = eig_wsum(p_all), excl = eig_wsum(p_excl);
      const double f = row[i - 1];
      if (excl == 0.0) {   // nothing left of the sub-diagonal: no reflector
        if (lane == 0) { sc[2] = 0.0; ee[i] = f; }
      } else {
        const double rs = eig_rsqrt(sigma), nrm = sigma * rs, g = f > 0 ? -nrm : nrm, h = sigma - f * g;
        for (int k = lane; k < i; k += 64) u[k] = (k == i - 1) ? f - g : row[k];
        if (lane == 0) { sc[0] = h; sc[1] = eig_rcp(h); sc[2] = 1.0; ee[i] = g; }
      }
    }
    __syncthreads();
    if (sc[2] != 0.0) {
      {  // A u (rows < i) and Z u (all rows): four lanes per row (columns q, q + 4, ...), the quad's sum on DPP
        // (measured and dropped, profiles/r6_g_marg_ql.txt: both passes' loads batched six columns at a time from clamped indices — the
        // index arithmetic per load costs more issue slots than the LDS round trips it saves: 335 k -> 481 k clocks for the reduction)
        const int q4 = tid & 3, rows_per_pass = MARG_THREADS / 4;
        for (int rr0 = 0; rr0 < i + n; rr0 += rows_per_pass) {   // (uniform trip count: the quad moves need every lane of the wave)
          const int rr = rr0 + (tid >> 2);
          const bool on = rr < i + n, is_a = rr < i;
          const int r = on ? (is_a ? rr : rr - i) : 0;
          const double *row = (is_a ? A : Z) + r * ld;
          double acc = 0.0;
          for (int k = q4; k < i; k += 4) acc += row[k] * u[k];
          acc = eig_quad_sum(acc);
          if (on && q4 == 0) (is_a ? praw : zv)[r] = acc;
        }
      }
      __syncthreads();
      {  // q = A u / h - K u with K = u^T A u / (2 h^2): every wave sums K itself (no barrier for one scalar)
        double part = 0.0;
        for (int k = lane; k < i; k += 64) part += u[k] * praw[k];
        const double hinv = sc[1];
        const double K = eig_wsum(part) * (0.5 * hinv * hinv);
        // this lane's columns (n <= 128: two), their u and q
        const int c0 = lane, c1 = lane + 64;
        const bool on0 = c0 < i, on1 = c1 < i;
        const double u0 = on0 ? u[c0] : 0.0, u1 = on1 ? u[c1] : 0.0;
        const double q0 = on0 ? praw[c0] * hinv - K * u0 : 0.0, q1 = on1 ? praw[c1] * hinv - K * u1 : 0.0;
        for (int r = wave; r < n; r += NW) {
          const double zr = zv[r] * hinv;
          const bool in_a = r < i;
          const double ur = in_a ? u[r] : 0.0, qr = in_a ? praw[r] * hinv - K * ur : 0.0;
          if (on0) {
            Z[r * ld + c0] -= zr * u0;
            if (in_a) A[r * ld + c0] -= ur * q0 + qr * u0;
          }
          if (on1) {
            Z[r * ld + c1] -= zr * u1;
            if (in_a) A[r * ld + c1] -= ur * q1 + qr * u1;
          }
        }
      }
    }
    __syncthreads();
  }
  const long long t_red = clock64();
  long long t_prod = 0, t_wait = 0;
  // ---------------- implicit QL on (d, e), rotations accumulated into Z.  d and e live in LDS, touched by wave 0 only; every lane of
  // wave 0 runs the same recurrence (reads of one address are a broadcast), lane 0 stores
  int l = 0, m = 0, iter = 0, sweeps = 0;
  bool in_loop = false;
  double fsh = 0.0, tst1 = 0.0;
  const double eps = 2.220446049250313e-16;
  if (wave == 0) {
    for (int k = lane; k < n; k += 64) { dd[k] = A[k * ld + k]; }
    const double e1 = n > 1 ? A[1 * ld + 0] : 0.0;
    eig_wave_sync();
    // e shifted down by one (tql2: e[i - 1] = e[i]), e[n - 1] = 0; entry 1 of the reduction's e is A[1][0]
    double sh0 = 0.0, sh1 = 0.0;
    if (lane + 1 < n) sh0 = lane + 1 == 1 ? e1 : ee[lane + 1];
    if (lane + 65 < n) sh1 = ee[lane + 65];
    eig_wave_sync();
    if (lane < n) ee[lane] = sh0;
    if (lane + 64 < n) ee[lane + 64] = sh1;
    eig_wave_sync();
  }
  // wave 0: the next sweep's rotations into cs buffer `buf`, its range into cmd[buf]
  auto produce = [&](int buf) {
    double *cb = cs + buf * 2 * n;
    int *cm = cmd + buf * 4;
    for (;;) {
      if (!in_loop) {
        if (l >= n) { if (lane == 0) { cm[0] = 0; cm[1] = 0; cm[2] = 1; } return; }
        const double dl = dd[l], el = ee[l];
        tst1 = fmax(tst1, fabs(dl) + fabs(el));
        // first m >= l with |e_m| <= eps tst1 (e_{n-1} = 0: it exists)
        const double thr = eps * tst1;
        const double e_lo = lane < n ? ee[lane] : 0.0, e_hi = lane + 64 < n ? ee[lane + 64] : 0.0;
        const unsigned long long blo = __ballot(lane >= l && lane < n && fabs(e_lo) <= thr);
        const unsigned long long bhi = __ballot(lane + 64 >= l && lane + 64 < n && fabs(e_hi) <= thr);
        m = blo ? __ffsll((long long)blo) - 1 : (bhi ? 64 + __ffsll((long long)bhi) - 1 : n - 1);
        m = __builtin_amdgcn_readfirstlane(m);
        if (m == l) {
          if (lane == 0) { dd[l] = dl + fsh; ee[l] = 0.0; }
          eig_wave_sync();
          ++l; continue;
        }
        in_loop = true; iter = 0;
      }
      ++iter; ++sweeps;
      double g = dd[l];
      const double el = ee[l];
      double p = (dd[l + 1] - g) * eig_rcp(2.0 * el);
      const double w1 = __builtin_fma(p, p, 1.0);
      double r = w1 * eig_rsqrt(w1);
      if (p < 0) r = -r;
      const double d_l = el * eig_rcp(p + r), dl1 = el * (p + r);
      double h = g - d_l;
      const double el1 = ee[l + 1];
      const double pm = dd[m];
      double ei = ee[m - 1], di = dd[m - 1];          // first rotation's operands (m - 1 >= l; index l + 1 > m - 1 only when m == l + 1: then di below is d[l], replaced)
      eig_wave_sync();                                 // (every read of the old values above, before the stores below)
      if (lane == 0) { dd[l] = d_l; dd[l + 1] = dl1; }
      for (int k = lane; k < n; k += 64) if (k >= l + 2) dd[k] -= h;
      eig_wave_sync();
      fsh += h;
      // operands read before the shift: correct them (d[m] and d[m - 1] were shifted by h when their index is >= l + 2; d[l], d[l + 1] were replaced)
      p = (m >= l + 2) ? pm - h : (m == l + 1 ? dl1 : d_l);
      di = (m - 1 >= l + 2) ? di - h : (m - 1 == l + 1 ? dl1 : d_l);
      double c = 1.0, c2 = 1.0, c3 = 1.0, s = 0.0, s2 = 0.0;
      for (int i = m - 1; i >= l; --i) {
        // the next rotation's operands, requested now (they are not written before they are used: rotation i stores e[i + 1], d[i + 1])
        double ein = 0.0, din = 0.0;
        if (i > l) { ein = ee[i - 1]; din = dd[i - 1]; }
        c3 = c2; c2 = c; s2 = s;
        g = c * ei;
        h = c * p;
        const double rr2 = __builtin_fma(p, p, ei * ei);
        const double rinv = eig_rsqrt(rr2);   // (rr2 > 0: |e_i| is above the deflation threshold for l <= i < m)
        r = rr2 * rinv;
        const double e_up = s * r;
        s = ei * rinv;
        c = p * rinv;
        p = __builtin_fma(c, di, -(s * g));
        const double d_up = h + s * __builtin_fma(c, g, s * di);
        if (lane == 0) { ee[i + 1] = e_up; dd[i + 1] = d_up; cb[2 * i] = c; cb[2 * i + 1] = s; }
        ei = ein; di = din;
      }
      p = -s * s2 * c3 * el1 * el * eig_rcp(dl1);   // (e[l] is not written inside the sweep)
      const double e_l = s * p, d_new = c * p;
      const bool conv = !(fabs(e_l) > eps * tst1) || iter >= 200;   // the sweep that converged l (or the cap): close it
      eig_wave_sync();
      if (lane == 0) { ee[l] = conv ? 0.0 : e_l; dd[l] = conv ? d_new + fsh : d_new; cm[0] = l; cm[1] = m; cm[2] = 0; }
      eig_wave_sync();
      if (conv) { ++l; in_loop = false; }
      return;
    }
  };
  int buf = 0;
  if (wave == 0) produce(0);
  __syncthreads();
  for (;;) {
    if (cmd[buf * 4 + 2]) break;
    if (wave == 0) {
      const long long c0 = clock64();
      produce(buf ^ 1);
      t_prod += clock64() - c0;
    } else {
      const int r = tid - 64, cl = cmd[buf * 4], cm_ = cmd[buf * 4 + 1];
      if (r < n) {
        const double *cb = cs + buf * 2 * n;
        double *zr = Z + r * ld;
        double carry = zr[cm_];
        for (int i = cm_ - 1; i >= cl; --i) {
          const double c = cb[2 * i], s = cb[2 * i + 1], zi = zr[i];
          zr[i + 1] = s * zi + c * carry;
          carry = c * zi - s * carry;
        }
        zr[cl] = carry;
      }
    }
    const long long c1 = clock64();
    __syncthreads();
    t_wait += clock64() - c1;
    buf ^= 1;
  }
  if (wave == 0)
    for (int k = lane; k < n; k += 64) A[k * ld + k] = dd[k];
  __syncthreads();
  if (prof && tid == 0) { prof[0] = double(t_red - t_in); prof[1] = double(t_prod); prof[2] = double(t_wait); prof[3] = double(clock64() - t_red); }
  return sweeps;   // (wave 0's count; thread 0 reports it)
}

struct MargLds {
  double *S, *V;      // np2 x ld2 each
  double *a1, *v1;    // 16 x 17 each: the marginalised block and its eigenvectors
  double *ainv;       // 16 x 16
  double *T;          // n x 16
  double *bs, *ev, *w, *cs;
  double *work;       // the eigensolver's vectors: 9 max(n, 16) + 16
  int *ord, *flag;
};

// The dense tail for one system by one workgroup of MARG_THREADS threads.  A: (m + n)^2 row-major, b: m + n (global memory, written
// before a block barrier when the caller assembled them itself); lds: marg_lds_doubles(n) doubles.
__device__ void marg_schur_body(const double *__restrict__ A, const double *__restrict__ b, int m, int n, double eps, double *__restrict__ lin_jac,
                                double *__restrict__ lin_res, double *__restrict__ evals, double *__restrict__ info, double *lds, double *__restrict__ stamps = nullptr) {
  const int tid = threadIdx.x, N = m + n;
  const int np2 = (n + 1) & ~1, ld2 = np2 + 1;
  MargLds L;
  double *ptr = lds;
  L.S = ptr; ptr += np2 * ld2;
  L.V = ptr; ptr += np2 * ld2;
  L.a1 = ptr; ptr += 16 * 17;
  L.v1 = ptr; ptr += 16 * 17;
  L.ainv = ptr; ptr += 16 * 16;
  L.T = ptr; ptr += n * 16;
  L.bs = ptr; ptr += np2;
  L.ev = ptr; ptr += np2;
  L.w = ptr; ptr += np2;
  L.cs = ptr; ptr += np2 + 16;
  L.work = ptr; ptr += 9 * (n > 16 ? n : 16) + 16;
  L.ord = reinterpret_cast<int *>(ptr); ptr += (np2 + 1) / 2 + 1;
  L.flag = reinterpret_cast<int *>(ptr);

  // (stamps: shader-clock stamps of the phases for the batch's test hook — a handful of stores by thread 0)
  auto stamp = [&](int k) { if (tid == 0 && stamps) stamps[k] = double(clock64()); };
  stamp(1);
  // ---- Amm (symmetrised) and its eigendecomposition
  for (int e = tid; e < 16 * 17; e += MARG_THREADS) { L.a1[e] = 0.0; L.v1[e] = 0.0; }
  __syncthreads();
  for (int e = tid; e < 16 * 16; e += MARG_THREADS) {
    const int i = e >> 4, j = e & 15;
    if (i < m && j < m) L.a1[i * 17 + j] = 0.5 * (A[size_t(i) * N + j] + A[size_t(j) * N + i]);
    if (i == j) L.v1[i * 17 + j] = 1.0;
  }
  __syncthreads();
  const int sweeps1 = tridiag_ql_lds(L.a1, L.v1, m, 17, L.work);
  stamp(2);
  for (int e = tid; e < 16 * 16; e += MARG_THREADS) {
    const int i = e >> 4, j = e & 15;
    double s = 0.0;
    if (i < m && j < m)
      for (int k = 0; k < m; ++k) {
        const double ev = L.a1[k * 17 + k];
        s += L.v1[i * 17 + k] * (ev > eps ? 1.0 / ev : 0.0) * L.v1[j * 17 + k];
      }
    L.ainv[e] = s;
  }
  __syncthreads();
  // ---- T = Arm Amm^+ (n x m, padded to 16 columns), bs = br - T bm
  for (int e = tid; e < n * 16; e += MARG_THREADS) {
    const int i = e >> 4, j = e & 15;
    double s = 0.0;
    if (j < m)
      for (int k = 0; k < m; ++k) s += A[size_t(m + i) * N + k] * L.ainv[k * 16 + j];
    L.T[e] = s;
  }
  for (int e = tid; e < np2 * ld2; e += MARG_THREADS) { L.S[e] = 0.0; L.V[e] = 0.0; }
  __syncthreads();
  for (int i = tid; i < n; i += MARG_THREADS) {
    double s = 0.0;
    for (int k = 0; k < m; ++k) s += L.T[i * 16 + k] * b[k];
    L.bs[i] = b[m + i] - s;
    L.V[i * ld2 + i] = 1.0;
  }
  if (tid == 0 && np2 > n) L.V[n * ld2 + n] = 1.0;
  // ---- S = Arr - T Amr on the matrix cores: 16 x 16 output tiles, K = 16 (the zero-padded m), lower triangle of tiles
  {
    const int lane = tid & 63, wave = tid >> 6, nwave = MARG_THREADS / 64;
    const int i = lane & 15, kq = lane >> 4;
    const int nt = (n + 15) / 16;
    int t = 0;
    for (int I = 0; I < nt; ++I)
      for (int J = 0; J <= I; ++J, ++t) {
        if (t % nwave != wave) continue;
        const int rb = 16 * I, cb = 16 * J;
        v4f64 acc;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int row = rb + kq + 4 * rr, col = cb + i;
          acc[rr] = (row < n && col < n) ? A[size_t(m + row) * N + m + col] : 0.0;
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int k = 4 * kk + kq;
          const double aop = (rb + i < n) ? -L.T[(rb + i) * 16 + k] : 0.0;
          const double bop = (k < m && cb + i < n) ? A[size_t(k) * N + m + cb + i] : 0.0;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc, 0, 0, 0);
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int row = rb + kq + 4 * rr, col = cb + i;
          if (row < n && col <= row) { L.S[row * ld2 + col] = acc[rr]; L.S[col * ld2 + row] = acc[rr]; }   // SelfAdjointEigenSolver reads the lower triangle
        }
      }
  }
  __syncthreads();
  stamp(3);
  const int sweeps2 = tridiag_ql_lds(L.S, L.V, n, ld2, L.work, stamps ? stamps + 8 : nullptr);
  stamp(4);
  // ---- ascending eigenvalues (rank sort; ties by index), then the square-root factors
  for (int i = tid; i < n; i += MARG_THREADS) L.ev[i] = L.S[i * ld2 + i];
  __syncthreads();
  for (int i = tid; i < n; i += MARG_THREADS) {
    const double x = L.ev[i];
    int rank = 0;
    for (int j = 0; j < n; ++j) { const double y = L.ev[j]; rank += (y < x || (y == x && j < i)) ? 1 : 0; }
    L.ord[rank] = i;
  }
  __syncthreads();
  for (int k = tid; k < n; k += MARG_THREADS) {
    const int col = L.ord[k];
    const double s = L.ev[col];
    double vb = 0.0;
    for (int i = 0; i < n; ++i) vb += L.V[i * ld2 + col] * L.bs[i];
    const double res = s > eps ? vb / sqrt(s) : 0.0;
    lin_res[k] = res;
    L.T[k] = res;          // (T is dead: the caller's J^T r reads the residuals from here)
    evals[k] = s;
    L.w[k] = s > eps ? sqrt(s) : 0.0;
  }
  __syncthreads();
  // the factor J (row k = sqrt(s_k) v_k^T) goes to global memory and, densely (leading dimension n), over S — dead since its diagonal
  // was read — for the caller's J^T J
  for (int k = tid / 64; k < n; k += MARG_THREADS / 64) {
    const double wk = L.w[k];
    const int col = L.ord[k];
    for (int i = tid & 63; i < n; i += 64) {
      const double v = wk * L.V[i * ld2 + col];
      lin_jac[k * n + i] = v;
      L.S[k * n + i] = v;
    }
  }
  if (tid == 0 && info) { info[0] = double(sweeps1); info[1] = double(sweeps2); }
  stamp(5);
}

__global__ void __launch_bounds__(MARG_THREADS) k_marg_schur(const double *__restrict__ A, const double *__restrict__ b, int m, int n, double eps,
                                                            double *__restrict__ lin_jac, double *__restrict__ lin_res, double *__restrict__ evals,
                                                            double *__restrict__ info) {
  extern __shared__ double lds[];
  marg_schur_body(A, b, m, n, eps, lin_jac, lin_res, evals, info, lds);
}

static size_t marg_lds_doubles(int n) {
  const int np2 = (n + 1) & ~1, ld2 = np2 + 1;
  return size_t(2) * np2 * ld2 + 2 * 16 * 17 + 16 * 16 + size_t(n) * 16 + 3 * np2 + np2 + 16 + 9 * size_t(n > 16 ? n : 16) + 16 + (np2 + 1) / 2 + 1 + 2;
}
static size_t marg_lds_bytes(int n) { return marg_lds_doubles(n) * sizeof(double); }

// ------------------------------------------------------------------------------------------------
// MarginalizationInfo::Marginalize for every window of a batch (MarginalizationFactor.cc:185-311), all of it on the device:
//   launch 1  k_bw_marg_aux     what the factors that touch the dropped blocks contribute at the linearisation point x (the window
//                               after DoubleToVector): block i < Wo the 18 x 13 lidar map of frame i + 1 (extrinsic free), block Wo
//                               the first ImuFactor (ImuFactor.h:53-168), block Wo + 1 the old prior's gradient and cost;
//   launch 2  k_bw_marg_schur   one workgroup per window: frame blocks L S L^T from the solve's final moments (they depend on the
//                               relative poses only, which the yaw re-anchoring leaves unchanged), A and b in the layout
//                               [pose 0, speed-bias 0 | pose 1, speed-bias 1, pose 2 .. Wo, extrinsic] in the summation order of
//                               WindowSystem::evaluate, then the dense tail above, then J^T J and J^T r of the new prior — which
//                               stays on the device as the next solve's prior_mats.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(MARG_AUX_THREADS) k_bw_marg_aux(const BatchSolve *__restrict__ bs, BatchBases bb) {
  const BatchSolve &S = bs[blockIdx.y];
  const DevMarg *mg = S.marg ? rebase(bb.mg, S.marg) : nullptr;
  if (!mg || !mg->active) return;
  __shared__ double aux_lds[1536];
  const DevExec x{int(threadIdx.x), int(blockDim.x), int(threadIdx.x & 63), int(threadIdx.x >> 6), int(blockDim.x >> 6)};
  const int Wo = mg->Wo, i = blockIdx.x;
  const DevParams &P = mg->x;
  if (i < Wo) {
    aux_lmap(x, P.pose[0], P.pose[i + 1], P.ex, rebase(bb.slab, S.marg_lmap) + size_t(i) * DS_LMAP_OUT, aux_lds);
  } else if (i == Wo) {
    double *marg_imu = rebase(bb.slab, S.marg_imu);
    if (mg->has_imu) aux_imu(x, rebase(bb.pb, S.pb)->pim[0], P.pose[0], P.sb[0], P.pose[1], P.sb[1], marg_imu, aux_lds);
    else for (int k = threadIdx.x; k < DS_IMU_OUT; k += blockDim.x) marg_imu[k] = 0.0;
  } else if (i == Wo + 1) {
    if (mg->have_prior) aux_prior(x, *rebase(bb.pb, S.pb), rebase(bb.slab, S.prior_mats), P, rebase(bb.slab, S.marg_prior_out), aux_lds);
  }
}

__global__ void __launch_bounds__(MARG_THREADS) k_bw_marg_schur(const BatchSolve *__restrict__ bs, BatchBases bb, double eps) {
  extern __shared__ double lds[];
  const BatchSolve &S = bs[blockIdx.x];
  const DevMarg *mgp = S.marg ? rebase(bb.mg, S.marg) : nullptr;
  if (!mgp || !mgp->active) return;
  const DevMarg &mg = *mgp;
  const DevExec x{int(threadIdx.x), int(blockDim.x), int(threadIdx.x & 63), int(threadIdx.x >> 6), int(blockDim.x >> 6)};
  const int Wo = mg.Wo, m = mg.m, n = mg.n, N = m + n;
  double *mstamps = rebase(bb.slab, S.marg_info) + MARG_MAX_N + 8;
  if (x.tid == 0) mstamps[0] = double(clock64());
  const DevState &st = *rebase(bb.st, S.st);
  const double *Sm = rebase(bb.slab, S.S_buf) + size_t(st.s_cur) * Wo * LIO_MOMENT_OUT;   // the moments at the point the solver stopped at
  // ---- frame blocks (LDS: zb Wo x 344, LS Wo x 234)
  double *zb = lds, *LS = lds + size_t(Wo) * 344;
  const double *marg_imu = rebase(bb.slab, S.marg_imu), *marg_prior_out = rebase(bb.slab, S.marg_prior_out);
  ds_lidar_blocks(x, Wo, rebase(bb.slab, S.marg_lmap), Sm, LS, zb);
  // ---- A, b: prior, ImuFactor 0, lidar frames 1 .. Wo — one thread per entry, the contributions in that order
  const int np = rebase(bb.pb, S.pb)->n_prior;
  const double *JtJ = rebase(bb.slab, S.prior_mats);
  double *A = rebase(bb.slab, S.marg_A), *bv = A + size_t(N) * N;
  const int c_p0 = mg.pose_col[0], c_ex = mg.ex_col;
  for (int e = x.tid; e < N * (N + 1); e += x.nthr) {
    const int r = e / (N + 1), cc = e % (N + 1);
    const bool is_g = cc == N;
    const int c = is_g ? 0 : cc;
    double v = 0.0;
    if (mg.have_prior && mg.prior_col[r] >= 0) {
      if (is_g) v += marg_prior_out[mg.prior_col[r]];
      else if (mg.prior_col[c] >= 0) v += JtJ[size_t(mg.prior_col[r]) * np + mg.prior_col[c]];
    }
    // ImuFactor 0 spans [pose 0 | sb 0 | pose 1 | sb 1]: local index of a column (-1: not in the factor)
    auto imu_local = [&](int col) {
      if (col >= mg.pose_col[0] && col < mg.pose_col[0] + 6) return col - mg.pose_col[0];
      if (mg.sb_col[0] >= 0 && col >= mg.sb_col[0] && col < mg.sb_col[0] + 9) return 6 + col - mg.sb_col[0];
      if (col >= mg.pose_col[1] && col < mg.pose_col[1] + 6) return 15 + col - mg.pose_col[1];
      if (mg.sb_col[1] >= 0 && col >= mg.sb_col[1] && col < mg.sb_col[1] + 9) return 21 + col - mg.sb_col[1];
      return -1;
    };
    if (mg.has_imu && marg_imu[931] != 0.0) {
      const int lr = imu_local(r);
      if (lr >= 0) {
        if (is_g) v += marg_imu[900 + lr];
        else { const int lc = imu_local(c); if (lc >= 0) v += marg_imu[lr * 30 + lc]; }
      }
    }
    // lidar frame i touches (pose 0, pose i, extrinsic): local rows 0..5, 6..11, 12..17
    auto lidar_kind = [&](int col, int &frame, int &loc) {   // 0 pivot, 1 frame `frame`, 2 extrinsic, -1 none
      if (col >= c_p0 && col < c_p0 + 6) { loc = col - c_p0; return 0; }
      if (col >= c_ex && col < c_ex + 6) { loc = 12 + col - c_ex; return 2; }
      for (int i = 1; i <= Wo; ++i)
        if (col >= mg.pose_col[i] && col < mg.pose_col[i] + 6) { frame = i; loc = 6 + col - mg.pose_col[i]; return 1; }
      return -1;
    };
    int fr_r = 0, lr = 0, fr_c = 0, lc = 0;
    const int kr = lidar_kind(r, fr_r, lr);
    const int kc = is_g ? 0 : lidar_kind(c, fr_c, lc);
    if (kr >= 0 && kc >= 0) {
      for (int i = 1; i <= Wo; ++i) {
        if (Sm[(i - 1) * LIO_MOMENT_OUT + 257] == 0.0) continue;
        if ((kr == 1 && fr_r != i) || (!is_g && kc == 1 && fr_c != i)) continue;
        v += is_g ? zb[(i - 1) * 344 + 324 + lr] : zb[(i - 1) * 344 + lr * 18 + lc];
      }
    }
    if (is_g) bv[r] = v; else A[size_t(r) * N + c] = v;
  }
  __syncthreads();   // (global writes of this block are visible to it behind the barrier)
  // ---- dense tail: the new prior's square-root factors go straight into the next solve's prior_mats
  double *out = rebase(bb.slab, S.next_prior_mats);
  double *o_JtJ = out, *o_jac = out + size_t(n) * n, *o_res = o_jac + size_t(n) * n, *o_Jtr = o_res + n;
  marg_schur_body(A, bv, m, n, eps, o_jac, o_res, rebase(bb.slab, S.marg_info) + 2, rebase(bb.slab, S.marg_info), lds, mstamps);
  __syncthreads();
  // J^T J and J^T r of the new prior (MargPrior::finalize): ascending k; J and the residuals are still in LDS (marg_schur_body left them
  // over S and T — the first form re-read them from global memory, 72 of the kernel's 720 us)
  {
    const int np2 = (n + 1) & ~1, ld2 = np2 + 1;
    const double *Jl = lds, *rl = lds + 2 * np2 * ld2 + 2 * 16 * 17 + 16 * 16;
    for (int i = x.tid / 64; i < n; i += x.nthr / 64)
      for (int j = x.tid & 63; j <= n; j += 64) {
        double sacc = 0.0;
        const double *rhs = j < n ? Jl + j : rl;   // column j of J, or the residuals
        const int rstr = j < n ? n : 1;
        for (int k0 = 0; k0 < n; k0 += 8) {        // eight terms' loads in flight, added in ascending k
          double u8[8], v8[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) { const int k = k0 + q < n ? k0 + q : n - 1; u8[q] = Jl[k * n + i]; v8[q] = rhs[k * rstr]; }
#pragma unroll
          for (int q = 0; q < 8; ++q) if (k0 + q < n) sacc += u8[q] * v8[q];
        }
        if (j < n) o_JtJ[size_t(i) * n + j] = sacc; else o_Jtr[i] = sacc;
      }
  }
  if (x.tid == 0) mstamps[6] = double(clock64());
}

void prepare_bw_marg_kernel() {   // per device (EstimatorBatch's constructor)
  LIO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bw_marg_schur), hipFuncAttributeMaxDynamicSharedMemorySize, int(marg_lds_bytes(MARG_MAX_N))));
}
void launch_bw_marginalize(const BatchSolve *bs, const BatchBases &bb, int B, int max_wo, int max_n, hipStream_t s) {
  if (B <= 0) return;
  hipLaunchKernelGGL(k_bw_marg_aux, dim3(max_wo + 2, B), dim3(MARG_AUX_THREADS), 0, s, bs, bb);
  const size_t lds = std::max(marg_lds_doubles(max_n), size_t(max_wo) * (344 + 234)) * sizeof(double);
  hipLaunchKernelGGL(k_bw_marg_schur, dim3(B), dim3(MARG_THREADS), lds, s, bs, bb, 1e-8);
  LIO_HIP(hipGetLastError());
}

MargSchurDev::MargSchurDev(int device) : device_(device) {
  LIO_HIP(hipSetDevice(device_));
  LIO_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  const size_t N = MARG_MAX_M + MARG_MAX_N;
  LIO_HIP(hipMalloc(reinterpret_cast<void **>(&d_in_), (N * N + N) * sizeof(double)));
  LIO_HIP(hipMalloc(reinterpret_cast<void **>(&d_out_), (size_t(MARG_MAX_N) * MARG_MAX_N + 2 * MARG_MAX_N + 8) * sizeof(double)));
  LIO_HIP(hipHostMalloc(reinterpret_cast<void **>(&h_io_), (N * N + N + size_t(MARG_MAX_N) * MARG_MAX_N + 2 * MARG_MAX_N + 8) * sizeof(double)));
  LIO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_marg_schur), hipFuncAttributeMaxDynamicSharedMemorySize, int(marg_lds_bytes(MARG_MAX_N))));
}
MargSchurDev::~MargSchurDev() {
  if (h_io_) (void)hipHostFree(h_io_);
  if (d_in_) (void)hipFree(d_in_);
  if (d_out_) (void)hipFree(d_out_);
  if (stream_) (void)hipStreamDestroy(stream_);
}

bool MargSchurDev::Run(const double *A, const double *b, int m, int n, double eps, double *lin_jac, double *lin_res, double *evals, int *sweeps) {
  if (m < 1 || m > MARG_MAX_M - 1 || n < 1 || n > MARG_MAX_N) return false;
  const auto t0 = std::chrono::steady_clock::now();
  LIO_HIP(hipSetDevice(device_));   // the caller is the estimator's worker thread: the current device is per thread
  const size_t N = size_t(m) + n, n_in = N * N + N, n_out = size_t(n) * n + 2 * size_t(n) + 2;
  std::memcpy(h_io_, A, N * N * sizeof(double));
  std::memcpy(h_io_ + N * N, b, N * sizeof(double));
  LIO_HIP(hipMemcpyAsync(d_in_, h_io_, n_in * sizeof(double), hipMemcpyHostToDevice, stream_));
  double *d_jac = d_out_, *d_res = d_out_ + size_t(n) * n, *d_ev = d_res + n, *d_info = d_ev + n;
  hipLaunchKernelGGL(k_marg_schur, dim3(1), dim3(MARG_THREADS), marg_lds_bytes(n), stream_, d_in_, d_in_ + N * N, m, n, eps, d_jac, d_res, d_ev, d_info);
  LIO_HIP(hipGetLastError());
  double *h_out = h_io_ + n_in;
  LIO_HIP(hipMemcpyAsync(h_out, d_out_, n_out * sizeof(double), hipMemcpyDeviceToHost, stream_));
  LIO_HIP(hipStreamSynchronize(stream_));
  std::memcpy(lin_jac, h_out, size_t(n) * n * sizeof(double));
  std::memcpy(lin_res, h_out + size_t(n) * n, n * sizeof(double));
  if (evals) std::memcpy(evals, h_out + size_t(n) * n + n, n * sizeof(double));
  if (sweeps) { sweeps[0] = int(h_out[size_t(n) * n + 2 * n]); sweeps[1] = int(h_out[size_t(n) * n + 2 * n + 1]); }
  last_ms_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  static const bool dbg = std::getenv("LIO_DEBUG_TIMING") != nullptr;
  if (dbg)
    std::fprintf(stderr, "[lio_hip marg timing] device Schur + eigen m %d n %d: %.3f ms (Jacobi sweeps %d + %d)\n", m, n, last_ms_,
                 int(h_out[size_t(n) * n + 2 * n]), int(h_out[size_t(n) * n + 2 * n + 1]));
  return true;
}

}  // namespace lio
