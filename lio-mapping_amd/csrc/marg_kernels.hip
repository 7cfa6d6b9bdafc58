// marg_kernels.hip — see marg_kernels.h.  One workgroup; every matrix lives in LDS.
//
//   Amm   = 1/2 (A[:m,:m] + A[:m,:m]^T)                     (MarginalizationFactor.cc:275)
//   Amm^+ = V diag(1 / s_i if s_i > eps else 0) V^T          (:276-283, SelfAdjointEigenSolver)
//   T     = Arm Amm^+,  S = Arr - T Amr,  bs = br - T bm     (:285-291; the n x m x n product runs on v_mfma_f64_16x16x4)
//   S     = V2 diag(s) V2^T                                  (:293)
//   linearized_jacobians = diag(sqrt(s_k > eps ? s_k : 0)) V2^T,  linearized_residuals = diag(1/sqrt(s_k) or 0) V2^T bs   (:294-302)
//
// The eigensolver is the cyclic two-sided Jacobi method in the round-robin ordering: n / 2 disjoint rotations per step, every
// step three block barriers (rotation angles; rows; columns of S and of the eigenvector matrix).  It is the textbook choice
// for one workgroup — no serial QL sweep — and it resolves the small eigenvalues of the badly graded S (entries from 1e9 down
// to 1e-4) at least as well as the tridiagonal QL iteration the host uses; which eigenvalues fall on which side of the absolute
// 1e-8 cut is rounding noise on either path (tests/golden/README.md).
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

// cyclic Jacobi on the symmetric np x np matrix S (leading dimension ld, np even; a padding row / column must be zero),
// eigenvectors accumulated in the columns of V (identity on entry).  cs: np doubles of scratch (16-byte aligned), flag: two ints.
//
// A step applies np / 2 disjoint rotations J (round-robin pairing): S <- J^T S J, V <- V J.  The pairs partition the indices, so
// S falls into (np / 2)^2 blocks of 2 x 2 — rows of pair P, columns of pair Q — and a block's new value depends on its old value and
// the two angles only: one thread rotates a block's rows and then its columns in registers and stores it and its transpose (S stays
// exactly symmetric; only one block of every unordered pair {P, Q} is computed).  Two barriers per step: angles | blocks and
// eigenvector rows.  Where the time went (n = 45, 12 sweeps x 45 steps, shader clocks per step, profiles/r6_b_marg_phases.txt):
// round 5's form — a row pass and a column pass over the whole matrix, four barriers — 9.3 k; the fused block form with correctly
// rounded divides / square roots in the angles and a division per item 6.1 k; hardware reciprocal / rsqrt + Newton, division-free
// item map, 512 threads 3.3 k, of which 2.5 k were vector-instruction ISSUE (selects, address products, the identity row rotation of
// the eigenvector items) — not LDS latency and not arithmetic (a step is 8.6 k fp64 operations per workgroup).  This form: triangle of
// blocks, eigenvector rows on their own, fused multiply-adds.
__device__ __forceinline__ double jac_rcp(double d) {   // 1 / d: hardware estimate + two Newton steps (<= 1 ulp)
  double y = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-d, y, 1.0);
  return __builtin_fma(y, e, y);
}
__device__ __forceinline__ double jac_rsqrt(double d) {  // 1 / sqrt(d) likewise
  double y = __builtin_amdgcn_rsq(d);
  const double h = 0.5 * d;
  y = y * __builtin_fma(-h * y, y, 1.5);
  return y * __builtin_fma(-h * y, y, 1.5);
}
__device__ int jacobi_eig_lds(double *S, double *V, int np, int ld, double *cs, int *flag, double *jprof = nullptr) {
  const int tid = threadIdx.x, half = np / 2, nm1 = np - 1;
#if defined(LIO_MARG_PROF)
  long long tA = 0, tB1 = 0, tB = 0, tB2 = 0;
#endif
  // round-robin pairing of step r: index np - 1 stays, the others rotate; pair t = (r + t, r - t) mod (np - 1), pair 0 = (np - 1, r)
  // (both sums lie in [0, 2 (np - 1)): no division; every thread derives the pairs it needs — a pair table in LDS cost a round trip)
  auto pair_of = [&](int r, int t, int &p, int &q) {
    p = r + t; q = r - t + nm1;
    p = p >= nm1 ? p - nm1 : p; q = q >= nm1 ? q - nm1 : q;
    if (t == 0) { p = nm1; q = r; }
    if (p > q) { const int x = p; p = q; q = x; }
  };
  // A thread keeps ONE pair index cq = tid mod half and a row offset rp0 = tid / half:
  //   blocks of S: pair P = cq against pair Q = cq + j (mod half), j = rp0, rp0 + rstep, ... <= half / 2 — every unordered {P, Q} once
  //                (for even half the distance half / 2 would come twice: only P < half / 2 takes it);
  //   eigenvectors: column pair cq of rows k = rp0, rp0 + rstep, ...
  const int cq = tid % half, rp0 = tid / half, rstep = MARG_THREADS / half;
  const bool worker = rp0 < rstep;
  const int jmax = half / 2;
  const bool even_half = (half & 1) == 0;
  const double2 *cs2 = reinterpret_cast<const double2 *>(cs);   // (c, s) of pair t
  int sweeps = 0;
  for (; sweeps < 40; ++sweeps) {
    if (tid == 0) flag[0] = 0;
    __syncthreads();
    for (int r = 0; r < nm1; ++r) {
#if defined(LIO_MARG_PROF)
      const long long c0 = clock64();
#endif
      if (tid < half) {   // (one wave: its stores to the step's flag are ordered)
        if (tid == 0) flag[1] = 0;
        int p, q;
        pair_of(r, tid, p, q);
        const double app = S[p * ld + p], aqq = S[q * ld + q], apq = S[p * ld + q];
        double c = 1.0, s = 0.0;
        // |apq| > 2.3e-16 sqrt(|app aqq|), compared in squares (an apq whose square underflows counts as zero)
        if (fabs(apq) > 1e-300 && apq * apq > 5.29e-32 * fabs(app * aqq)) {
          const double tau = (aqq - app) * jac_rcp(2.0 * apq);
          const double w = __builtin_fma(tau, tau, 1.0);
          const double t = (tau >= 0 ? 1.0 : -1.0) * jac_rcp(fabs(tau) + w * jac_rsqrt(w));
          c = jac_rsqrt(__builtin_fma(t, t, 1.0));
          s = t * c;
          flag[0] = 1; flag[1] = 1;
        }
        cs[2 * tid] = c; cs[2 * tid + 1] = s;
      }
#if defined(LIO_MARG_PROF)
      const long long c1_ = clock64();
#endif
      __syncthreads();
#if defined(LIO_MARG_PROF)
      const long long c2_ = clock64();
#endif
      if (worker && flag[1]) {   // (a step without a rotation — the last sweep is made of them — leaves everything as it is)
        const double2 aq = cs2[cq];
        int pc, qc;
        pair_of(r, cq, pc, qc);
        // ---- blocks of S: rows of pair P = cq, columns of pair Q
        for (int j = rp0; j <= jmax; j += rstep) {
          if (even_half && j == jmax && cq >= jmax) break;
          int tq = cq + j;
          tq = tq >= half ? tq - half : tq;
          const double2 bq = cs2[tq];
          int p2, q2;
          pair_of(r, tq, p2, q2);
          const double c1 = aq.x, s1 = aq.y, c2 = bq.x, s2 = bq.y;
          const int ra = pc * ld, rb = qc * ld;
          const double a = S[ra + p2], b = S[ra + q2], c = S[rb + p2], d = S[rb + q2];
          // rows pc, qc <- J_P^T (rows), then columns p2, q2 <- (columns) J_Q
          const double a1 = __builtin_fma(c1, a, -(s1 * c)), c1r = __builtin_fma(s1, a, c1 * c), b1 = __builtin_fma(c1, b, -(s1 * d)), d1 = __builtin_fma(s1, b, c1 * d);
          double ao = __builtin_fma(c2, a1, -(s2 * b1)), bo = __builtin_fma(s2, a1, c2 * b1), co = __builtin_fma(c2, c1r, -(s2 * d1)), dd = __builtin_fma(s2, c1r, c2 * d1);
          if (j == 0 && s1 != 0.0) { bo = 0.0; co = 0.0; }   // the rotated pair is exactly decoupled
          S[ra + p2] = ao; S[ra + q2] = bo; S[rb + p2] = co; S[rb + q2] = dd;
          if (j != 0) { S[p2 * ld + pc] = ao; S[q2 * ld + pc] = bo; S[p2 * ld + qc] = co; S[q2 * ld + qc] = dd; }   // the transposed block
        }
        // ---- eigenvectors: V <- V J, columns of pair cq
        if (aq.y != 0.0) {
          const double c2 = aq.x, s2 = aq.y;
          for (int k0 = rp0; k0 < np; k0 += 4 * rstep) {
            double u[4], w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { const int k = k0 + e * rstep, kc = k < np ? k : rp0; u[e] = V[kc * ld + pc]; w[e] = V[kc * ld + qc]; }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int k = k0 + e * rstep;
              if (k < np) { V[k * ld + pc] = __builtin_fma(c2, u[e], -(s2 * w[e])); V[k * ld + qc] = __builtin_fma(s2, u[e], c2 * w[e]); }
            }
          }
        }
      }
#if defined(LIO_MARG_PROF)
      const long long c3_ = clock64();
#endif
      __syncthreads();
#if defined(LIO_MARG_PROF)
      const long long c4_ = clock64();
      tA += c1_ - c0; tB1 += c2_ - c1_; tB += c3_ - c2_; tB2 += c4_ - c3_;
#endif
    }
    const int any = flag[0];
    __syncthreads();
    if (!any) break;
  }
#if defined(LIO_MARG_PROF)
  if (jprof && tid == 0) { jprof[0] = double(tA); jprof[1] = double(tB1); jprof[2] = double(tB); jprof[3] = double(tB2); }
#endif
  return sweeps;
}

struct MargLds {
  double *S, *V;      // np2 x ld2 each
  double *a1, *v1;    // 16 x 17 each: the marginalised block and its eigenvectors
  double *ainv;       // 16 x 16
  double *T;          // n x 16
  double *bs, *ev, *w, *cs;
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
  L.cs = ptr; ptr += np2 + 16;        // (c, s) of the step's np2 / 2 rotations, 16-byte aligned
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
  const int sweeps1 = jacobi_eig_lds(L.a1, L.v1, 16, 17, L.cs, L.flag);
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
  const int sweeps2 = jacobi_eig_lds(L.S, L.V, np2, ld2, L.cs, L.flag, stamps ? stamps + 8 : nullptr);
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
  return size_t(2) * np2 * ld2 + 2 * 16 * 17 + 16 * 16 + size_t(n) * 16 + 3 * np2 + np2 + 16 + (np2 + 1) / 2 + 1 + 2;
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
