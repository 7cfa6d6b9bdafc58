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

#define MARG_THREADS 256

// cyclic Jacobi on the symmetric np x np matrix S (leading dimension ld, np even; a padding row / column must be zero),
// eigenvectors accumulated in the columns of V (identity on entry).  cs: np doubles of scratch, flag: one int.
__device__ int jacobi_eig_lds(double *S, double *V, int np, int ld, double *cs, int *flag) {
  const int tid = threadIdx.x, half = np / 2;
  int sweeps = 0;
  for (; sweeps < 40; ++sweeps) {
    if (tid == 0) *flag = 0;
    __syncthreads();
    for (int r = 0; r < np - 1; ++r) {
      // round-robin pairing: index np - 1 stays, the others rotate
      auto pair_of = [&](int t, int &p, int &q) {
        if (t == 0) { p = np - 1; q = r; }
        else { p = (r + t) % (np - 1); q = (r - t + (np - 1)) % (np - 1); }
        if (p > q) { const int x = p; p = q; q = x; }
      };
      if (tid < half) {
        int p, q;
        pair_of(tid, p, q);
        const double app = S[p * ld + p], aqq = S[q * ld + q], apq = S[p * ld + q];
        double c = 1.0, s = 0.0;
        if (fabs(apq) > 2.3e-16 * sqrt(fabs(app) * fabs(aqq)) && fabs(apq) > 1e-300) {
          const double tau = (aqq - app) / (2.0 * apq);
          const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
          c = 1.0 / sqrt(1.0 + t * t);
          s = t * c;
          *flag = 1;
        }
        cs[2 * tid] = c; cs[2 * tid + 1] = s;
      }
      __syncthreads();
      // rows: S <- J^T S
      for (int item = tid; item < half * np; item += MARG_THREADS) {
        const int t = item / np, k = item - t * np;
        const double c = cs[2 * t], s = cs[2 * t + 1];
        if (s == 0.0) continue;
        int p, q;
        pair_of(t, p, q);
        const double x = S[p * ld + k], y = S[q * ld + k];
        S[p * ld + k] = c * x - s * y;
        S[q * ld + k] = s * x + c * y;
      }
      __syncthreads();
      // columns: S <- S J, V <- V J
      for (int item = tid; item < half * np; item += MARG_THREADS) {
        const int t = item / np, k = item - t * np;
        const double c = cs[2 * t], s = cs[2 * t + 1];
        if (s == 0.0) continue;
        int p, q;
        pair_of(t, p, q);
        const double x = S[k * ld + p], y = S[k * ld + q];
        S[k * ld + p] = c * x - s * y;
        S[k * ld + q] = s * x + c * y;
        const double u = V[k * ld + p], w = V[k * ld + q];
        V[k * ld + p] = c * u - s * w;
        V[k * ld + q] = s * u + c * w;
      }
      __syncthreads();
      if (tid < half && cs[2 * tid + 1] != 0.0) {   // the rotated pair is exactly decoupled
        int p, q;
        pair_of(tid, p, q);
        S[p * ld + q] = 0.0; S[q * ld + p] = 0.0;
      }
      // (the next step's angle phase reads only after its own barrier below; entries written here belong to this thread's pair)
      __syncthreads();
    }
    const int any = *flag;
    __syncthreads();
    if (!any) break;
  }
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
                                double *__restrict__ lin_res, double *__restrict__ evals, double *__restrict__ info, double *lds) {
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
  L.ord = reinterpret_cast<int *>(ptr); ptr += (np2 + 1) / 2 + 1;
  L.flag = reinterpret_cast<int *>(ptr);

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
  const int sweeps2 = jacobi_eig_lds(L.S, L.V, np2, ld2, L.cs, L.flag);
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
    lin_res[k] = s > eps ? vb / sqrt(s) : 0.0;
    evals[k] = s;
    L.w[k] = s > eps ? sqrt(s) : 0.0;
  }
  __syncthreads();
  for (int e = tid; e < n * n; e += MARG_THREADS) {
    const int k = e / n, i = e - k * n;
    lin_jac[e] = L.w[k] * L.V[i * ld2 + L.ord[k]];
  }
  if (tid == 0 && info) { info[0] = double(sweeps1); info[1] = double(sweeps2); }
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
__global__ void __launch_bounds__(MARG_THREADS) k_bw_marg_aux(const BatchSolve *__restrict__ bs, BatchBases bb) {
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
  marg_schur_body(A, bv, m, n, eps, o_jac, o_res, rebase(bb.slab, S.marg_info) + 2, rebase(bb.slab, S.marg_info), lds);
  __syncthreads();
  // J^T J and J^T r of the new prior (MargPrior::finalize): ascending k
  for (int e = x.tid; e < n * (n + 1); e += x.nthr) {
    const int i = e / (n + 1), j = e % (n + 1);
    double sacc = 0.0;
    if (j < n) { for (int k = 0; k < n; ++k) sacc += o_jac[size_t(k) * n + i] * o_jac[size_t(k) * n + j]; o_JtJ[size_t(i) * n + j] = sacc; }
    else { for (int k = 0; k < n; ++k) sacc += o_jac[size_t(k) * n + i] * o_res[k]; o_Jtr[i] = sacc; }
  }
}

void prepare_bw_marg_kernel() {   // per device (EstimatorBatch's constructor)
  LIO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bw_marg_schur), hipFuncAttributeMaxDynamicSharedMemorySize, int(marg_lds_bytes(MARG_MAX_N))));
}
void launch_bw_marginalize(const BatchSolve *bs, const BatchBases &bb, int B, int max_wo, int max_n, hipStream_t s) {
  if (B <= 0) return;
  hipLaunchKernelGGL(k_bw_marg_aux, dim3(max_wo + 2, B), dim3(MARG_THREADS), 0, s, bs, bb);
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
