// diag_block.hip — variants of the 16 x 16 diagonal-block step of the LDS-resident L D L^T (one wave): where do the 10-12 k clocks of
// DevExec::panel_factor_regs go, and what does a fused-multiply-add form buy?  Every variant is checked against a host factorisation.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I lio-mapping_amd/csrc tools/micro/diag_block.hip -o /tmp/diag_block
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "solve_step.h"
#include "solve_device.h"
using namespace lio;

#define NB 16
__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double rcp2(double d) {
  double y = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-d, y, 1.0);
  y = __builtin_fma(y, e, y);
  return y;
}

// MODE bits: 1 = pivot loop, 2 = write-out, 4 = inverse; FMA: fused multiply-adds in the updates
template <int MODE, bool FMA>
__device__ __forceinline__ int block_variant(double *A, int ld, int p, double *invd, double *scr, int lane) {
  const int r = lane & 15;
  double a[NB];
  const double *row = A + size_t(p + r) * ld + p;
#pragma unroll
  for (int c = 0; c < NB; ++c) a[c] = row[c];
  int ok = 1;
  double myinv = 0.0;
  if (MODE & 1) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      double *bc = scr + (j & 1) * NB;
      if (lane < NB) bc[lane] = a[j];
      wsync();
      const double d = bc[j];
      ok &= (d > 0.0) ? 1 : 0;
      const double inv = rcp2(d);
      const double l = a[j] * inv;
#pragma unroll
      for (int c = j + 1; c < NB; ++c) a[c] = FMA ? __builtin_fma(-l, bc[c], a[c]) : a[c] - l * bc[c];
      a[j] = (r > j) ? l : a[j];
      myinv = (r == j) ? inv : myinv;
    }
  }
  if (MODE & 2) {
    if (lane < NB) {
      double *orow = A + size_t(p + r) * ld + p;
#pragma unroll
      for (int c = 0; c < NB; ++c)
        if (c <= r) orow[c] = a[c];
      invd[p + r] = myinv;
    }
    wsync();
  }
  if (MODE & 4) {
    const int c = lane & 15;
    double xcol[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) xcol[k] = (k == c) ? 1.0 : 0.0;
#pragma unroll
    for (int rr = 1; rr < NB; ++rr) {
      const double *lr = A + size_t(p + rr) * ld + p;
      double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
      for (int k = 0; k < rr; ++k) {
        if (FMA) { if (k & 1) acc1 = __builtin_fma(lr[k], xcol[k], acc1); else acc0 = __builtin_fma(lr[k], xcol[k], acc0); }
        else { if (k & 1) acc1 += lr[k] * xcol[k]; else acc0 += lr[k] * xcol[k]; }
      }
      xcol[rr] = (rr > c) ? -(acc0 + acc1) : xcol[rr];
    }
    double *Xs = scr + 4 * NB;
    if (lane < NB) {
#pragma unroll
      for (int rr = 0; rr < NB; ++rr) Xs[rr * NB + c] = xcol[rr];
    }
  }
  return ok;
}

// Variant G: the inverse from the registers of the factorisation — lane r keeps row r of L; X is built ROW-wise, row r of X by lane r:
// X[r][:] = e_r - sum_{k<r} L[r][k] X[k][:], the rows k < r arriving as LDS broadcasts one step at a time (16 steps, one exchange each)
template <bool FMA>
__device__ __forceinline__ int block_rowwise(double *A, int ld, int p, double *invd, double *scr, int lane) {
  const int r = lane & 15;
  double a[NB], xr[NB];
  const double *row = A + size_t(p + r) * ld + p;
#pragma unroll
  for (int c = 0; c < NB; ++c) { a[c] = row[c]; xr[c] = (c == r) ? 1.0 : 0.0; }
  int ok = 1;
  double myinv = 0.0;
  // one exchange per pivot carries column j of the block AND row j of X (final once pivots < j are applied): 32 doubles
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    double *bc = scr + (j & 1) * 2 * NB;
    if (lane < NB) bc[lane] = a[j];
    if (lane == j) {
#pragma unroll
      for (int c = 0; c < NB; ++c) bc[NB + c] = xr[c];
    }
    wsync();
    const double d = bc[j];
    ok &= (d > 0.0) ? 1 : 0;
    const double inv = rcp2(d);
    const double l = a[j] * inv;
#pragma unroll
    for (int c = j + 1; c < NB; ++c) a[c] = FMA ? __builtin_fma(-l, bc[c], a[c]) : a[c] - l * bc[c];
    if (r > j) {
#pragma unroll
      for (int c = 0; c <= j; ++c) xr[c] = FMA ? __builtin_fma(-l, bc[NB + c], xr[c]) : xr[c] - l * bc[NB + c];
    }
    a[j] = (r > j) ? l : a[j];
    myinv = (r == j) ? inv : myinv;
  }
  if (lane < NB) {
    double *orow = A + size_t(p + r) * ld + p;
    double *Xs = scr + 4 * NB;
#pragma unroll
    for (int c = 0; c < NB; ++c) {
      if (c <= r) orow[c] = a[c];
      Xs[r * NB + c] = xr[c];
    }
    invd[p + r] = myinv;
  }
  return ok;
}

#include <type_traits>

// Variant P: fma pivot loop, staging block + 64-lane triangle copy, inverse by 16 lanes with L11's rows PREFETCHED into registers in
// two batches (rows 1..10, rows 11..15): the just-in-time broadcast reads of the plain form wait for LDS sixty times
template <int NACC>
__device__ __forceinline__ int block_prefetch(double *A, int ld, int p, double *invd, double *scr, int lane) {
  const int r = lane & 15;
  double a[NB];
  const double *row = A + size_t(p + r) * ld + p;
#pragma unroll
  for (int c = 0; c < NB; ++c) a[c] = row[c];
  int ok = 1;
  double myinv = 0.0;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    double *bc = scr + (j & 1) * NB;
    if (lane < NB) bc[lane] = a[j];
    wsync();
    const double d = bc[j];
    ok &= (d > 0.0) ? 1 : 0;
    const double inv = rcp2(d);
    const double l = a[j] * inv;
#pragma unroll
    for (int c = j + 1; c < NB; ++c) a[c] = __builtin_fma(-l, bc[c], a[c]);
    a[j] = (r > j) ? l : a[j];
    myinv = (r == j) ? inv : myinv;
  }
  double *Ls = scr + 4 * NB + NB * NB;
  if (lane < NB) {
#pragma unroll
    for (int c = 0; c < NB; ++c) Ls[r * NB + c] = a[c];
    invd[p + r] = myinv;
  }
  wsync();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = lane + 64 * q, rr = e >> 4, cc = e & 15;
    if (cc <= rr) A[size_t(p + rr) * ld + p + cc] = Ls[e];
  }
  const int c = lane & 15;
  double xcol[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) xcol[k] = (k == c) ? 1.0 : 0.0;
  auto rows = [&](auto lo_tag, auto hi_tag) {
    constexpr int LO = decltype(lo_tag)::value, HI = decltype(hi_tag)::value;
    double Lr[(HI * (HI + 1) - LO * (LO - 1)) / 2 + 1];
    int n = 0;
#pragma unroll
    for (int rr = LO; rr <= HI; ++rr)
#pragma unroll
      for (int k = 0; k < rr; ++k) Lr[n++] = Ls[rr * NB + k];
    n = 0;
#pragma unroll
    for (int rr = LO; rr <= HI; ++rr) {
      double acc[NACC];
#pragma unroll
      for (int q = 0; q < NACC; ++q) acc[q] = 0.0;
#pragma unroll
      for (int k = 0; k < rr; ++k) { acc[k % NACC] = __builtin_fma(Lr[n], xcol[k], acc[k % NACC]); ++n; }
      double s = acc[0];
      if (NACC == 2) s = acc[0] + acc[1];
      if (NACC == 4) s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
      xcol[rr] = (rr > c) ? -s : xcol[rr];
    }
  };
  rows(std::integral_constant<int, 1>{}, std::integral_constant<int, 10>{});
  rows(std::integral_constant<int, 11>{}, std::integral_constant<int, 15>{});
  double *Xs = scr + 4 * NB;
  if (lane < NB) {
#pragma unroll
    for (int rr = 0; rr < NB; ++rr) Xs[rr * NB + c] = xcol[rr];
  }
  return ok;
}

template <int V>
__global__ void __launch_bounds__(256) k_block(const double *Ain, double *Aout, double *Xout, double *dout, long long *ticks, int reps) {
  extern __shared__ double dyn[];
  const int lane = threadIdx.x & 63;
  if (threadIdx.x >= 64) return;
  const int ld = 97;
  double *A = dyn, *invd = dyn + 96 * ld, *scr = invd + 128;
  const DevExec x{int(threadIdx.x), 256, lane, 0, 4};
  long long total = 0;
  int ok = 1;
  for (int k = 0; k < reps; ++k) {
    for (int e = lane; e < NB * NB; e += 64) A[(e / NB) * ld + e % NB] = Ain[e];
    wsync();
    const long long t0 = clock64();
    if (V == 0) ok &= x.panel_factor_regs(A, ld, 0, invd, scr);
    if (V == 1) ok &= block_variant<7, false>(A, ld, 0, invd, scr, lane);
    if (V == 2) ok &= block_variant<7, true>(A, ld, 0, invd, scr, lane);
    if (V == 3) ok &= block_variant<3, true>(A, ld, 0, invd, scr, lane);
    if (V == 4) ok &= block_variant<1, true>(A, ld, 0, invd, scr, lane);
    if (V == 5) ok &= block_rowwise<true>(A, ld, 0, invd, scr, lane);
    if (V == 6) ok &= block_rowwise<false>(A, ld, 0, invd, scr, lane);
    if (V == 7) ok &= block_prefetch<2>(A, ld, 0, invd, scr, lane);
    if (V == 8) ok &= block_prefetch<4>(A, ld, 0, invd, scr, lane);
    wsync();
    total += clock64() - t0;
  }
  for (int e = lane; e < NB * NB; e += 64) { Aout[e] = A[(e / NB) * ld + e % NB]; Xout[e] = scr[4 * NB + e]; }
  if (lane < NB) dout[lane] = invd[lane];
  if (lane == 0) { ticks[0] = total; ticks[1] = ok; }
}

int main() {
  std::vector<double> A(NB * NB), L(NB * NB, 0.0), D(NB), X(NB * NB, 0.0);
  for (int i = 0; i < NB; ++i)
    for (int j = 0; j < NB; ++j) A[i * NB + j] = (i == j) ? 20.0 + i : 1.0 / (1.0 + i + j);
  // host: L D L^T and X = L^-1
  {
    std::vector<double> W = A;
    for (int j = 0; j < NB; ++j) {
      D[j] = W[j * NB + j];
      for (int r = j + 1; r < NB; ++r) L[r * NB + j] = W[r * NB + j] / D[j];
      for (int r = j + 1; r < NB; ++r)
        for (int c = j + 1; c <= r; ++c) W[r * NB + c] -= L[r * NB + j] * W[c * NB + j];
    }
    for (int c = 0; c < NB; ++c) {
      X[c * NB + c] = 1.0;
      for (int r = c + 1; r < NB; ++r) { double s = 0; for (int k = c; k < r; ++k) s += L[r * NB + k] * X[k * NB + c]; X[r * NB + c] = -s; }
    }
  }
  double *dA, *dAo, *dX, *dd; long long *dt;
  hipMalloc(&dA, sizeof(double) * NB * NB); hipMalloc(&dAo, sizeof(double) * NB * NB); hipMalloc(&dX, sizeof(double) * NB * NB); hipMalloc(&dd, sizeof(double) * NB);
  hipMalloc(&dt, 64);
  hipMemcpy(dA, A.data(), sizeof(double) * NB * NB, hipMemcpyHostToDevice);
  const int reps = 512;
  const char *names[9] = {"current panel_factor_regs", "same, local copy (pivot + write-out + inverse)", "fused multiply-adds", "fma, no inverse (timing only)",
                          "fma, pivot loop only (timing only)", "row-wise inverse inside the pivot loop, fma", "row-wise inverse inside the pivot loop, mul + add",
                          "fma, staging block, inverse from prefetched rows, 2 sums", "fma, staging block, inverse from prefetched rows, 4 sums"};
#define RUN(V)                                                                                                                        \
  {                                                                                                                                   \
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_block<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);          \
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(k_block<V>, dim3(1), dim3(256), 96 * 1024, 0, dA, dAo, dX, dd, dt, reps);        \
    hipDeviceSynchronize();                                                                                                           \
    long long t[2]; std::vector<double> Ao(NB * NB), Xo(NB * NB), di(NB);                                                             \
    hipMemcpy(t, dt, sizeof(t), hipMemcpyDeviceToHost); hipMemcpy(Ao.data(), dAo, sizeof(double) * NB * NB, hipMemcpyDeviceToHost);   \
    hipMemcpy(Xo.data(), dX, sizeof(double) * NB * NB, hipMemcpyDeviceToHost); hipMemcpy(di.data(), dd, sizeof(double) * NB, hipMemcpyDeviceToHost); \
    double eL = 0, eX = 0, eD = 0;                                                                                                    \
    for (int i = 0; i < NB; ++i) {                                                                                                    \
      eD = std::fmax(eD, std::fabs(di[i] * D[i] - 1.0));                                                                              \
      for (int j = 0; j < i; ++j) eL = std::fmax(eL, std::fabs(Ao[i * NB + j] - L[i * NB + j]));                                     \
      for (int j = 0; j <= i; ++j) eX = std::fmax(eX, std::fabs(Xo[i * NB + j] - X[i * NB + j]));                                    \
    }                                                                                                                                 \
    std::printf("V%d %-52s %8.0f ticks per block (ok %lld)  |L err| %.1e  |1/d err| %.1e  |X err| %.1e\n", V, names[V], double(t[0]) / reps, t[1], eL, eD, eX); \
  }
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(7) RUN(8)
  return 0;
}
