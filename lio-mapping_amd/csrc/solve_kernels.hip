// solve_kernels.hip — gfx950 kernels: S_i = sum_k rho'_k z_k z_k^T on v_mfma_f64_16x16x4_f64.
//
// One wave consumes 4 residuals per MFMA: lane l supplies element (l & 15) of residual (l >> 4) as BOTH
// the A operand (A[i = l&15][k = l>>4]) and the B operand (B[k = l>>4][j = l&15]) — the fragment layouts
// of the f64 16x16x4 form coincide for a symmetric rank-4 update, so no LDS transpose is needed.
// The accumulator is 4 f64 per lane: D[row = (l>>4) + 4 r][col = l & 15]  (cdna_hip_programming.md §3:
// the f64 MFMA does NOT use the f32 C/D map).
#include <hip/hip_runtime.h>

#include <cfloat>

#include "solve_kernels.h"

namespace lio {

typedef double v4f64 __attribute__((ext_vector_type(4)));

#define MOMENT_THREADS 256

int moment_blocks_per_frame(int max_slots) {
  // 16 residuals per block-iteration; aim for >= 8 iterations per block, cap so partial reduction stays small
  int b = cdiv(max_slots, 16 * 8);
  return b < 1 ? 1 : (b > 64 ? 64 : b);
}

__global__ void __launch_bounds__(MOMENT_THREADS) k_lidar_moments(MomentArgs a, const uint8_t *__restrict__ valid,
                                                                  const float4 *__restrict__ coef, double *__restrict__ partials) {
  const MomentFrame &fr = a.fr[blockIdx.y];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int e = lane & 15, grp = lane >> 4;
  const int ea = e >> 2, eb = e & 3;
  const int waves_total = gridDim.x * (MOMENT_THREADS / 64);
  const int wid = blockIdx.x * (MOMENT_THREADS / 64) + wv;
  v4f64 acc = {0.0, 0.0, 0.0, 0.0};
  double cost = 0.0, cnt = 0.0;
  for (int base = wid * 4; base < fr.nslots; base += waves_total * 4) {
    int sidx = base + grp;
    double operand = 0.0;
    if (sidx < fr.nslots && valid[fr.slot_off + sidx]) {
      float4 po = fr.stack[sidx % fr.M];
      float4 c = coef[fr.slot_off + sidx];
      double px = po.x, py = po.y, pz = po.z;
      double w0 = c.x, w1 = c.y, w2 = c.z, d = c.w;
      double qx = fr.R[0] * px + fr.R[1] * py + fr.R[2] * pz + fr.t[0];
      double qy = fr.R[3] * px + fr.R[4] * py + fr.R[5] * pz + fr.t[1];
      double qz = fr.R[6] * px + fr.R[7] * py + fr.R[8] * pz + fr.t[2];
      double r = w0 * qx + w1 * qy + w2 * qz + d;
      double sq = r * r;
      double inv = 1.0 / (1.0 + sq);           // CauchyLoss(1): rho' = 1/(1+s); rho'' < 0 => alpha = 0
      double rho1 = inv > DBL_MIN ? inv : DBL_MIN;
      double sw = sqrt(rho1);
      double wa = ea == 0 ? w0 : (ea == 1 ? w1 : w2);
      double pb = eb == 0 ? px : (eb == 1 ? py : (eb == 2 ? pz : 1.0));
      double z = e < 12 ? wa * pb : (e == 12 ? d : 0.0);
      operand = sw * z;
      if (e == 0) { cost += 0.5 * log(1.0 + sq); cnt += 1.0; }
    }
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(operand, operand, acc, 0, 0, 0);
  }
  __shared__ double sm[MOMENT_THREADS / 64][LIO_MOMENT_OUT];
#pragma unroll
  for (int r = 0; r < 4; ++r) sm[wv][(grp + 4 * r) * 16 + e] = acc[r];
  for (int off = 32; off > 0; off >>= 1) { cost += __shfl_down(cost, off, 64); cnt += __shfl_down(cnt, off, 64); }
  if (lane == 0) { sm[wv][256] = cost; sm[wv][257] = cnt; }
  __syncthreads();
  double *dst = partials + (size_t(blockIdx.y) * gridDim.x + blockIdx.x) * LIO_MOMENT_OUT;
  for (int k = threadIdx.x; k < 258; k += MOMENT_THREADS) {
    double v = 0;
    for (int w = 0; w < MOMENT_THREADS / 64; ++w) v += sm[w][k];
    dst[k] = v;
  }
}

__global__ void k_moment_reduce(const double *__restrict__ partials, int bpf, double *__restrict__ out) {
  const double *src = partials + size_t(blockIdx.x) * bpf * LIO_MOMENT_OUT;
  for (int k = threadIdx.x; k < 258; k += blockDim.x) {
    double v = 0;
    for (int b = 0; b < bpf; ++b) v += src[size_t(b) * LIO_MOMENT_OUT + k];  // fixed order: deterministic
    out[size_t(blockIdx.x) * LIO_MOMENT_OUT + k] = v;
  }
}

void launch_lidar_moments(const MomentArgs &a, const uint8_t *valid, const float4 *coef, double *partials, double *out, hipStream_t s) {
  if (a.nframes <= 0) return;
  hipLaunchKernelGGL(k_lidar_moments, dim3(a.blocks_per_frame, a.nframes), dim3(MOMENT_THREADS), 0, s, a, valid, coef, partials);
  hipLaunchKernelGGL(k_moment_reduce, dim3(a.nframes), dim3(256), 0, s, partials, a.blocks_per_frame, out);
  LIO_HIP(hipGetLastError());
}

}  // namespace lio
