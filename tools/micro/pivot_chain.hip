// pivot_chain.hip — where the latency of one pivot step of the LDS-resident L D L^T goes (one wave, dependent chains, gfx950).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I lio-mapping_amd/csrc tools/micro/pivot_chain.hip -o /tmp/pivot_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "solve_step.h"
#include "solve_device.h"
using namespace lio;

__global__ void __launch_bounds__(512) k_chain(double *out, long long *ticks, int reps) {
  __shared__ double lds[64 * 4];
  extern __shared__ double dyn[];
  const int lane = threadIdx.x & 63;
  if (threadIdx.x >= 64) return;
  double v = 1.0 + 1e-3 * lane;
  // (a) reciprocal chain: v_rcp_f64 + two Newton steps, dependent
  long long t0 = clock64();
  for (int k = 0; k < reps; ++k) {
    double y = __builtin_amdgcn_rcp(v);
    double e = __builtin_fma(-v, y, 1.0); y = __builtin_fma(y, e, y);
    e = __builtin_fma(-v, y, 1.0); y = __builtin_fma(y, e, y);
    v = y + 0.5;
  }
  long long t1 = clock64();
  // (b) dependent fp64 FMA chain
  double w = v;
  for (int k = 0; k < reps; ++k) { w = __builtin_fma(w, 0.999999, 1e-7); w = __builtin_fma(w, 1.000001, -1e-7); w = __builtin_fma(w, 0.999999, 1e-7); w = __builtin_fma(w, 1.000001, -1e-7); }
  long long t2 = clock64();
  // (c) LDS round trip: write own value, wave barrier, read a neighbour's
  double u = w;
  for (int k = 0; k < reps; ++k) {
    lds[lane] = u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    u = lds[(lane + 1) & 63] * 0.5 + 0.25;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
  }
  long long t3 = clock64();
  // (d) v_readlane broadcast chain
  double r = u;
  for (int k = 0; k < reps; ++k) { r = ds_bcast_lane(r, 5) * 0.5 + 0.125 * lane; }
  long long t4 = clock64();
  // (e) the 16 x 16 diagonal block as the step kernel factors it
  const int ld = 97;
  double *A = dyn, *invd = dyn + 96 * ld, *scr = invd + 128;
  for (int e2 = lane; e2 < 16 * 16; e2 += 64) { const int i = e2 / 16, j = e2 % 16; A[i * ld + j] = (i == j) ? 20.0 + i : 1.0 / (1.0 + i + j); }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const DevExec x{int(threadIdx.x), 512, lane, 0, 8};
  long long t5 = clock64();
  int ok = 1;
  for (int k = 0; k < reps / 16 + 1; ++k) {
    ok &= x.panel_factor_regs(A, ld, 0, invd, scr);
    if (lane < 16) A[lane * ld + lane] += 20.0;   // keep it positive definite for the next repetition
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  long long t6 = clock64();
  long long w0 = wall_clock64();
  for (int k = 0; k < 64; ++k) { w = __builtin_fma(w, 0.999999, 1e-7); }
  long long c0 = clock64();
  for (volatile int k = 0; k < 20000; ++k) {}
  long long w1 = wall_clock64(), c1 = clock64();
  if (lane == 0) {
    ticks[0] = t1 - t0; ticks[1] = t2 - t1; ticks[2] = t3 - t2; ticks[3] = t4 - t3; ticks[4] = t6 - t5; ticks[5] = w1 - w0; ticks[6] = c1 - c0; ticks[7] = ok;
    out[0] = v + w + u + r;
  }
}

int main() {
  double *d_out; long long *d_t;
  hipMalloc(&d_out, 64); hipMalloc(&d_t, 64 * 8);
  hipFuncSetAttribute(reinterpret_cast<const void *>(k_chain), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  const int reps = 4096;
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(k_chain, dim3(1), dim3(512), 96 * 1024, 0, d_out, d_t, reps);
  hipDeviceSynchronize();
  long long t[8];
  hipMemcpy(t, d_t, sizeof(t), hipMemcpyDeviceToHost);
  const double ghz = double(t[6]) / (double(t[5]) * 10.0);   // wall clock = 100 MHz
  std::printf("clock64 runs at %.3f GHz (against the 100 MHz wall clock)\n", ghz);
  std::printf("per step, clock64 ticks: rcp + 2 Newton %.1f | 4 dependent fp64 FMA %.1f | LDS write-barrier-read %.1f | v_readlane bcast + fma %.1f\n", double(t[0]) / reps,
              double(t[1]) / reps, double(t[2]) / reps, double(t[3]) / reps);
  std::printf("16 x 16 diagonal block (factor + inverse): %.0f ticks = %.2f us (ok %lld)\n", double(t[4]) / (reps / 16 + 1), double(t[4]) / (reps / 16 + 1) / ghz * 1e-3, t[7]);
  return 0;
}
