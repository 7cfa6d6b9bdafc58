// grid_sync.hip — what the pieces of a one-launch, many-phase kernel (k_vox_fused, csrc/cloud_kernels.hip) cost on the MI355X:
//   A  a grid barrier of 256 blocks x 1024 threads (16 arrival lines, agent-scope atomics, 16 polling lanes), back to back
//   B  agent-scope loads by one thread per block of lines that (i) nobody touched, (ii) one block hit with an atomic, (iii) every block hit
//      with an atomic right before the barrier
//   C  one returning agent-scope atomic per THREAD on a 32 MB table: (i) every lane its own 64 B line, (ii) the 64 lanes of a wave in
//      the same 64 B line (a voxel-sorted cloud), (iii) as (ii) after the lanes were dealt out with a stride
// Times are wall_clock64 ticks (100 MHz) of block 0 / the slowest block.  build: hipcc --offload-arch=gfx950 -O3 -o grid_sync grid_sync.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define LINES 16
template <typename T> __device__ __forceinline__ void ast(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ T ald(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gsync(unsigned *ctr, unsigned target) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    if (lane == 0) __hip_atomic_fetch_add(ctr + (blockIdx.x % LINES) * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
      const bool in = lane >= LINES || int(ald(ctr + lane * 16) - target) >= 0;
      if (__all(in)) break;
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
}
struct Args { unsigned *bar; unsigned target; unsigned *lines; unsigned *table; long long *out; int mode; };
// out[block * 16 + k]: stamps
__global__ void __launch_bounds__(1024) k(Args a) {
  const int tid = threadIdx.x, b = blockIdx.x, G = gridDim.x;
  long long *o = a.out + size_t(b) * 16;
  unsigned bi = 0;
  auto B = [&]() { gsync(a.bar + (bi++) * LINES * 16, a.target); };
  B();
  const long long t0 = wall_clock64();
  if (a.mode == 0) {   // A: eight barriers back to back
    for (int r = 0; r < 8; ++r) B();
    if (tid == 0) o[0] = wall_clock64() - t0;
  } else if (a.mode == 1) {   // B
    unsigned s = 0;
    if (tid == 0) { for (int k2 = 0; k2 < 7; ++k2) s += ald(a.lines + (0 + k2) * 16); }
    __syncthreads();
    if (tid == 0) o[0] = wall_clock64() - t0;
    if (b == 0 && tid < 7) __hip_atomic_fetch_add(a.lines + (16 + tid) * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    B();
    const long long t1 = wall_clock64();
    if (tid == 0) { for (int k2 = 0; k2 < 7; ++k2) s += ald(a.lines + (16 + k2) * 16); }
    __syncthreads();
    if (tid == 0) o[1] = wall_clock64() - t1;
    if (tid < 7) __hip_atomic_fetch_max(a.lines + (32 + tid) * 16, unsigned(b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long t1b = wall_clock64();
    B();
    const long long t2 = wall_clock64();
    if (tid == 0) { for (int k2 = 0; k2 < 7; ++k2) s += ald(a.lines + (32 + k2) * 16); }
    __syncthreads();
    if (tid == 0) { o[2] = wall_clock64() - t2; o[3] = t2 - t1b; o[15] = s; }
  } else {   // C
    const unsigned gid = unsigned(b) * 1024u + unsigned(tid), nthreads = unsigned(G) * 1024u;
    unsigned word;
    if (a.mode == 2) word = (gid * 16u) % (8u << 20);                                    // own line per lane
    else if (a.mode == 3) word = (gid / 4u) % (8u << 20);                                // consecutive bytes: a wave in one line
    else { const unsigned S = 4096u, K = nthreads / S; const unsigned i = (gid % S) * K + gid / S; word = (i / 4u) % (8u << 20); }   // dealt out
    const unsigned old = __hip_atomic_fetch_add(a.table + word, 1u << (8u * (gid & 3u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == 0xFFFFFFFFu) o[14] = 1;
    __syncthreads();
    if (tid == 0) o[0] = wall_clock64() - t0;
    B();
    if (tid == 0) o[1] = wall_clock64() - t0;
    a.table[word] = 0u;
  }
}
int main() {
  int cus = 0;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  const int G = std::min(256, cus) / LINES * LINES;
  unsigned *bar, *lines, *table; long long *out;
  CK(hipMalloc(&bar, 16 * LINES * 16 * 4)); CK(hipMalloc(&lines, 64 * 16 * 4)); CK(hipMalloc(&table, size_t(8u << 20) * 4)); CK(hipMalloc(&out, size_t(G) * 16 * 8));
  CK(hipMemset(bar, 0, 16 * LINES * 16 * 4)); CK(hipMemset(lines, 0, 64 * 16 * 4)); CK(hipMemset(table, 0, size_t(8u << 20) * 4));
  unsigned epoch = 0;
  std::vector<long long> h(size_t(G) * 16);
  const char *names[5] = {"A  8 barriers back to back", "B  7 loads by one thread per block: untouched | after 1 block's atomics | after every block's atomics (+ that barrier)",
                          "C  one returning atomic per thread, own line", "C  ... a wave in one line", "C  ... a wave in one line, lanes dealt out"};
  for (int mode = 0; mode < 5; ++mode) {
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipMemset(out, 0, size_t(G) * 16 * 8));
      ++epoch;
      Args a{bar, epoch * unsigned(G / LINES), lines, table, out, mode};
      hipLaunchKernelGGL(k, dim3(G), dim3(1024), 0, 0, a);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h.data(), out, size_t(G) * 16 * 8, hipMemcpyDeviceToHost));
      if (rep < 2) continue;
      std::printf("%s:", names[mode]);
      for (int k2 = 0; k2 < 4; ++k2) {
        long long mx = 0; for (int b = 0; b < G; ++b) mx = std::max(mx, h[size_t(b) * 16 + k2]);
        if (mx || k2 == 0) std::printf("  [%d] block0 %.2f us, slowest %.2f us", k2, h[k2] / 100.0, mx / 100.0);
      }
      std::printf("\n");
    }
  }
  return 0;
}
