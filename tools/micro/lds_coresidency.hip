// lds_coresidency.hip — does a workgroup with a large DYNAMIC LDS allocation keep its LDS to itself when blocks of another kernel
// (other stream) share its CU?  Round 6: the batched trust-region step (113 KB dynamic LDS, one workgroup per window) failed at random
// only when the other loop group's moments / aux blocks (34 KB / 12 KB static LDS) could be co-resident; with 160 KB requested
// (nothing co-resident) it never failed.
//   k_big   <<<n, 256, dyn>>>  fills its dynamic LDS with a block-specific pattern, spins ~spin clocks, verifies, `rounds` times
//   k_small <<<m, 256>>>       the same with 34 KB of static LDS
// build: hipcc --offload-arch=gfx950 -O3 -o lds_coresidency lds_coresidency.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void spin(long long clocks) { const long long t0 = clock64(); while (clock64() - t0 < clocks) __builtin_amdgcn_s_sleep(4); }

__global__ void __launch_bounds__(256) k_big(int ndoubles, int rounds, long long clocks, unsigned tag, unsigned *err) {
  extern __shared__ __attribute__((aligned(16))) double big[];
  for (int r = 0; r < rounds; ++r) {
    const double base = double(tag + blockIdx.x * 131u + r);
    for (int i = threadIdx.x; i < ndoubles; i += 256) big[i] = base + i;
    __syncthreads();
    spin(clocks);
    int bad = 0;
    for (int i = threadIdx.x; i < ndoubles; i += 256) if (big[i] != base + i) ++bad;
    if (bad) atomicAdd(err, unsigned(bad));
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256) k_small(int rounds, long long clocks, unsigned tag, unsigned *err) {
  __shared__ double sm[4352];   // 34816 B, the moments kernel's transpose buffers
  for (int r = 0; r < rounds; ++r) {
    const double base = -double(tag + blockIdx.x * 17u + r);
    for (int i = threadIdx.x; i < 4352; i += 256) sm[i] = base - i;
    __syncthreads();
    spin(clocks);
    int bad = 0;
    for (int i = threadIdx.x; i < 4352; i += 256) if (sm[i] != base - i) ++bad;
    if (bad) atomicAdd(err, unsigned(bad));
    __syncthreads();
  }
}

int main(int argc, char **argv) {
  const int reps = argc > 1 ? std::atoi(argv[1]) : 50;
  hipStream_t sa, sb;
  CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_big), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  unsigned *err; CK(hipMalloc(&err, 8)); 
  const int sizes[] = {113088, 65536, 60000, 90112, 131072, 160 * 1024 - 34816, 160 * 1024};
  for (int bytes : sizes) {
    for (int mode = 0; mode < 2; ++mode) {   // 0: big alone, 1: big beside the small kernel on another stream
      CK(hipMemset(err, 0, 8));
      for (int rep = 0; rep < reps; ++rep) {
        if (mode == 1) hipLaunchKernelGGL(k_small, dim3(800), dim3(256), 0, sb, 6, 20000LL, unsigned(rep * 7), err + 1);
        hipLaunchKernelGGL(k_big, dim3(32), dim3(256), size_t(bytes), sa, bytes / 8, 4, 20000LL, unsigned(rep * 3), err);
        if (mode == 1) hipLaunchKernelGGL(k_small, dim3(800), dim3(256), 0, sb, 6, 20000LL, unsigned(rep * 7 + 1), err + 1);
        hipLaunchKernelGGL(k_big, dim3(32), dim3(256), size_t(bytes), sa, bytes / 8, 4, 20000LL, unsigned(rep * 3 + 1), err);
      }
      CK(hipDeviceSynchronize());
      unsigned h[2]; CK(hipMemcpy(h, err, 8, hipMemcpyDeviceToHost));
      std::printf("dynamic LDS %6d B, %s: corrupted doubles seen by the big workgroups %u, by the 34 KB blocks %u\n", bytes, mode ? "beside 34 KB blocks of another stream" : "alone", h[0], h[1]);
    }
  }
  return 0;
}
