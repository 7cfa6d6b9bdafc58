// host_store_ack.hip — what a worker block of the resident moments kernel pays to publish a frame record to the host
// (first bullet: "the acknowledgement of the system-scope stores is the prime suspect").  One wave writes
// K x 512 B into coherent pinned host memory (or fine-grained device memory, for comparison), waits for its stores
// (s_waitcnt vmcnt(0)), stores a flag; the kernel stamps wall_clock64() (100 MHz) at the three points, the host stamps when it sees
// the flag.  Variants: store flavour (system-scope `sc0 sc1` as the product's host_store, agent-scope `sc1`, plain), store width
// (4 / 8 / 16 B per lane), K in {1, 2, 4, 8} (512 B .. 4 KB: the compact frame record is 93 doubles = 744 B), number of waves
// publishing at once (1, 5 = one per frame of the headline window, 64).
// Written at the end of round 3 (no GPU left): compile-checked only.
// build: hipcc --offload-arch=gfx950 -O3 -o host_store_ack host_store_ack.hip ; run: timeout 120 ./host_store_ack
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

enum { PLAIN = 0, AGENT = 1, SYSTEM = 2 };

template <int FLAVOUR, typename T>
__device__ __forceinline__ void put(T *dst, T v) {
  if (FLAVOUR == SYSTEM) __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  else if (FLAVOUR == AGENT) __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *dst = v;
}

// each block: one wave; pass p waits for the doorbell (so that passes are separated and the host can time them), then publishes
template <int FLAVOUR, int WIDTH>
__global__ void __launch_bounds__(64) k_publish(const volatile unsigned *door, unsigned char *payload, unsigned *flags, long long *stamps, int K, int n_pass,
                                                long long timeout) {
  const int b = blockIdx.x, lane = threadIdx.x;
  unsigned char *mine = payload + size_t(b) * 8192;
  const long long t_begin = wall_clock64();
  for (int p = 1; p <= n_pass; ++p) {
    for (;;) {
      const unsigned v = __hip_atomic_load((const unsigned *)door, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (v >= unsigned(p)) break;
      if (wall_clock64() - t_begin > timeout) return;
      __builtin_amdgcn_s_sleep(1);
    }
    const long long t0 = wall_clock64();
    for (int k = 0; k < K; ++k) {
      unsigned char *row = mine + size_t(k) * 64 * WIDTH;
      if (WIDTH == 4) put<FLAVOUR>(reinterpret_cast<unsigned *>(row) + lane, unsigned(p + k));
      else if (WIDTH == 8) put<FLAVOUR>(reinterpret_cast<unsigned long long *>(row) + lane, (unsigned long long)(p + k));
      else {   // 16 B per lane: two 8-byte halves issued back to back (the builtin has no 16-byte atomic store)
        put<FLAVOUR>(reinterpret_cast<unsigned long long *>(row) + 2 * lane, (unsigned long long)(p + k));
        put<FLAVOUR>(reinterpret_cast<unsigned long long *>(row) + 2 * lane + 1, (unsigned long long)(p + k));
      }
    }
    const long long t1 = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t2 = wall_clock64();
    if (lane == 0) {
      __hip_atomic_store(flags + b, unsigned(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (b == 0) { stamps[3 * p] = t1 - t0; stamps[3 * p + 1] = t2 - t1; stamps[3 * p + 2] = t2; }
    }
  }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <int FLAVOUR, int WIDTH>
static int run(const char *name, int K, int nblocks, volatile unsigned *door, unsigned char *payload_dev_view, unsigned *flags, long long *stamps, hipStream_t s) {
  const int n_pass = 200;
  *door = 0;
  std::memset(flags, 0, sizeof(unsigned) * 256);
  hipLaunchKernelGGL((k_publish<FLAVOUR, WIDTH>), dim3(nblocks), dim3(64), 0, s, door, payload_dev_view, flags, stamps, K, n_pass, 200000000LL);
  double w = now_us(); while (now_us() - w < 300) {}
  std::vector<double> host;
  const double t_end = now_us() + 3e6;
  for (int p = 1; p <= n_pass; ++p) {
    const double t0 = now_us();
    __atomic_store_n(door, unsigned(p), __ATOMIC_RELEASE);
    int k = 0;
    for (;;) {
      while (k < nblocks && __atomic_load_n(flags + k, __ATOMIC_ACQUIRE) == unsigned(p)) ++k;
      if (k == nblocks) break;
      if (now_us() > t_end) { std::printf("%-40s TIMEOUT at pass %d\n", name, p); (void)hipStreamSynchronize(s); return 0; }
    }
    host.push_back(now_us() - t0);
    double w2 = now_us(); while (now_us() - w2 < 20) {}
  }
  if (hipStreamSynchronize(s) != hipSuccess) { std::printf("%-40s kernel failed\n", name); return 0; }
  std::vector<double> issue, ack;
  for (int p = 20; p <= n_pass; ++p) { issue.push_back(stamps[3 * p] * 0.01); ack.push_back(stamps[3 * p + 1] * 0.01); }   // 100 MHz -> us
  std::sort(issue.begin(), issue.end()); std::sort(ack.begin(), ack.end()); std::sort(host.begin(), host.end());
  std::printf("%-40s K %d (%5d B / wave) waves %3d | issue %.2f us  wait for the stores %.2f us (median; p90 %.2f) | ring -> all flags seen by the host %.2f us (median)\n",
              name, K, K * 64 * WIDTH, nblocks, issue[issue.size() / 2], ack[ack.size() / 2], ack[ack.size() * 9 / 10], host[host.size() / 2]);
  return 0;
}

int main() {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unsigned *door, *flags;
  long long *stamps;
  unsigned char *pay_host, *pay_fine;
  CK(hipHostMalloc(&door, 64, hipHostMallocCoherent | hipHostMallocMapped));
  CK(hipHostMalloc(&flags, sizeof(unsigned) * 256, hipHostMallocCoherent | hipHostMallocMapped));
  CK(hipHostMalloc(&stamps, sizeof(long long) * 3 * 256, hipHostMallocCoherent | hipHostMallocMapped));
  CK(hipHostMalloc(&pay_host, 8192 * 64, hipHostMallocCoherent | hipHostMallocMapped));
  CK(hipExtMallocWithFlags(reinterpret_cast<void **>(&pay_fine), 8192 * 64, hipDeviceMallocFinegrained));
  for (int waves : {1, 5, 64}) {
    for (int K : {1, 2, 4, 8}) {
      run<SYSTEM, 8>("host memory, sc0 sc1, 8 B per lane", K, waves, door, pay_host, flags, stamps, s);
      run<SYSTEM, 16>("host memory, sc0 sc1, 2 x 8 B per lane", K, waves, door, pay_host, flags, stamps, s);
      run<SYSTEM, 4>("host memory, sc0 sc1, 4 B per lane", K, waves, door, pay_host, flags, stamps, s);
      run<AGENT, 8>("host memory, sc1 (agent), 8 B per lane", K, waves, door, pay_host, flags, stamps, s);
      run<PLAIN, 8>("host memory, plain, 8 B per lane", K, waves, door, pay_host, flags, stamps, s);
      run<SYSTEM, 8>("fine-grained device memory, sc0 sc1, 8 B", K, waves, door, pay_fine, flags, stamps, s);
    }
    std::printf("\n");
  }
  return 0;
}
