// pingpong.hip — latency of a host <-> resident-kernel round trip on the MI355X box, for the design of the resident moments
// kernel.  Host writes a sequence number to a doorbell, a spinning kernel echoes it into coherent host memory,
// the host measures ring -> echo seen.  Variants: where the doorbell lives (coherent host memory / fine-grained device memory
// written by the CPU through the BAR), how many blocks poll, sleep between polls.  Every kernel gives up after 2 s.
// build: hipcc --offload-arch=gfx950 -O3 -o pingpong pingpong.hip ; run: timeout 120 ./pingpong
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int SCOPE>
__global__ void k_echo(const volatile unsigned long long *door, unsigned *words, unsigned long long *stamps, int n_pass, int sleep, long long timeout) {
  const int b = blockIdx.x;
  if (threadIdx.x != 0) return;
  const long long t0 = wall_clock64();
  for (int p = 1; p <= n_pass; ++p) {
    for (;;) {
      const unsigned long long v = __hip_atomic_load((const unsigned long long *)door, __ATOMIC_RELAXED, SCOPE);
      if (v >= (unsigned long long)p) break;
      if (wall_clock64() - t0 > timeout) return;
      if (sleep == 1) __builtin_amdgcn_s_sleep(1); else if (sleep >= 8) __builtin_amdgcn_s_sleep(8);
    }
    if (stamps && b == 0) stamps[p] = wall_clock64();
    __hip_atomic_store(words + b, (unsigned)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }

static bool g_sfence = false;
static int run(const char *name, volatile unsigned long long *door_host_view, const unsigned long long *door_dev_view, bool agent_scope, int nblocks, int sleep,
               unsigned *words, hipStream_t s, int khz) {
  const int n_pass = 200;
  *door_host_view = 0;
  std::memset(words, 0, sizeof(unsigned) * 512);
  if (agent_scope) hipLaunchKernelGGL(k_echo<__HIP_MEMORY_SCOPE_AGENT>, dim3(nblocks), dim3(64), 0, s, door_dev_view, words, nullptr, n_pass, sleep, (long long)khz * 2000);
  else hipLaunchKernelGGL(k_echo<__HIP_MEMORY_SCOPE_SYSTEM>, dim3(nblocks), dim3(64), 0, s, door_dev_view, words, nullptr, n_pass, sleep, (long long)khz * 2000);
  std::vector<double> rtt_first, rtt_all;
  double t_end = now_us() + 2.5e6;
  // let the kernel start
  double w = now_us(); while (now_us() - w < 300) {}
  for (int p = 1; p <= n_pass; ++p) {
    const double t0 = now_us();
    __atomic_store_n(door_host_view, (unsigned long long)p, __ATOMIC_RELEASE);
    if (g_sfence) __builtin_ia32_sfence();   // push a write-combined store out of the CPU's WC buffer
    double t_first = 0;
    int k = 0;
    for (;;) {
      while (k < nblocks && __atomic_load_n(words + k, __ATOMIC_ACQUIRE) == (unsigned)p) { if (!t_first) t_first = now_us(); ++k; }
      if (k == nblocks) break;
      if (now_us() > t_end) { std::printf("%-44s TIMEOUT at pass %d (%d / %d words)\n", name, p, k, nblocks); hipStreamSynchronize(s); return 0; }
    }
    const double t1 = now_us();
    rtt_first.push_back(t_first - t0); rtt_all.push_back(t1 - t0);
    double q = now_us(); while (now_us() - q < 10) {}   // 10 us of "host work" between passes
  }
  CK(hipStreamSynchronize(s));
  std::sort(rtt_first.begin(), rtt_first.end()); std::sort(rtt_all.begin(), rtt_all.end());
  std::printf("%-44s blocks %3d sleep %2d : ring -> first echo median %.2f us, ring -> all echoes median %.2f us (p90 %.2f)\n", name, nblocks, sleep,
              rtt_first[n_pass / 2], rtt_all[n_pass / 2], rtt_all[n_pass * 9 / 10]);
  return 0;
}

int main() {
  int dev = 0, khz = 100000;
  CK(hipSetDevice(0));
  hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev);
  std::printf("wall clock %d kHz\n", khz);
  hipStream_t s;
  CK(hipStreamCreate(&s));
  unsigned *words;
  CK(hipHostMalloc((void **)&words, sizeof(unsigned) * 512, hipHostMallocCoherent));
  // (1) doorbell in coherent host memory
  unsigned long long *door_h;
  CK(hipHostMalloc((void **)&door_h, 4096, hipHostMallocCoherent));
  for (int nb : {1, 8, 50, 100}) for (int sl : {0, 1, 8}) run("doorbell in coherent host memory", door_h, door_h, false, nb, sl, words, s, khz);
  // (2) doorbell in non-coherent mapped host memory (GPU may cache it: expected to fail / be slow) -- skipped
  // (3) doorbell in fine-grained DEVICE memory, written by the CPU through the BAR
  unsigned long long *door_d = nullptr;
  hipError_t e = hipExtMallocWithFlags((void **)&door_d, 4096, hipDeviceMallocFinegrained);
  std::printf("hipExtMallocWithFlags(finegrained) -> %s\n", hipGetErrorString(e));
  if (e == hipSuccess) {
    CK(hipMemset(door_d, 0, 4096));
    CK(hipDeviceSynchronize());
    signal(SIGSEGV, on_segv); signal(SIGBUS, on_segv);
    if (sigsetjmp(jb, 1) == 0) {
      volatile unsigned long long probe = *(volatile unsigned long long *)door_d;   // is it CPU-readable at all?
      std::printf("CPU read of fine-grained device memory ok (%llu)\n", (unsigned long long)probe);
      g_sfence = true;
      for (int nb : {1, 8, 50, 100}) for (int sl : {0, 1}) run("doorbell in fine-grained device memory (BAR) + sfence", door_d, door_d, false, nb, sl, words, s, khz);
      for (int nb : {1, 100}) run("  same, agent-scope polls", door_d, door_d, true, nb, 1, words, s, khz);
      g_sfence = false;
    } else {
      std::printf("CPU access to fine-grained device memory faults: not usable as a doorbell\n");
    }
    signal(SIGSEGV, SIG_DFL); signal(SIGBUS, SIG_DFL);
  }
  // (4) managed memory
  unsigned long long *door_m = nullptr;
  e = hipMallocManaged((void **)&door_m, 4096);
  std::printf("hipMallocManaged -> %s\n", hipGetErrorString(e));
  if (e == hipSuccess) {
    hipMemAdvise(door_m, 4096, hipMemAdviseSetCoarseGrain, 0);   // ignore errors
    hipMemAdvise(door_m, 4096, hipMemAdviseUnsetCoarseGrain, 0);
    *door_m = 0;
    for (int nb : {1, 100}) run("doorbell in managed memory", door_m, door_m, false, nb, 1, words, s, khz);
  }
  // launch + sync baseline: what one empty launch + completion word costs
  {
    std::vector<double> t;
    for (int p = 1; p <= 200; ++p) {
      *door_h = 0; words[0] = 0;
      __atomic_store_n(door_h, 1ull, __ATOMIC_RELEASE);
      const double t0 = now_us();
      hipLaunchKernelGGL(k_echo<__HIP_MEMORY_SCOPE_SYSTEM>, dim3(1), dim3(64), 0, s, door_h, words, nullptr, 1, 0, (long long)khz * 100);
      while (__atomic_load_n(words, __ATOMIC_ACQUIRE) != 1u) {}
      t.push_back(now_us() - t0);
      hipStreamSynchronize(s);
    }
    std::sort(t.begin(), t.end());
    std::printf("one launch of an echo kernel: call -> word seen median %.2f us (p90 %.2f)\n", t[100], t[180]);
  }
  return 0;
}
