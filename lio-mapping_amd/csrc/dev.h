// dev.h — HIP runtime plumbing for the product: error propagation, growable device buffers, timers.
// No CPU fallback anywhere: a failed HIP call surfaces as LIO_ERR_DEVICE through the C-ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

namespace lio {

struct DeviceError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

#define LIO_HIP(call)                                                                                         \
  do {                                                                                                        \
    hipError_t e__ = (call);                                                                                  \
    if (e__ != hipSuccess)                                                                                    \
      throw ::lio::DeviceError(std::string(#call) + " -> " + hipGetErrorString(e__) + " @" + __FILE__ + ":" + \
                               std::to_string(__LINE__));                                                     \
  } while (0)

// Growable device array.  Capacity only grows (amortised); contents are NOT preserved on growth
// unless keep=true.
template <typename T>
struct DBuf {
  T *p = nullptr;
  size_t cap = 0;
  DBuf() = default;
  DBuf(const DBuf &) = delete;
  DBuf &operator=(const DBuf &) = delete;
  DBuf(DBuf &&o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
  DBuf &operator=(DBuf &&o) noexcept {
    if (this != &o) { release(); p = o.p; cap = o.cap; o.p = nullptr; o.cap = 0; }
    return *this;
  }
  ~DBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
  }
  void reserve(size_t n, hipStream_t s = nullptr, bool keep = false, size_t keep_n = 0) {
    if (n <= cap) return;
    size_t ncap = cap ? cap : 256;
    while (ncap < n) ncap = ncap + ncap / 2 + 256;
    T *np = nullptr;
    LIO_HIP(hipMalloc(reinterpret_cast<void **>(&np), ncap * sizeof(T)));
    if (keep && p && keep_n) LIO_HIP(hipMemcpyAsync(np, p, keep_n * sizeof(T), hipMemcpyDeviceToDevice, s));
    if (p) { LIO_HIP(hipStreamSynchronize(s)); (void)hipFree(p); }
    p = np; cap = ncap;
  }
};

struct Stopwatch {
  hipEvent_t a{}, b{};
  Stopwatch() { (void)hipEventCreate(&a); (void)hipEventCreate(&b); }
  ~Stopwatch() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
};

// A completion word in coherent (fine-grained) pinned host memory: the last block of the last kernel of a pass stores `seq`
// there with a system-scope release after its results, which live in the same kind of memory.  The host spins on the word
// instead of calling hipStreamSynchronize, whose fixed cost (~10 us per call on the MI355X box, even when the stream is
// already idle) is paid once per linearisation otherwise.  ticket: one device int, zero before the first use.
struct HostSignal {
  int *ticket = nullptr;
  unsigned *flag = nullptr;   // nullptr: no signalling
  unsigned seq = 0;
};
// spin until *flag == seq; every ~64k polls the stream is queried so that a faulted kernel raises instead of hanging
inline void wait_host_signal(const HostSignal &sig, hipStream_t s) {
  const volatile unsigned *flag = sig.flag;
  for (unsigned long it = 1;; ++it) {
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == sig.seq) return;
    __builtin_ia32_pause();
    if ((it & 0xFFFFu) == 0) {
      const hipError_t e = hipStreamQuery(s);
      if (e == hipSuccess) {   // the stream has drained: kernel end made everything visible
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != sig.seq) throw DeviceError("completion word not written by a finished pass");
        return;
      }
      if (e != hipErrorNotReady) throw DeviceError(std::string("device pass failed: ") + hipGetErrorString(e));
    }
  }
}
// device side, called by one thread after the block's results are stored and fenced (__threadfence_system + barrier)
#if defined(__HIPCC__)
__device__ __forceinline__ void post_host_signal(const HostSignal &sig) {
  __threadfence_system();
  __hip_atomic_store(sig.flag, sig.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
#endif


inline int cdiv(long long a, long long b) { return int((a + b - 1) / b); }

}  // namespace lio
