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

inline int cdiv(long long a, long long b) { return int((a + b - 1) / b); }

}  // namespace lio
