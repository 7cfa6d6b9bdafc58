// dev.h — HIP runtime plumbing for the product: error propagation, growable device buffers, timers.
// No CPU fallback anywhere: a failed HIP call surfaces as LIO_ERR_DEVICE through the C-ABI.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

namespace lio {

// A pointer READ FROM MEMORY (a field of a descriptor in device memory) is generic to the compiler: every access through it becomes a
// FLAT instruction, which counts in lgkmcnt as well as vmcnt — so each wait for an LDS operation also waits for every such load in
// flight, and a phase that mixes LDS work with descriptor-addressed global loads runs one memory round trip at a time (the prior pass
// of the batched step kernel: 28 k clocks for 36 loads per thread, profiles/r5_i_*).  Neither an assumption (is_shared / is_private
// false) nor a cast through address space 1 and back survives to the pass that decides; what does is arithmetic on a pointer that
// arrived as a KERNEL ARGUMENT (those are global): base + (p - base), with `base` the start of the allocation p points into.
#if defined(__HIPCC__)
template <class T, class U>
__device__ __forceinline__ U *rebase(T *base, U *p) {
  // (null stays null explicitly: base + offset is never null to the compiler, which would drop the callers' null checks)
  if (!p) return nullptr;
  return reinterpret_cast<U *>(reinterpret_cast<char *>(const_cast<typename std::remove_const<T>::type *>(base)) +
                               (reinterpret_cast<const char *>(p) - reinterpret_cast<const char *>(base)));
}
#endif


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

// Completion words in coherent (fine-grained) pinned host memory.  A kernel's block stores its results to memory of the same
// kind, waits until those stores are acknowledged, and then stores `seq` into ITS word; the host spins until all `nslots`
// words carry `seq` instead of calling hipStreamSynchronize, whose fixed cost (~10 us per call on the MI355X box, even when
// the stream is already idle) is otherwise paid once per linearisation.  No cross-block traffic: a ticket protocol needs
// agent-scope release fences, and on the eight-L2 MI355X each of those writes back the L2's dirty lines (measured: a reduce
// kernel went from 4.8 to 8 us, a bounds kernel from 4.6 to 14.6 us).
struct HostSignal {
  unsigned *flag = nullptr;   // nullptr: no signalling; else nslots words
  unsigned seq = 0;
  int nslots = 1;
};
// spin until every word == seq; every ~64k polls the stream is queried so that a faulted kernel raises instead of hanging
inline void wait_host_signal(const HostSignal &sig, hipStream_t s) {
  const volatile unsigned *flag = sig.flag;
  int k = 0;
  for (unsigned long it = 1;; ++it) {
    while (k < sig.nslots && __atomic_load_n(flag + k, __ATOMIC_ACQUIRE) == sig.seq) ++k;
    if (k == sig.nslots) return;
    __builtin_ia32_pause();
    if ((it & 0xFFFFu) == 0) {
      const hipError_t e = hipStreamQuery(s);
      if (e == hipSuccess) {   // the stream has drained: kernel end made everything visible
        for (k = 0; k < sig.nslots; ++k)
          if (__atomic_load_n(flag + k, __ATOMIC_ACQUIRE) != sig.seq) throw DeviceError("completion word not written by a finished pass");
        return;
      }
      if (e != hipErrorNotReady) throw DeviceError(std::string("device pass failed: ") + hipGetErrorString(e));
    }
  }
}
#if defined(__HIPCC__)
// Device side.  Results go out through host_store(): a system-scope store (sc0 sc1 on gfx950) is written through to the
// host's memory and acknowledged only then, so once the wave's stores are acknowledged (vmcnt 0; gfx9 counts stores there)
// the completion word cannot overtake them.  A plain store may be acknowledged by the L2 and pass the word on its way out;
// the system-scope release FENCE that would also order plain stores writes back every dirty line of the L2 (see above).
// A block whose results were stored by several waves calls host_signal_drain() in every storing thread, then a barrier,
// then one thread posts.
template <typename T>
__device__ __forceinline__ void host_store(T *dst, T v) { __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void host_signal_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void post_host_signal(const HostSignal &sig, int slot = 0) {
  host_signal_drain();
  host_store(sig.flag + slot, sig.seq);
}
// A small POD (<= 64 32-bit words) to the host's mailbox and then the completion word, by ONE WAVE: every lane of the wave
// calls this (after a barrier that made `src` visible); lane k carries word k, so the words travel together — one thread
// storing them one after the other pays a host round trip per word (measured: 13 words, +15 us).
__device__ __forceinline__ void post_host_mail(const HostSignal &sig, void *dst, const void *src, int nwords, int lane) {
  if (lane < nwords) host_store(reinterpret_cast<unsigned *>(dst) + lane, reinterpret_cast<const unsigned *>(src)[lane]);
  host_signal_drain();
  if (lane == 0) host_store(sig.flag, sig.seq);
}
#endif

inline int cdiv(long long a, long long b) { return int((a + b - 1) / b); }

}  // namespace lio
