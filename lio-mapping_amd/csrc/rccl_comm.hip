// rccl_comm.hip — in-library RCCL for the two exchange steps the path has:
//   * factor sharding: SUM all-reduce of the per-shard normal-equation moments (Wo x 260 doubles) on the estimator's stream,
//     between the fold kernel and the host read — the cross-rank form of the reference's four-thread partial sums
//     (MarginalizationFactor.cc:245-269);
//   * keyframe batch: all-gather of the refined poses straight from the device pose buffer.
// One communicator per process (= per GPU), created from a unique id the caller distributes by whatever side channel it has
// (bench.py: torch.distributed broadcast).  xGMI is point-to-point; both messages are tiny (10 KB / 36 B per keyframe), so they
// are latency-bound: one collective per linearisation, nothing chunked.
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and enums only: the library itself is loaded on first use

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../../include/lio_c.h"
#include "dev.h"
#include "rccl_comm.h"

namespace lio {

// RCCL is needed by the opt-in lio_rccl_* entry points only, so liblio_hip.so does not link it: the single-GPU product loads on
// a host without RCCL (or with ROCm installed elsewhere).  First use resolves the six calls from librccl by SONAME — inside a
// process that already carries RCCL (torch) that is the SAME library object, so both share one set of communicator internals —
// then from $ROCM_PATH/lib and /opt/rocm/lib.
struct RcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;      // optional: lio_rccl_world / lio_rccl_rank ask the communicator
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
  bool ok = false;
};
static const RcclApi &rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    std::string rocm = std::getenv("ROCM_PATH") ? std::getenv("ROCM_PATH") : "/opt/rocm";
    const std::string names[] = {"librccl.so.1", "librccl.so", rocm + "/lib/librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const std::string &n : names)
      if ((h = dlopen(n.c_str(), RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return;
    auto sym = [&](const char *n) { return dlsym(h, n); };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    api.CommCount = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
    api.CommUserRank = reinterpret_cast<decltype(api.CommUserRank)>(sym("ncclCommUserRank"));
    api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce && api.AllGather && api.GetErrorString;
  });
  return api;
}
static const RcclApi &rccl_api_or_throw() {
  const RcclApi &a = rccl_api();
  if (!a.ok) throw DeviceError("librccl could not be loaded: the lio_rccl_* entry points need RCCL on this host");
  return a;
}

#define LIO_NCCL(call)                                                                                             \
  do {                                                                                                             \
    ncclResult_t r__ = (call);                                                                                     \
    if (r__ != ncclSuccess) throw ::lio::DeviceError(std::string(#call) + " -> " + ::lio::rccl_api().GetErrorString(r__)); \
  } while (0)

void rccl_all_reduce_sum_f64(void *comm, double *dev_buf, size_t count, hipStream_t s) {
  LIO_NCCL(rccl_api_or_throw().AllReduce(dev_buf, dev_buf, count, ncclDouble, ncclSum, static_cast<ncclComm_t>(comm), s));
}
void rccl_all_gather_f32(void *comm, const float *dev_send, float *dev_recv, size_t count_per_rank, hipStream_t s) {
  LIO_NCCL(rccl_api_or_throw().AllGather(dev_send, dev_recv, count_per_rank, ncclFloat, static_cast<ncclComm_t>(comm), s));
}

}  // namespace lio

struct lio_rccl {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
};

extern "C" {

int lio_rccl_unique_id(unsigned char id[LIO_RCCL_ID_BYTES]) {
  if (!id) return LIO_ERR_ARG;
  static_assert(sizeof(ncclUniqueId) == LIO_RCCL_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId u;
  const lio::RcclApi &api = lio::rccl_api();
  if (!api.ok || api.GetUniqueId(&u) != ncclSuccess) return LIO_ERR_DEVICE;
  std::memcpy(id, &u, sizeof(u));
  return LIO_OK;
}

lio_rccl *lio_rccl_init(const unsigned char id[LIO_RCCL_ID_BYTES], int rank, int world) {
  if (!id || world < 1 || rank < 0 || rank >= world) return nullptr;
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) return nullptr;
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  lio_rccl *h = new (std::nothrow) lio_rccl;
  if (!h) return nullptr;
  const lio::RcclApi &api = lio::rccl_api();
  if (!api.ok || api.CommInitRank(&h->comm, world, u, rank) != ncclSuccess) { delete h; return nullptr; }
  h->rank = rank; h->world = world;
  return h;
}

void lio_rccl_destroy(lio_rccl *h) {
  if (!h) return;
  if (h->comm && lio::rccl_api().ok) (void)lio::rccl_api().CommDestroy(h->comm);
  delete h;
}

int lio_rccl_bench_all_reduce(lio_rccl *h, int count, int reps, double *avg_us) {
  if (!h || !h->comm || count < 1 || reps < 1 || !avg_us) return LIO_ERR_ARG;
  try {
    const lio::RcclApi &api = lio::rccl_api_or_throw();
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    double *buf = nullptr;
    LIO_HIP(hipStreamCreate(&s));
    LIO_HIP(hipEventCreate(&e0)); LIO_HIP(hipEventCreate(&e1));
    LIO_HIP(hipMalloc(reinterpret_cast<void **>(&buf), sizeof(double) * size_t(count)));
    LIO_HIP(hipMemsetAsync(buf, 0, sizeof(double) * size_t(count), s));
    for (int k = 0; k < 3; ++k) LIO_NCCL(api.AllReduce(buf, buf, size_t(count), ncclDouble, ncclSum, h->comm, s));   // warm-up: channel setup
    LIO_HIP(hipEventRecord(e0, s));
    for (int k = 0; k < reps; ++k) LIO_NCCL(api.AllReduce(buf, buf, size_t(count), ncclDouble, ncclSum, h->comm, s));
    LIO_HIP(hipEventRecord(e1, s));
    LIO_HIP(hipStreamSynchronize(s));
    float ms = 0;
    LIO_HIP(hipEventElapsedTime(&ms, e0, e1));
    *avg_us = 1e3 * double(ms) / reps;
    (void)hipFree(buf); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(s);
    return LIO_OK;
  } catch (...) { return LIO_ERR_DEVICE; }
}
// what the COMMUNICATOR reports (ncclCommUserRank / ncclCommCount), not what the caller passed to lio_rccl_init: bench.py prints
// it next to n_gpus so that a line claiming N ranks shows RCCL agreeing
int lio_rccl_rank(const lio_rccl *h) {
  if (!h) return -1;
  int r = h->rank;
  const lio::RcclApi &api = lio::rccl_api();
  if (h->comm && api.CommUserRank && api.CommUserRank(h->comm, &r) != ncclSuccess) return -1;
  return r;
}
int lio_rccl_world(const lio_rccl *h) {
  if (!h) return 0;
  int n = h->world;
  const lio::RcclApi &api = lio::rccl_api();
  if (h->comm && api.CommCount && api.CommCount(h->comm, &n) != ncclSuccess) return 0;
  return n;
}

}  // extern "C"

namespace lio {
void *rccl_raw_comm(const lio_rccl *h) { return h ? static_cast<void *>(h->comm) : nullptr; }
}  // namespace lio
