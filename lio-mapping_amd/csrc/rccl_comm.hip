// rccl_comm.hip — in-library RCCL for the two exchange steps the path has:
//   * factor sharding: SUM all-reduce of the per-shard normal-equation moments (Wo x 260 doubles) on the estimator's stream,
//     between the fold kernel and the host read — the cross-rank form of the reference's four-thread partial sums
//     (MarginalizationFactor.cc:245-269);
//   * keyframe batch: all-gather of the refined poses straight from the device pose buffer.
// One communicator per process (= per GPU), created from a unique id the caller distributes by whatever side channel it has
// (bench.py: torch.distributed broadcast).  xGMI is point-to-point; both messages are tiny (10 KB / 36 B per keyframe), so they
// are latency-bound: one collective per linearisation, nothing chunked.
#include <rccl/rccl.h>

#include <cstring>

#include "../../include/lio_c.h"
#include "dev.h"
#include "rccl_comm.h"

namespace lio {

#define LIO_NCCL(call)                                                                                             \
  do {                                                                                                             \
    ncclResult_t r__ = (call);                                                                                     \
    if (r__ != ncclSuccess) throw DeviceError(std::string(#call) + " -> " + ncclGetErrorString(r__));              \
  } while (0)

void rccl_all_reduce_sum_f64(void *comm, double *dev_buf, size_t count, hipStream_t s) {
  LIO_NCCL(ncclAllReduce(dev_buf, dev_buf, count, ncclDouble, ncclSum, static_cast<ncclComm_t>(comm), s));
}
void rccl_all_gather_f32(void *comm, const float *dev_send, float *dev_recv, size_t count_per_rank, hipStream_t s) {
  LIO_NCCL(ncclAllGather(dev_send, dev_recv, count_per_rank, ncclFloat, static_cast<ncclComm_t>(comm), s));
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
  if (ncclGetUniqueId(&u) != ncclSuccess) return LIO_ERR_DEVICE;
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
  if (ncclCommInitRank(&h->comm, world, u, rank) != ncclSuccess) { delete h; return nullptr; }
  h->rank = rank; h->world = world;
  return h;
}

void lio_rccl_destroy(lio_rccl *h) {
  if (!h) return;
  if (h->comm) (void)ncclCommDestroy(h->comm);
  delete h;
}

int lio_rccl_rank(const lio_rccl *h) { return h ? h->rank : -1; }
int lio_rccl_world(const lio_rccl *h) { return h ? h->world : 0; }

}  // extern "C"

namespace lio {
void *rccl_raw_comm(const lio_rccl *h) { return h ? static_cast<void *>(h->comm) : nullptr; }
}  // namespace lio
