// rccl_comm.h — the two collectives of the path, kept behind plain pointers so that no RCCL type leaks into the other headers
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

struct lio_rccl;
namespace lio {
void *rccl_raw_comm(const lio_rccl *h);   // the ncclComm_t behind a handle
// in place, on stream s
void rccl_all_reduce_sum_f64(void *nccl_comm, double *dev_buf, size_t count, hipStream_t s);
void rccl_all_gather_f32(void *nccl_comm, const float *dev_send, float *dev_recv, size_t count_per_rank, hipStream_t s);
}  // namespace lio
