// kf_batch.h — batched keyframe refinement (BASELINE.json configs[4]).
// Each keyframe runs the scan-to-map Gauss-Newton loop of MapBuilder::OptimizeMap (MapBuilder.cc:624-1014, 4-DoF) or
// PointMapping::OptimizeTransformTobeMapped (PointMapping.cc:325-753, 6-DoF) against its local map.  Keyframes are
// independent, so the batch is the parallel dimension: every stage of a round is ONE launch over all keyframes.
#pragma once
#include <memory>
#include <vector>

#include "../../include/lio_c.h"
#include "cloud_kernels.h"
#include "seg_sort.h"
#include "hmath.h"

namespace lio {

class KfBatchDev {
 public:
  explicit KfBatchDev(const lio_map_config &cfg);
  ~KfBatchDev();
  int AddMap(const float *corner, size_t nc, const float *surf, size_t ns);
  int AddKeyframe(int map, const float *corner, size_t nc, const float *surf, size_t ns, const Rigid<float> &T_init);
  void ClearKeyframes();
  void Refine();
  void BuildQueryOrder(hipStream_t s);
  // Refine + all-gather of the packed results (9 floats per keyframe: q, p, iterations, rows) over an RCCL communicator, straight
  // from the device pose buffer; packed_all (host) receives world * slots_per_rank * 9 floats
  void RefineGather(void *nccl_comm, int world, int slots_per_rank, float *packed_all);
  size_t n_keyframes() const { return h_kd_.size(); }
  size_t n_maps() const { return maps_.size(); }
  const std::vector<OdomState> &states() const { return h_st_; }
  int rounds_ = 0;          // rounds launched by the last Refine
  double device_ms_ = 0;    // HIP-event time of the last Refine's round loop
  long long n_queries_ = 0; // stack points over all keyframes

 private:
  struct Map {
    DBuf<float4> corner, surf;
    KnnGrid gc, gs;
    size_t nc = 0, ns = 0;
  };
  lio_map_config cfg_;
  hipStream_t stream_ = nullptr;
  hipEvent_t ev0_ = nullptr, ev1_ = nullptr;
  std::vector<std::unique_ptr<Map>> maps_;
  std::vector<KfMapDesc> h_md_;
  std::vector<float4> h_stack_;
  std::vector<KfDesc> h_kd_;
  std::vector<OdomState> h_st0_, h_st_;
  DBuf<KfMapDesc> d_md_;
  DBuf<float4> d_stack_, coef_;
  DBuf<KfDesc> d_kd_;
  DBuf<OdomState> d_st_;
  DBuf<uint8_t> valid_;
  DBuf<double> partials_;
  DBuf<int> d_nconv_;
  DBuf<float> d_pack_, d_gather_;
  // the queries' processing order (sorted by map cell under the keyframes' starting poses; rebuilt when the keyframe set changes) and the
  // segmented sort's scratch
  DBuf<uint32_t> order_, qkeys_, qkeys2_, qvals2_, qhist_;
  DBuf<SegDesc> d_qseg_;
  bool order_valid_ = false;
  int *h_nconv_ = nullptr;  // pinned
  bool md_dirty_ = true, kf_dirty_ = true;
  int max_Mc_ = 0, max_Ms_ = 0, max_nb_ = 1, total_nb_ = 0, n_gated_ = 0;
};

}  // namespace lio
