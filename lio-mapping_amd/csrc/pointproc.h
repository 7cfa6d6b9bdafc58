// pointproc.h — PointProcessor on the GPU (src/point_processor/PointProcessor.cc:185-783; §8a a1-a5).
//   ring_bin      : PointToRing (:207-426): elevation -> ring, azimuth -> rel time, stable per-ring order
//   ring_pick     : PrepareRing (:542-585) + PrepareSubregion (:587-622) + pick loops (:685-732) +
//                   MaskPickedInRing (:624-645) — one workgroup per ring, the whole ring in LDS
//   less_flat     : per-ring pcl::VoxelGrid(0.2) (:737-751) + rel-time recompute (:755-778), batched over rings
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/lio_c.h"
#include "cloud_kernels.h"

#define LIO_PP_MAX_RINGS 128
#define LIO_PP_MAX_RING_POINTS 4080   // 8 subregions of <= 512 sort slots each, ring resident in LDS

namespace lio {

struct PPDeviceCounts {
  int n_ring_points;     // points kept by ring binning
  int n_less_flat;       // voxel-filtered less-flat points
  int n_class[4];        // [1] sharp, [2] less_sharp, [3] flat
  int overflow;          // a ring exceeded LIO_PP_MAX_RING_POINTS
};

class PointProcessorDev {
 public:
  PointProcessorDev(float lower, float upper, int rings, const lio_pp_config &cfg);
  ~PointProcessorDev();
  // ring != nullptr: the PointIR variant (ring field per point, rel-time over the swept range; PointProcessor.cc:428-536)
  void Process(const float *xyzi, size_t n, const uint16_t *ring = nullptr);
  size_t Count(int which) const;
  void GetCloud(int which, float *out);
  void GetIndices(int which, int32_t *ring, int32_t *idx);
  void GetRingOffsets(int32_t *out);
  void GetCurvature(float *curv, int32_t *mask);
  // device-resident results (valid until the next Process)
  const float4 *d_less_flat() const { return less_flat_.p; }
  size_t n_less_flat() const { return size_t(counts_.n_less_flat); }

 private:
  float lower_, upper_, factor_;
  int rings_;
  lio_pp_config cfg_;
  hipStream_t stream_ = nullptr;
  PPDeviceCounts counts_{};
  struct HostOut { PPDeviceCounts counts; int ring_offsets[LIO_PP_MAX_RINGS + 1]; };
  HostOut *h_out_ = nullptr;   // pinned landing zone of the per-sweep results
  std::vector<int> ring_offsets_;
  DBuf<float4> in_, ring_cloud_, less_flat_, lf_tmp_, class_cloud_[4];
  DBuf<float> azi_, curv_;
  DBuf<uint32_t> keys_;
  DBuf<int> ring_total_;
  DBuf<int> ring_table_;   // [ring][block] counts -> exclusive offsets (the stable ring split)
  DBuf<int> d_ring_offsets_, first_valid_, mask_, end_ori_;
  DBuf<uint16_t> ring_in_;
  DBuf<int8_t> label_;
  DBuf<int> pick_idx_, pick_cnt_, class_ring_, class_idx_;
  DBuf<int> lf_ring_count_;
  DBuf<PPDeviceCounts> d_counts_;
};

}  // namespace lio
