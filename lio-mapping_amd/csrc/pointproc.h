// pointproc.h — PointProcessor on the GPU (src/point_processor/PointProcessor.cc:185-783; §8a a1-a5).
//   ring_bin      : PointToRing (:207-426): elevation -> ring, azimuth -> rel time, stable per-ring order
//   ring_pick     : PrepareRing (:542-585) + PrepareSubregion (:587-622) + pick loops (:685-732) +
//                   MaskPickedInRing (:624-645) — one workgroup per ring, the whole ring in LDS
//   less_flat     : per-ring pcl::VoxelGrid(0.2) (:737-751) + rel-time recompute (:755-778), batched over rings
#pragma once
#include <chrono>
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
  int pad;
  long long pick_stamps[8];   // LIO_DEBUG_TIMING: wall-clock ticks of ring 0's block at the phase boundaries of k_ring_pick
};

// config_.infer_start_ori_ (PointProcessor.cc:348-387): two ten-deep histories of the start azimuth — as measured (buf2) and
// as used (buf1).  A start azimuth that jumps by more than rad_diff from the last used one is replaced by the last one plus
// the mean step of the used history; once the measured history steps evenly again (all nine steps within 0.05 rad of the used
// history's mean step, and the two mean steps within 0.05 rad) the azimuth of ring 0's first point is taken instead.
class StartOriFilter {
 public:
  // measured = azimuth of the sweep's first kept point; ring0_front = azimuth of ring 0's first point (NaN: ring 0 is empty —
  // the reference dereferences an empty cloud there; the value is then left as it is).  Returns start_ori_ for this sweep.
  float Update(float measured, float ring0_front, double rad_diff);
  void Reset() { n1_ = n2_ = h1_ = h2_ = 0; }

 private:
  static constexpr int kDepth = 10;
  float used_[kDepth], seen_[kDepth];   // start_ori_buf1_, start_ori_buf2_ (CircularBuffer<float>{10})
  int n1_ = 0, h1_ = 0, n2_ = 0, h2_ = 0;
  static void Push(float *buf, int &n, int &head, float v) {
    if (n < kDepth) buf[n++] = v; else { buf[head] = v; head = (head + 1) % kDepth; }
  }
};

class PointProcessorDev {
 public:
  PointProcessorDev(float lower, float upper, int rings, const lio_pp_config &cfg);
  ~PointProcessorDev();
  // ring != nullptr: the PointIR variant (ring field per point, rel-time over the swept range; PointProcessor.cc:428-536)
  void Process(const float *xyzi, size_t n, const uint16_t *ring = nullptr);
  size_t Count(int which) const;
  void GetCloud(int which, float *out);
  // the two halves of Process: Launch enqueues the whole sweep on the handle's stream and returns; Finish waits for it.  Several
  // handles launched one after the other keep that many sweeps in flight on one GPU (lio_pp_process_async / lio_pp_wait).
  void ProcessLaunch(const float *xyzi, size_t n, const uint16_t *ring = nullptr);
  void ProcessFinish();
  // B sweeps through ONE launch chain (lio_pp_process_batch): every kernel of the chain runs once over all sweeps (the sweep in
  // blockIdx.z), one copy brings every sweep's counts back.  xyzi[k]: host memory, or device memory when `on_device` (copied device
  // to device into the handle's segments).  filters[k] (may be null): the ten-sweep start-azimuth history sweep k belongs to
  // (infer_start_ori) — the handle's own when null and B = 1.  The accessors read the sweep SelectSweep() chose (0 after a launch).
  void ProcessLaunchBatch(const float *const *xyzi, const uint16_t *const *ring, const size_t *n, int B, bool on_device, StartOriFilter *const *filters);
  int sweeps() const { return nsw_; }
  void SelectSweep(int k);
  StartOriFilter &start_ori_filter() { return start_ori_filter_; }
  bool SameSensor(const PointProcessorDev &o) const;   // same constructor arguments: the two can share a launch chain
  float lower() const { return lower_; }
  float upper() const { return upper_; }
  int rings() const { return rings_; }
  const lio_pp_config &config() const { return cfg_; }
  void GetIndices(int which, int32_t *ring, int32_t *idx);
  void GetRingOffsets(int32_t *out);
  void GetRingIntensity(float *out);
  void GetCurvature(float *curv, int32_t *mask);
  // device-resident results (valid until the next Process)
  const float4 *d_less_flat() const { return less_flat_.p + size_t(sel_) * pts_stride_; }
  size_t n_less_flat() const { return size_t(counts_.n_less_flat); }
  float StartOri();   // start_ori_ of the selected sweep of the last Process (one small D2H unless infer_start_ori already fetched it)

 private:
  float lower_, upper_, factor_;
  int rings_;
  lio_pp_config cfg_;
  hipStream_t stream_ = nullptr;
  // sweeps of the last launch; the sweep the accessors read; elements per sweep of the point-indexed arrays, of one packed class list
  int nsw_ = 0, sel_ = 0;
  bool last_empty_ = false;            // the last launch had no point at all: nothing ran, every count reads zero
  std::vector<size_t> n_sw_;           // input points of the last launch's sweeps
  size_t pts_stride_ = 0, cls_stride_ = 0;
  int state_stride_ = 0;   // ints per sweep of the state record: PPDeviceCounts | ring offsets [LIO_PP_MAX_RINGS + 1] | first_valid [2] | end_ori | pad
  PPDeviceCounts counts_{};            // of the selected sweep
  std::vector<int> ring_offsets_;      // of the selected sweep
  // pinned landing zone: the state records of all sweeps, then three floats per sweep (start-azimuth probe x 2, the value used)
  int *h_state_ = nullptr;
  float *h_ori_ = nullptr;
  const float4 **h_ptr_ = nullptr;     // staging of the table of input pointers (sweeps already in device memory)
  int h_cap_sweeps_ = 0;
  void ReserveHost(int B);
  const int *h_record(int k) const { return h_state_ + size_t(k) * state_stride_; }
  DBuf<float4> in_, ring_cloud_, less_flat_, lf_tmp_, class_cloud_[4];
  DBuf<float> ring_intensity_;   // intensity_scans' intensity channel, ring order
  DBuf<float> azi_, curv_, start_ori_dev_;   // start_ori_dev_: 2 probes per sweep, then one override per sweep
  StartOriFilter start_ori_filter_;
  bool processed_ = false, start_ori_known_ = false, in_flight_ = false;
  std::chrono::steady_clock::time_point t_begin_{};
  DBuf<uint32_t> keys_;
  DBuf<int> ring_total_;
  DBuf<int> ring_table_;   // [ring][block] counts -> exclusive offsets (the stable ring split)
  DBuf<int> mask_;
  DBuf<int> d_state_;      // the sweeps' state records
  DBuf<int> d_n_;          // points per sweep
  DBuf<const float4 *> d_in_table_;   // where every sweep's input lies (lio_pp_process_batch_device)
  DBuf<uint16_t> ring_in_;
  DBuf<int8_t> label_;
  DBuf<int> pick_idx_, pick_cnt_, class_ring_, class_idx_;
  DBuf<int> lf_ring_count_;
};

}  // namespace lio
