// mapping.h — PointMapping (LOAM scan-to-map step + cube map) on the GPU.
// Reference: src/point_processor/PointMapping.cc:303-323 (point association), :325-753 (OptimizeTransformTobeMapped),
// :755-763 (TransformAssociateToMap / TransformUpdate), :765-1110 (Process), :1112-1208 (UpdateMapDatabase).
//
// MI355X-first layout: instead of 2 x 4851 per-cube clouds the map of one feature class is ONE device pool of
// float4 points plus a packed absolute cube key per point.  A cube's cloud is the subsequence of the pool with
// its key (pool order == the reference's per-cube order).  Shifting the 21x21x11 window only changes which keys
// are inside it; extracting the FOV cubes is one 8-bit radix pass + gather that leaves the pool as
// [rest | valid cubes in valid_idx order], so laser_cloud_*_from_map_ is a contiguous tail of the pool; the per-cube
// VoxelGrid of UpdateMapDatabase is ONE segmented voxel pass with 64-bit (cube rank, voxel) keys.
#pragma once
#include <vector>

#include "../../include/lio_c.h"
#include "cloud_kernels.h"
#include "hmath.h"

namespace lio {

#define LIO_MAP_MAX_VALID 125
#define LIO_MAP_RANK_REST 254u
#define LIO_MAP_RANK_DROP 255u

struct MapValidSet {
  uint32_t key[LIO_MAP_MAX_VALID];  // packed absolute cube keys, ascending == valid_idx order (i-major)
  int n;
  int lo[3], hi[3];                 // absolute cube range [lo, hi) of the current 21x21x11 window
};

struct MapCounters { int n_valid, n_rest, n_new_valid, n_new_rest, n_out, pad[3]; };

class MappingDev {
 public:
  static constexpr int L = 21, Wd = 21, H = 11;  // laser_cloud_length_/width_/height_ (PointMapping.cc:79-81)

  explicit MappingDev(const lio_map_config &cfg);
  ~MappingDev();
  MappingDev(const MappingDev &) = delete;
  MappingDev &operator=(const MappingDev &) = delete;

  void Process(const float *corner_last, size_t n_corner, const float *surf_last, size_t n_surf, const Rigid<float> &transform_sum);
  void UpdateMapDatabase(const float *corner_ds, size_t n_corner, const float *surf_ds, size_t n_surf, const uint32_t *valid_idx, size_t n_valid,
                         const Rigid<float> &T, const int cube_center[3]);
  size_t GetCloud(int which, float *out);
  size_t GetCube(int cls, uint32_t cube_idx, float *out);
  size_t GetScorePointCoeff(float *score, float *point, float *coeff);
  // laser_cloud_{corner,surf}_stack_downsampled_ where they live (HBM), for the estimator's pre-initialisation pushes
  const float4 *StackDevice(int cls) const { return cls_[cls].stack_ds.p; }
  size_t StackSize(int cls) const { return cls_[cls].n_stack; }
  void Sync() { LIO_HIP(hipStreamSynchronize(stream_)); }

  Rigid<float> transform_sum_, transform_tobe_mapped_, transform_bef_mapped_, transform_aft_mapped_;
  bool imu_inited_ = false;
  int cen_[3] = {10, 10, 5};  // laser_cloud_cen_length_/width_/height_ (:76-78)
  std::vector<uint32_t> valid_idx_;
  int iterations_ = 0, num_selected_ = 0;
  bool degenerate_ = false;
  int kz_ = 0;   // leading update components masked by round 0's degeneracy test (PointMapping.cc:650-680)

 private:
  struct ClassMap {
    DBuf<float4> pool, pool2;       // points in the map frame; pool2 = gather target (swapped in)
    DBuf<uint32_t> pkey, pkey2;     // packed absolute cube key per pool point
    DBuf<uint32_t> vrank;           // cube rank of the valid tail
    size_t n = 0, n_rest = 0, n_valid = 0;
    bool layout_ok = false;
    std::vector<uint32_t> layout_keys;
    int layout_lo[3] = {0, 0, 0}, layout_hi[3] = {0, 0, 0};
    // scratch
    DBuf<uint32_t> rk, rk2, vals, vals2;
    DBuf<char> tmp;
    DBuf<MapCounters> counters;
    DBuf<float4> new_pts, u_pts;
    DBuf<uint32_t> new_key, u_rank;
    DBuf<unsigned long long> k64, k64b;
    DBuf<int> flags, pos;
    DBuf<int> cube_bounds;          // [LIO_MAP_MAX_VALID][6] order-preserving int images of min/max
    DBuf<float> partial;
    DBuf<VoxParams> bounds;
    KnnGrid grid;
    VoxelGridDev vox;
    DBuf<float4> in, stack_raw, stack_ds;
    size_t n_stack = 0;
    MapCounters *h_counters = nullptr;  // pinned
    VoxParams *h_bounds = nullptr;      // pinned
  };

  MapValidSet MakeValidSet(const uint32_t *valid_idx, size_t n, const int cen_of_idx[3]) const;
  bool LayoutMatches(const ClassMap &m, const MapValidSet &vs) const;
  void LayoutLaunch(ClassMap &m, const MapValidSet &vs);   // rank + sort + gather, counters D2H queued
  void LayoutFinish(ClassMap &m, const MapValidSet &vs);   // after a stream sync
  void UpdateLaunch(ClassMap &m, const float4 *new_sensor_pts, size_t n_new, const MapValidSet &vs, const Rigid<float> &T, float leaf);
  void UpdateFinish(ClassMap &m);
  void Optimize(bool four_dof);

  lio_map_config cfg_;
  hipStream_t stream_ = nullptr, stream2_ = nullptr;   // stream2_: the surf stack's VoxelGrid beside the corner one
  hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr;
  ClassMap cls_[2];  // 0 corner, 1 surf
  float pz_[3] = {0, 0, 10};
  DBuf<float4> stack_all_;
  DBuf<uint8_t> f_valid_;
  DBuf<float4> f_coef_, f_abs_;
  DBuf<OdomState> d_state_;
  DBuf<double> d_partials_;
  OdomState *h_state_ = nullptr;  // pinned, coherent: the state's mailbox
  unsigned *h_flag_ = nullptr;    // its completion word
  unsigned seq_ = 0;
  size_t n_score_slots_ = 0;
  bool score_ready_ = false;
  size_t n_from_map_[2] = {0, 0};  // sizes of laser_cloud_{corner,surf}_from_map_ of the last Process
  bool from_map_in_u_ = false;      // the map update moved them into the work list
  bool system_init_ = false;        // MapBuilder.h:65
  int odom_count_ = 0;              // MapBuilder.h:69
};

}  // namespace lio
