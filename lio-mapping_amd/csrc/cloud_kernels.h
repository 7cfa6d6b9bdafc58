// cloud_kernels.h — device-side point-cloud stages of the estimator (host-callable launchers).
//   rigid transform / concat    : pcl::transformPointCloud call sites Estimator.cc:1425,1498,2602
//   VoxelGridDev                : pcl::VoxelGrid<PointXYZI> call sites Estimator.cc:678-687,1518-1519
//   KnnGrid + feature kernels   : pcl::KdTreeFLANN + Estimator::CalculateFeatures (Estimator.cc:970-1097)
//   laser-odom rows/update      : Estimator::CalculateLaserOdom (Estimator.cc:1242-1359)
//   deskew                      : TransformToEnd (Estimator.cc:62-103)
#pragma once
#include <cstdint>

#include "dev.h"

namespace lio {

struct VoxParams {
  float mn[3], mx[3];
  int minb[3], divb[3];
  int overflow;  // PCL's "leaf size too small" guard: output = input
  int n_valid;
};

struct Affine3f { float m[12]; };  // row-major 3x4: [R | t]

#define LIO_MAX_FRAMES 32

struct ConcatSeg { const float4 *src; int n; int dst_off; int set_intensity; float intensity; Affine3f tf; int identity; };
struct ConcatArgs { ConcatSeg seg[LIO_MAX_FRAMES]; int nseg; int total; };
// dst[seg.dst_off + k] = tf * src[k] (or src[k] when identity); intensity optionally overwritten
void launch_transform_concat(const ConcatArgs &a, float4 *dst, hipStream_t s);

// TransformToEnd (Estimator.cc:62-103) in place; tes = q(xyzw), p
void launch_deskew_to_end(float4 *pts, int n, const float q[4], const float p[3], float time_factor, hipStream_t s);

// min/max (VoxParams.mn/.mx, n_valid) of a device cloud; `partial` is scratch.  No host sync.
void launch_cloud_bounds(const float4 *pts, int n, DBuf<float> &partial, VoxParams *d_out, hipStream_t s);

class VoxelGridDev {
 public:
  // Filters `in` (device, n points) with cubic leaf; result in `out`; returns the output count (host sync).
  // host_params (optional) receives the bounds used.
  size_t run(const float4 *in, size_t n, float leaf, DBuf<float4> &out, hipStream_t s, VoxParams *host_params = nullptr);
  // the same in two halves: launch() only enqueues; finish() returns the count as soon as the device has posted it — the
  // centroids themselves may still be in flight on `s`, so consumers on OTHER streams must wait for `s` through an event
  void launch(const float4 *in, size_t n, float leaf, DBuf<float4> &out, hipStream_t s);
  size_t finish(VoxParams *host_params = nullptr);
  void set_host_signal(bool on) { use_signal_ = on; }   // false: counts come back by copy + hipStreamSynchronize (lio_est_config.stream_sync)
  VoxelGridDev() = default;
  VoxelGridDev(const VoxelGridDev &) = delete;
  VoxelGridDev &operator=(const VoxelGridDev &) = delete;
  ~VoxelGridDev();

 private:
  const float4 *p_in_ = nullptr; size_t p_n_ = 0; DBuf<float4> *p_out_ = nullptr; hipStream_t p_stream_ = nullptr;  // the pending launch
  float p_leaf_ = 0.f;
  void enqueue(bool exact);
  int *h_count_ = nullptr;          // pinned: output count, followed by the VoxParams
  VoxParams *h_params_ = nullptr;
  DBuf<float> partial_;
  DBuf<VoxParams> params_;
  DBuf<uint32_t> keys_, keys2_, vals_, vals2_;
  DBuf<int> count_, tile_heads_;
  DBuf<char> tmp_;
  unsigned *h_flag_ = nullptr;      // completion word behind the mailbox (dev.h: HostSignal)
  unsigned seq_ = 0;
  HostSignal sig_{};
  bool use_signal_ = true;
};
// LIO_HOST_SIGNAL=0: every wait is a hipStreamSynchronize again
bool host_signal_enabled();

struct GridDesc {
  int origin[3];   // cell coordinate of cell (0,0,0)
  int dims[3];
  float inv_cell;
  int n_points;
};

class KnnGrid {
 public:
  // Uniform grid over `pts` (device, n points).  bounds = min/max of the region to index (host values); points
  // outside it are clamped into the border cells (still compared by true distance).
  void build(const float4 *pts, size_t n, const float mn[3], const float mx[3], float cell, hipStream_t s);
  const float4 *sorted() const { return sorted_.p; }   // xyz + original index in .w (int bits)
  const int *cells() const { return cells_.p; }         // ncells + 1 run starts: cell c holds sorted()[cells[c] .. cells[c + 1])
  const GridDesc &desc() const { return desc_; }

 private:
  GridDesc desc_{};
  DBuf<uint32_t> keys_, keys2_, vals_, vals2_;
  DBuf<float4> sorted_;
  DBuf<int> cells_, cnt_;
  bool cnt_dirty_ = false;   // k_cell_count ran, k_cell_place (which re-zeroes cnt_) did not
  DBuf<char> tmp_;
};

// order (optional): the queries are PROCESSED in this order — order[t] = slot (slot_off + index into stack) of the t-th query — so that the lanes
// of a wave search neighbouring map cells and their gathers fall into the same cache lines; results go to the queries' own slots, so nothing
// downstream sees the order
struct FeatFrame { const float4 *stack; int M; int slot_off; int tf_index; const uint32_t *order = nullptr; };
struct FeatArgs {
  FeatFrame fr[LIO_MAX_FRAMES];
  int nframes;
  int max_M;
  float min_match_sq_dis, min_plane_dis;
  // scan-to-map variant (PointMapping.cc:519-619): coefficient sign follows pd2, coef.w = s*pd2, the FOV apex
  // point_on_z_axis_ is the one fixed before the iterations (:803-806), abs_coeff is written when requested
  int mapping_mode;   // 0 estimator, 1 PointMapping, 2 MapBuilder::OptimizeMap (no sign flip)
  float fixed_pz[3];
};
// transforms: device array of 8 floats per entry (qx,qy,qz,qw,px,py,pz,pad).  skip_flag: optional device int;
// when *skip_flag != 0 the launch is a no-op (converged laser-odom loop).
void launch_features(const FeatArgs &a, const float *transforms, const float4 *map_sorted, const int *cells, const GridDesc &g,
                     uint8_t *valid, float4 *coef, float *score, const int *skip_flag, hipStream_t s, float4 *abs_coef = nullptr);

// Corner branch of PointMapping::OptimizeTransformTobeMapped (PointMapping.cc:377-517): 5-NN in the corner map, 3x3
// covariance eigen-decomposition, line residual; slots [slot_off, slot_off+M).  transform: device, 8 floats.
void launch_line_features(const float4 *stack, int M, int slot_off, const float *transform, const float fixed_pz[3], float min_match_sq_dis,
                          const float4 *map_sorted, const int *cells, const GridDesc &g, uint8_t *valid, float4 *coef,
                          const int *skip_flag, hipStream_t s);

// stateless K-NN (lio_knn entry point): idx/sqd are m*k
void launch_knn(const float4 *query, int m, int k, float radius_sq, const float4 *map_sorted, const int *cells, const GridDesc &g,
                int32_t *idx, float *sqd, hipStream_t s);

// Both branches of one scan-to-map round in a single launch (surf: FeatArgs with one frame, mapping_mode 1 or 2; corner: the
// first Mc points of the concatenated stack, slots [0, Mc)).
void launch_map_round(const FeatArgs &surf, const float4 *corner_stack, int Mc, const float *transform, const float4 *corner_map,
                      const int *corner_cells, const GridDesc &corner_grid, const float4 *surf_map, const int *surf_cells,
                      const GridDesc &surf_grid, uint8_t *valid, float4 *coef, float4 *abs_coef, const int *skip_flag, hipStream_t s);

// What a solve's feature stage starts from, in ONE launch instead of a fill and two small uploads (each a command of its own in
// the stream, the uploads staged through the runtime's pinned buffer): valid[0, n_valid) <- 0, the (W + 1) x 8 local transforms,
// the newest frame's Gauss-Newton state.
struct SolveSetup {
  float tf[LIO_MAX_FRAMES][8];
  int ntf;
  float odom_T[8];
  int set_odom;
};
struct OdomState;
void launch_solve_setup(const SolveSetup &a, float *d_transforms, OdomState *d_odom, uint8_t *valid, size_t n_valid, hipStream_t s);

struct OdomState {
  float T[8];        // qx,qy,qz,qw,px,py,pz,pad : local_transform of the newest frame
  int converged;
  int iters;
  int degenerate;
  int kz;            // number of leading components masked (A.6)
  int nsel;          // rows selected in the last round
};
// rows of mat_A / mat_B (Estimator.cc:1272-1301) over slots [0,nslots) of the newest frame, reduced to
// per-block partials (28 doubles each).  Point of slot s = stack[s % M].
// b_from_coef != 0: mat_B = -coef.w (the distance stored at feature time, PointMapping.cc:640,650); 2 additionally maps the
// rotation columns into the map frame with the MapBuilder weights (MapBuilder.cc:903-914).
void launch_odom_rows(const float4 *stack, int M, int nslots, const uint8_t *valid, const float4 *coef, const OdomState *st,
                      double *partials, int nblocks, hipStream_t s, int b_from_coef = 0);
// reduce + 6x6 solve + degeneracy mask + transform update + convergence test (Estimator.cc:1303-1357)
// min_rows > 0: a round with fewer selected rows leaves the transform untouched (`continue`, PointMapping.cc:623-626).
// left_update != 0: rot = DeltaQ(x) * rot (MapBuilder.cc:978-979) instead of rot * DeltaQ(x).
void launch_odom_update(const double *partials, int nblocks, OdomState *st, int iter, hipStream_t s, int min_rows = 0, int left_update = 0,
                        OdomState *mail = nullptr, const HostSignal &sig = HostSignal());
int odom_rows_blocks(int nslots);
// One round of the newest frame's loop in two launches (search + plane fit + rows per block; fold + update).  a.fr[0] names the
// stack; slots of round r start at base_slot (+ r * M with keep != 0: keep_features, Estimator.cc:978-980); partials holds
// odom_round_blocks(M, lanes per query) x 28 doubles.
int odom_round_blocks(int M, int lpq);
void launch_odom_round(const FeatArgs &a, int base_slot, int round, int keep, OdomState *st, const float4 *map_sorted, const int *cells, const GridDesc &g,
                       uint8_t *valid, float4 *coef, float *score, double *partials, hipStream_t s, OdomState *mail = nullptr,
                       const HostSignal &sig = HostSignal(), int lpq = 8);

// ---- batched keyframe refinement (config 5: B independent OptimizeMap / OptimizeTransformTobeMapped loops, MapBuilder.cc:624-1014,
// PointMapping.cc:325-753).  Slots of keyframe k = [slot_off, slot_off + Mc) corner, then Ms surf, in one concatenated stack.
struct KfDesc {
  int slot_off, Mc, Ms;
  int map;          // index into the KfMapDesc array
  int nb;           // row-reduction blocks of this keyframe (= odom_rows_blocks(Mc + Ms), as in the single-keyframe path)
  int part_off;     // first partial row (28 doubles each)
  float pz[3];      // point_on_z_axis_ fixed before the iterations
};
struct KfMapDesc {
  const float4 *corner_sorted; const int *corner_cells; GridDesc corner_grid;
  const float4 *surf_sorted; const int *surf_cells; GridDesc surf_grid;
};
// grid z limit: n_keyframes <= 65535 per launch
// order_or_null: processing order of every keyframe's queries (global slots; a keyframe's corner queries at [slot_off, slot_off + Mc), its surf
// queries behind them), e.g. sorted by map cell (launch_kf_query_keys + the segmented sort)
void launch_kf_query_keys(const KfDesc *kd, const KfMapDesc *md, const OdomState *st, int n_keyframes, int max_Mc, int max_Ms, const float4 *stack_all, uint32_t *keys,
                          hipStream_t s);
void launch_kf_round(const KfDesc *kd, const KfMapDesc *md, const OdomState *st, int n_keyframes, int max_Mc, int max_Ms, long long total_queries,
                     const float4 *stack_all, const uint32_t *order_or_null, float min_match_sq_dis, float min_plane_dis, int mapping_mode, uint8_t *valid, float4 *coef,
                     hipStream_t s);
void launch_kf_rows(const KfDesc *kd, const OdomState *st, int n_keyframes, int max_nb, const float4 *stack_all, const uint8_t *valid, const float4 *coef,
                    double *partials, int b_from_coef, hipStream_t s);
// n_converged (device int) is incremented once per keyframe when it converges
void launch_kf_update(const KfDesc *kd, OdomState *st, int n_keyframes, const double *partials, int iter, int min_rows, int left_update, int *n_converged,
                      hipStream_t s);

#if defined(__HIPCC__)
// Fixed-order sum of `nblocks` rows of 28 doubles by a 256-thread block: 8 groups of 32 lanes take the rows b = g (mod 8)
// in ascending order, then the 8 group sums are added in ascending g.  (One lane per column walking all rows issues
// ~nblocks dependent-latency loads: 18 us for 38 rows.)
__device__ inline void reduce_partials28(const double *__restrict__ partials, int nblocks, double *ssum /* shared, >= 28 */) {
  __shared__ double part28[8][32];
  const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
  double v = 0;
  if (c < 28 && g < 8)
    for (int b = g; b < nblocks; b += 8) v += partials[b * 28 + c];
  if (g < 8) part28[g][c] = v;
  __syncthreads();
  if (threadIdx.x < 28) {
    double s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += part28[k][threadIdx.x];
    ssum[threadIdx.x] = s;
  }
  __syncthreads();
}
#endif

}  // namespace lio
