// batch_kernels.h — the map / feature stages of Estimator::SolveOptimization for B windows at once:
// every stage of BuildLocalMap (Estimator.cc:1361-1646), CalculateFeatures (:970-1097) and CalculateLaserOdom (:1242-1359) is
// ONE launch over all windows of the batch, indexed by blockIdx.y / .z through a per-window descriptor in device memory.
// The per-window arithmetic is the single-window kernels' (cloud_device.h), and every partition (blocks of a round, tiles of a
// filter) is a function of the window's own sizes: a window gives the same bits alone (B = 1) and inside any batch.
#pragma once
#include "cloud_kernels.h"
#include "seg_sort.h"

namespace lio {

#define LIO_BW_MAX_SEG 8        // segments of a local map: the pivot's cloud + the Wo - 1 frames behind it (Wo <= 7)
#define BW_KEY_NONE 0x7FFFFFFFu  // stored voxel key of the batched filter (absolute cells, 31 bits): all ones = no point
#define LIO_BW_MAX_STATIC 7     // frames whose features do not depend on the newest frame's rounds

struct BwSeg { const float4 *src; int n; int dst_off; int identity; int set_intensity; float intensity; Affine3f tf; };

struct BatchWin {
  // ---- BuildLocalMap: concat into local_all[loc_off, loc_off + n_local), keys, filter
  BwSeg seg[LIO_BW_MAX_SEG];
  int nseg, n_local;
  int loc_off, loc_cap;          // this window's range of the batch's point arrays (both multiples of 256)
  float inv_leaf;
  // ---- feature slots: this window's range of the batch's slot arrays (valid / coef / score)
  int slot_base, n_slots;
  FeatFrame fr[LIO_BW_MAX_STATIC];   // slot_off is an offset into the batch's arrays, tf_index an index into tf
  int nstatic;
  FeatFrame newest;              // the frame CalculateLaserOdom iterates on (M = 0: none)
  int nb_round, part_off, keep;  // its search blocks per round, its first row of the batch's partials, keep_features
  float tf[LIO_BW_MAX_STATIC + 1][8];   // local transforms (qx qy qz qw px py pz pad); the last entry is the newest frame's start
  float min_match_sq_dis, min_plane_dis;
};

// what the filter reports per window (read by the host after the stream has drained)
struct BwVoxOut { int count; VoxParams params; int range_overflow; };

// the K-NN grid of a window (host-computed from the filter's bounds, uploaded before the cell build)
struct BatchGrid { GridDesc g; int cell_off; int n_filtered; };

// Execution choices of a batch that the results do NOT depend on (every one of them keeps the sums' orders): 0 (occupancy: -1) = by the
// size of the launch.  They live in the batch handle (lio_est_batch_set_option); the environment variables named here only set the
// defaults of a new batch, read once at lio_est_batch_create.
struct BatchKnobs {
  int lanes_per_query = 0;   // 1 / 2 / 4 / 8 lanes per query of the search kernels (LIO_BW_LPQ); by size: 1 from 100 k queries, 4 from 15 k, else 8
  int occupancy = -1;        // 0 / 6 / 8 waves per SIMD of the one-lane-per-query kernels (LIO_BW_OCC); by default features 8, rounds as compiled
  int loop_groups = 0;       // 1 .. 4 launch chains of the trust-region loop side by side (LIO_BW_GROUPS); by size: 4 from 32 windows, 2 from 256
  int aux_threads = 0;       // 64 / 128 / 256 threads per block of the aux row (LIO_BW_AUX_THREADS); by size: 64 from 128 windows per launch
  int aux_stream = 0;        // 1: the aux row on a side stream (LIO_BW_AUX_STREAM; measured slower)
  int finish_threads = 0;    // 1 .. 8 host threads of the write-back; by size: 4 from 128 windows
  int time_kernels = 0;      // 1: HIP events around every launch of the trust-region loop's three kernels on the stream they run on (measurement
                             //    runs only: the events serialise the host's enqueue; BatchClock::kernel_ms / kernel_launches)
};
BatchKnobs batch_knobs_from_env();

int bw_round_blocks(int M);   // search blocks of one round of a window's newest frame: 64 queries each, whatever the lanes per query

// concat + voxel keys (absolute cells) + per-block bounds; then, one block per window, the bounds folded into VoxParams and the key layout
// the segmented sort runs on (seg_sort.h).  keys: loc-array sized; partial: 8 floats per 256-point block
// max_bits: what the passes the caller is going to run can order, minus one (a window that needs more is flagged in range_overflow)
void launch_bw_concat_keys(const BatchWin *win, int B, int max_local, float4 *local_all, uint32_t *keys, float *partial, VoxParams *params, KeyLayout *layout,
                           int *range_overflow, int max_bits, hipStream_t s);
// sorted (relative) keys -> tile heads -> centroids, counts
void launch_bw_vox_finish(const BatchWin *win, int B, int max_cap, const float4 *local_all, const uint32_t *keys_sorted, const uint32_t *vals_sorted, int *tile_heads,
                          float4 *filtered_all, const VoxParams *params, int *range_overflow, BwVoxOut *out, hipStream_t s);
// feature flags cleared, the newest frames' Gauss-Newton states started
void launch_bw_setup(const BatchWin *win, int B, int max_slots, uint8_t *valid_all, OdomState *odom, int *n_converged, hipStream_t s);
// K-NN grid: cell keys of the filtered points; (segmented sort by the caller); cell-sorted points + the dense table of run starts
void launch_bw_cell_keys(const BatchWin *win, const BatchGrid *grid, int B, int max_filtered, const float4 *filtered_all, uint32_t *keys, hipStream_t s);
void launch_bw_cell_table(const BatchWin *win, const BatchGrid *grid, int B, int max_filtered, const float4 *filtered_all, const uint32_t *keys_sorted,
                          const uint32_t *vals_sorted, int *cells_all, float4 *sorted_all, hipStream_t s);
// total_queries: stack points of the launch over all windows — picks the lanes per query (the results do not depend on it)
void launch_bw_features(const BatchWin *win, const BatchGrid *grid, int B, int max_M, int max_static, long long total_queries, const BatchKnobs &knobs,
                        const float4 *sorted_all, const int *cells_all, uint8_t *valid_all, float4 *coef_all, float *score_all, hipStream_t s);
void launch_bw_odom_round(const BatchWin *win, const BatchGrid *grid, int B, int max_nb, long long total_queries, const BatchKnobs &knobs, int round, OdomState *odom,
                          const float4 *sorted_all, const int *cells_all, uint8_t *valid_all, float4 *coef_all, float *score_all, double *partials,
                          int *n_converged, hipStream_t s);

}  // namespace lio
