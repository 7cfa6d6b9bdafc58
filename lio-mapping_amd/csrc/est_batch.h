// est_batch.h — Estimator::SolveOptimization (Estimator.cc:1648-2438) for B windows at once: SURVEY.md 8(d)(ii)'s batched form.
//
// A batch adopts B estimator handles (independent windows: different vehicles, different logs, or the shards of an offline
// re-optimisation).  One call solves all of them, every stage ONE launch over all windows:
//
//   concat + voxel keys | segmented sort | tile heads | centroids     BuildLocalMap        Estimator.cc:1480-1519
//   cell keys | segmented sort | cell-sorted points + run starts      KdTreeFLANN build    :1544-1545
//   features of the frames behind the pivot                           CalculateFeatures    :970-1097
//   <= 10 x (search + fit + rows | fold + 6x6 step)                   CalculateLaserOdom   :1242-1359
//   <= max_iterations + 1 x (moments + IMU / prior / lidar-map aux row | trust-region step, one workgroup per window)
//                                                                     ceres::Solve         :1909-1990
//   aux row at the marginalization point | Schur complement + eigensolves on fp64 MFMA, one workgroup per window
//                                                                     Marginalize          MarginalizationFactor.cc:185-311
//
// The host touches a window three times per solve (descriptors, the K-NN grid's dimensions from the filter's bounds, the write-back
// of the state) and waits for the device three times per BATCH.  The marginalization's result — the next solve's prior — stays on
// the device; a host-side reader (lio_est_get_prior, a snapshot, the single-window solver) fetches it on demand.
// Per-window arithmetic does not depend on the batch: a window gives the same bits alone (B = 1, which is what
// lio_est_config.device_solve selects for a single handle) and inside any batch.  Windows whose problem the device loop does not
// take (not initialised yet, the convergence_flag_ logic changing the problem's shape, factor sharding) are solved by the
// single-window path inside the same call.
#pragma once
#include <memory>
#include <vector>

#include "batch_kernels.h"
#include "estimator.h"
#include "marg_kernels.h"

namespace lio {

struct BatchClock {   // host wall clock of the last Solve(), ms
  double describe = 0, map = 0, grid_features = 0, pack = 0, solve = 0, finish = 0, fallback = 0, total = 0;
  int n_device = 0, n_host = 0, rounds = 0, iterations = 0;
  // device time of the last Solve()'s stages (HIP events on the batch's stream), ms: filter chain (concat, keys, sort, heads,
  // centroids), K-NN grids, features of the older frames, newest-frame rounds, trust-region loop, marginalization
  double dev[6] = {0, 0, 0, 0, 0, 0};
  // with the option time_kernels: summed duration and count of the last Solve()'s launches of k_bw_aux, k_bw_moments, k_bw_solve_step
  double kernel_ms[3] = {0, 0, 0};
  int kernel_launches[3] = {0, 0, 0};
  double dev_marg_wait = 0;   // of dev[3]: the stream's wait for the PREVIOUS solve's marginalization (it sits in front of the problems' upload)
};

class EstimatorBatch {
 public:
  explicit EstimatorBatch(const std::vector<Estimator *> &members);
  ~EstimatorBatch();
  EstimatorBatch(const EstimatorBatch &) = delete;
  EstimatorBatch &operator=(const EstimatorBatch &) = delete;
  int size() const { return int(m_.size()); }
  // SolveOptimization of every member; reps: size() reports, or null.  Returns the number of windows that were solved.
  int Solve(lio_solve_report *reps);
  // waits for everything the batch has enqueued (the marginalizations of the last Solve)
  void Sync();
  const BatchClock &clock();   // waits for the batch's stream (the device stage times come from events on it)
  // Test hook (lio_est_batch_stage_digest): a 64-bit digest per window of what stage `stage` of the LAST Solve() left on the device —
  // 0 filtered local map (in order), 1 K-NN grid (cell table relative to the window's base + the cell-sorted points as a multiset per
  // cell run: the placement order inside a cell is not defined), 2 feature flags, 3 plane coefficients of the set flags, 4 the newest
  // frame's Gauss-Newton state, 5 the trust-region loop's final state, 6 the final moments partials.  Waits for the batch.
  void StageDigest(int stage, unsigned long long *out);
  hipStream_t stream() const { return stream_; }
  // execution choices (batch_kernels.h: BatchKnobs) by name: lanes_per_query, occupancy, loop_groups, aux_threads, aux_stream, finish_threads;
  // false: unknown name or a value the knob does not take
  bool SetOption(const char *name, int value);
  const BatchKnobs &knobs() const { return knobs_; }
  bool window_ok(int w) const { return w >= 0 && w < int(ok_.size()) && ok_[size_t(w)] != 0; }   // the last Solve() solved window w

 private:
  struct Slab {   // offsets (doubles) into a window's scratch slab
    size_t prior[2], imu, lmap, prior_out, exprior, Hcur, Sbuf, prof, marg_imu, marg_lmap, marg_prior_out, marg_A, marg_info, total;
  };
  struct Win {
    Estimator *e = nullptr;
    bool device = false;                       // on the device path in the Solve in progress
    std::shared_ptr<MargPrior> prior_used;     // the prior of the problem in flight
    std::shared_ptr<MargPrior> dev_prior[2];   // which host object each of the two device prior buffers holds
    int cur = 0;                               // buffer holding prior_used
    int bpf = 1, max_slots = 0;
    int key_bits = 0;                          // bits the window's relative voxel keys took in its last solve (0: not known yet)
    size_t part_off = 0;                       // doubles into partials_
  };
  void FetchPrior(int w, int buf, MargPrior &pr);
  void Materialize(int w, int buf);
  std::vector<Estimator *> m_;
  std::vector<Win> win_;
  hipStream_t stream_ = nullptr;
  // The trust-region loop runs in up to kGroups groups of windows, each a launch chain on a stream of its own: a group's step
  // kernel (one workgroup per window: a quarter of the chip at 64 windows) overlaps the other groups' moments passes.  The
  // marginalization runs on a stream of its own behind the loop and is joined before the next solve's problems go up: it
  // overlaps the next solve's filter, features and rounds.
  static constexpr int kGroups = 4;
  // (LIO_BW_AUX_STREAM=1, an experiment kept for re-measurement: the aux row of an iteration on a side stream beside the moments pass.)
  hipStream_t stream_grp_[kGroups] = {}, stream_aux_[kGroups] = {}, stream_marg_ = nullptr;
  hipEvent_t ev_fork_ = nullptr, ev_grp_[kGroups] = {}, ev_aux_[kGroups] = {},
             ev_step_[kGroups] = {}, ev_marg_ = nullptr;
  bool marg_in_flight_ = false;
  BatchKnobs knobs_;
  std::vector<char> ok_;
  int device_id_ = 0;
  hipEvent_t ev_[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  std::vector<hipEvent_t> ev_k_;        // time_kernels: 4 events per iteration and group (aux | moments | step boundaries)
  int ev_k_used_ = 0;
  hipEvent_t ev_wait_[2] = {nullptr, nullptr};   // around the wait for the previous marginalization
  bool ev_wait_valid_ = false;
  bool ev_valid_ = false;
  Slab lay_{};
  BatchClock clk_;
  // pinned staging (one entry per window)
  BatchWin *h_win_ = nullptr; BatchGrid *h_grid_ = nullptr; BwVoxOut *h_vout_ = nullptr; OdomState *h_odom_ = nullptr;
  BatchSolve *h_bs_ = nullptr; DevProblem *h_pb_ = nullptr; DevState *h_st_ = nullptr; DevMarg *h_mg_ = nullptr;
  double *h_prior_ = nullptr;   // one ds_prior_mats_size(MARG_MAX_N) slot per window: priors on their way to the device
  int *h_nconv_ = nullptr;
  // device
  DBuf<BatchWin> d_win_; DBuf<BatchGrid> d_grid_; DBuf<BwVoxOut> d_vout_; DBuf<OdomState> d_odom_;
  DBuf<BatchSolve> d_bs_; DBuf<DevProblem> d_pb_; DBuf<DevState> d_st_; DBuf<DevMarg> d_mg_;
  DBuf<double> slab_, partials_, odom_partials_;
  DBuf<float4> local_all_, filtered_all_, sorted_all_, coef_all_;
  DBuf<uint32_t> keys_, keysb_, vals_, valsb_, ckeys_, sort_hist_;   // the segmented sort's ping-pong pairs (filter, then K-NN grid) and its histograms
  DBuf<SegDesc> d_seg_;
  DBuf<KeyLayout> d_layout_;
  SegDesc *h_seg_ = nullptr;
  DBuf<float> bounds_partial_, score_all_;
  DBuf<int> tile_heads_, range_overflow_, cells_all_, nconv_;
  DBuf<VoxParams> vparams_;
  DBuf<uint8_t> valid_all_;
};

}  // namespace lio
