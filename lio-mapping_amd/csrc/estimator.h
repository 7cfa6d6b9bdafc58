// estimator.h — the product's mirror of lio::Estimator's post-initialisation surface
// (include/imu_processor/Estimator.h:110-299): ProcessImu, ProcessLaserOdom (INITED branch),
// BuildLocalMap, SolveOptimization, SlideWindow.  Clouds live in HBM for the life of the window;
// the host keeps only the (W+1) x {P,R,V,Ba,Bg} states, the pre-integrations and the prior.
#pragma once
#include <condition_variable>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/lio_c.h"
#include "cloud_kernels.h"
#include "host_init.h"
#include "host_solver.h"
#include "batch_kernels.h"
#include "solve_step.h"

namespace lio {

typedef Rigid<float> Rigidf;
typedef Rigid<double> Rigidd;

struct EstConfig {
  int W = 15, Wo = 5;
  float corner_filter_size = 0.2f, surf_filter_size = 0.4f, min_match_sq_dis = 1.0f, min_plane_dis = 0.2f;
  Rigidf transform_lb;
  bool opt_extrinsic = false, imu_factor = true, point_distance_factor = false, prior_factor = false, marginalization_factor = true,
       enable_deskew = true, cutoff_deskew = false, keep_features = false;
  PimNoise pim;
  int max_num_iterations = 10;
  double max_solver_time = 0.10;
  int extrinsic_stage = 2;
  int init_window_factor = 3;
  // execution switches (lio_est_config's trailing block; environment overrides are applied in the constructor)
  bool device_solve = false, inline_marg = false, stream_sync = false;
  int moments_form = 0, resident_moments = 0;
};

struct DeviceCloud {
  DBuf<float4> buf;
  size_t n = 0;
  uint64_t id = 0;  // content id: equal ids in the same slot => identical content (restore() skips the copy)
  DeviceCloud() = default;
  // a moved-from cloud is EMPTY (n = 0, id = 0), not a null buffer with a stale count
  DeviceCloud(DeviceCloud &&o) noexcept : buf(std::move(o.buf)), n(o.n), id(o.id) { o.n = 0; o.id = 0; }
  DeviceCloud &operator=(DeviceCloud &&o) noexcept {
    if (this != &o) { buf = std::move(o.buf); n = o.n; id = o.id; o.n = 0; o.id = 0; }
    return *this;
  }
};

struct StampedPose { double time; Rigidf T; };

// HIP-event kernel timing on the estimator's stream (lio_est_enable_kernel_timing)
enum { KT_FEATURES = 0, KT_ODOM_FEATURES, KT_ODOM_ROWS, KT_ODOM_UPDATE, KT_MOMENTS, KT_VOXEL, KT_KNN_GRID, KT_CONCAT, KT_COUNT };
struct KernelTimers {
  bool on = false;
  int sample = 1;            // record every sample-th launch of each kind (events between kernels cost ~10 % when all are timed)
  int seen[KT_COUNT] = {0};
  struct Rec { hipEvent_t a, b; int id; double bytes; };
  struct Acc { int n = 0; double ms = 0, bytes = 0; };
  std::vector<Rec> pending;
  std::vector<hipEvent_t> pool;
  Acc acc[KT_COUNT];
  hipEvent_t get() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e; LIO_HIP(hipEventCreate(&e)); return e;
  }
  int begin(int id, double bytes, hipStream_t s) {
    if (!on) return -1;
    if ((seen[id]++ % sample) != 0) return -1;
    Rec r{get(), get(), id, bytes};
    LIO_HIP(hipEventRecord(r.a, s));
    pending.push_back(r);
    return int(pending.size()) - 1;
  }
  void end(int h, hipStream_t s) { if (h >= 0) LIO_HIP(hipEventRecord(pending[h].b, s)); }
  void resolve() {  // call after the stream has been synchronised
    for (Rec &r : pending) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { acc[r.id].n++; acc[r.id].ms += ms; acc[r.id].bytes += r.bytes; }
      pool.push_back(r.a); pool.push_back(r.b);
    }
    pending.clear();
  }
  void reset() { for (Acc &a : acc) a = Acc(); for (int &v : seen) v = 0; }
  ~KernelTimers() { for (hipEvent_t e : pool) (void)hipEventDestroy(e); for (Rec &r : pending) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); } }
};

// Marginalization is host-only work (one factor evaluation from the solve's final lidar moments, a Schur complement, two
// symmetric eigendecompositions: 0.12 ms) whose result — the prior — is first read by the NEXT solve, after that solve
// has built its local map and features (0.45 ms, mostly device time).  A persistent worker thread computes it meanwhile;
// every reader of the prior joins first, so the sequence of values is exactly the synchronous one.
class MargWorker {
 public:
  using Task = std::function<std::shared_ptr<MargPrior>()>;
  ~MargWorker() {
    { std::lock_guard<std::mutex> lk(mu_); quit_ = true; }
    cv_.notify_all();
    if (th_.joinable()) th_.join();
  }
  void submit(Task t) {
    std::unique_lock<std::mutex> lk(mu_);
    if (!th_.joinable()) th_ = std::thread([this] { run(); });
    cv_.wait(lk, [this] { return state_ == IDLE || state_ == DONE; });  // an unjoined (discarded) task finishes first
    task_ = std::move(t); state_ = PENDING; err_ = nullptr; result_.reset();
    lk.unlock();
    cv_.notify_all();
  }
  bool busy() { std::lock_guard<std::mutex> lk(mu_); return state_ != IDLE; }
  // blocks until the submitted task is done; returns false when nothing was submitted since the last join
  bool join(std::shared_ptr<MargPrior> &out) {
    std::unique_lock<std::mutex> lk(mu_);
    if (state_ == IDLE) return false;
    cv_.wait(lk, [this] { return state_ == DONE; });
    state_ = IDLE;
    if (err_) { std::exception_ptr e = err_; err_ = nullptr; std::rethrow_exception(e); }
    out = std::move(result_);
    return true;
  }

 private:
  enum { IDLE, PENDING, RUNNING, DONE };
  void run() {
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      cv_.wait(lk, [this] { return quit_ || state_ == PENDING; });
      if (quit_) return;
      Task t = std::move(task_);
      state_ = RUNNING;
      lk.unlock();
      std::shared_ptr<MargPrior> r;
      std::exception_ptr e;
      try { r = t(); } catch (...) { e = std::current_exception(); }
      lk.lock();
      result_ = std::move(r); err_ = e; state_ = DONE;
      cv_.notify_all();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_;
  std::thread th_;
  Task task_;
  std::shared_ptr<MargPrior> result_;
  std::exception_ptr err_;
  int state_ = IDLE;
  bool quit_ = false;
};

class Estimator {
 public:
  explicit Estimator(const EstConfig &cfg);
  ~Estimator();

  void ProcessImu(double dt, const V3d &acc, const V3d &gyr, double stamp);
  bool ProcessLaserOdom(const Rigidf &transform_in, const float *surf, size_t n_surf, const float *corner, size_t n_corner, double stamp,
                        lio_solve_report *rep);
  // surf_on_device: the surf cloud already lives in HBM (the scan-to-map stage's down-sampled stack before initialisation)
  bool ProcessLaserOdom(const Rigidf &transform_in, const float4 *surf, size_t n_surf, bool surf_on_device, double stamp, lio_solve_report *rep);
  bool PushFrame(const Rigidf &transform_in, const float *surf, size_t n_surf, const float *corner, size_t n_corner, double stamp,
                 bool surf_on_device = false);
  bool RunInitialization();
  void SetStatesFromLaser();
  // Estimator.cc:1648-2438.  With a hook installed (lio_est_config.device_solve: a batch of one window, est_batch.h) the whole
  // solve runs there; SolveOptimizationHost is the default path (device kernels + the trust-region loop on the host).
  bool SolveOptimization(lio_solve_report *rep) { return solve_hook_ ? solve_hook_(rep) : SolveOptimizationHost(rep); }
  bool SolveOptimizationHost(lio_solve_report *rep);
  std::function<bool(lio_solve_report *)> solve_hook_;
  void SlideWindow();
  void BuildLocalMap(lio_solve_report *rep);

  // ---- the per-window host halves of a BATCHED solve (est_batch.h drives them; the device work between them is one launch per
  // stage over all windows of the batch)
  bool BatchEligible() const;
  // host half of BuildLocalMap (Estimator.cc:1361-1646): segments of the local map, frames, local transforms, slot layout.
  // Offsets are the window's own (from 0); the batch shifts them to its arrays.
  void BatchDescribe(BatchWin &bw);
  // the problem of Estimator.cc:1660-1921 as the device loop reads it; *prior = the marginalization prior the problem uses.
  // false: this window's problem does not fit the device loop.
  bool BatchPackProblem(int bpf, DevProblem &pb, DevState &st, std::shared_ptr<MargPrior> *prior);
  // after the device loop: DoubleToVector, the report, convergence_flag_.  true: the window marginalises now — mg is filled
  // (linearisation point, layout) and *shell is the new prior without its matrices (they are computed on the device).
  bool BatchFinish(const DevState &st, const std::shared_ptr<MargPrior> &prior_used, lio_solve_report &R, DevMarg &mg, std::shared_ptr<MargPrior> *shell);
  void BatchSetOdom(const OdomState &st);   // CalculateLaserOdom's outcome as the rounds left it (Estimator.cc:1242-1359)

  // test hooks
  void SetWindow(const double *Ps, const double *Rs, const double *Vs, const double *Bas, const double *Bgs, const double g[3]);
  void SetSurfStack(int frame, const float *xyzi, size_t n);
  size_t GetSurfStack(int frame, float *out);
  void SetPreintegration(int frame, std::shared_ptr<Preintegration> p) { pre_integrations_[frame] = std::move(p); frames_dirty_ = true; }
  void BeginFrame(const V3d &acc, const V3d &gyr);
  size_t GetLocalMap(float *out);
  size_t GetFeatures(int frame, double *pt, double *co, double *sc);
  void Snapshot();
  bool Restore();
  // this handle's snapshot <- a copy of src's (same configuration; clouds copied into buffers of this handle): B windows of the
  // same data at distinct addresses for the batched measurements, without replaying the sequence B times
  bool CopySnapshotOf(Estimator &src);
  // B copies of the current window's lidar factors through ONE moments launch (distinct memory per copy); returns the average
  // launch-pair duration in ms and the algorithmic bytes per launch.  false when no features have been built yet.
  bool BenchBatchedMoments(int n_windows, int reps, double *avg_ms, double *bytes);

  EstConfig cfg_;
  int W_, Wo_;
  std::vector<V3d> Ps_, Vs_, Bas_, Bgs_;
  std::vector<M3d> Rs_;
  V3d g_vec_, acc_last_, gyr_last_;
  Rigidf transform_lb_, laser_odom_transform_;
  bool inited_ = false, first_imu_ = false, init_local_map_ = false, convergence_flag_ = false;
  int cir_buf_count_ = 0;
  // initialisation stage (Estimator.cc:430-618, 858-958)
  std::vector<LaserFrame> all_laser_transforms_;
  int n_state_ = 0;    // CircularBuffer size of Ps_/Rs_/Vs_/Bas_/Bgs_
  int n_frames_ = 0;   // CircularBuffer size of pre_integrations_/all_laser_transforms_/the stacks
  int laser_odom_recv_count_ = 0, extrinsic_stage_ = 2;
  double initial_time_ = -1;
  M3d R_WI_;
  enum { EV_SKIPPED = 0, EV_FILLING = 1, EV_INIT_FAILED = 2, EV_INITIALISED = 3, EV_SOLVED = 4 };
  int last_event_ = EV_SKIPPED;
  std::shared_ptr<MargPrior> last_marg_;   // read through JoinMarg() only: a marginalization may still be running
  MargWorker marg_worker_;
  bool async_marg_ = true;                  // LIO_ASYNC_MARG=0 computes it inside SolveOptimization
  unsigned marg_epoch_ = 0, marg_task_epoch_ = 0;   // Restore() bumps the epoch: a result computed for a discarded state is dropped
  // materialize: a prior that a batched solve left on the device (MargPrior::on_device) is brought to the host as well — every
  // reader of its matrices on the host needs that; the batch itself, which feeds the next solve from the device copy, does not
  void JoinMarg(bool materialize = true) {
    std::shared_ptr<MargPrior> r;
    const bool discard = marg_task_epoch_ != marg_epoch_;
    try {
      if (marg_worker_.join(r) && !discard) last_marg_ = std::move(r);
    } catch (...) {
      if (!discard) throw;  // a failure of a task whose result is dropped anyway (state restored meanwhile) is not the caller's problem
    }
    if (materialize && last_marg_) last_marg_->materialize();
  }
  std::vector<std::shared_ptr<Preintegration>> pre_integrations_;
  std::shared_ptr<Preintegration> tmp_pre_integration_;
  int laser_odom_iters_ = 0, laser_odom_kz_ = 0;
  double dbg_eval_ms_ = 0, dbg_sync_ms_ = 0; int dbg_eval_n_ = 0;  // LIO_DEBUG_TIMING: time inside LidarEval (launch + D2H + sync)
  KernelTimers timers_;
  // factor sharding across ranks (lio_est_set_factor_sharding)
  int shard_rank_ = 0, shard_world_ = 1;
  lio_allreduce_fn allreduce_ = nullptr;
  void *allreduce_user_ = nullptr;
  void *rccl_comm_ = nullptr;               // ncclComm_t (lio_est_set_factor_sharding_rccl): the all-reduce runs on stream_, in HBM
  bool Sharded() const { return shard_world_ > 1 && (allreduce_ || rccl_comm_); }

 private:
  friend class EstimatorBatch;
  struct HostState;  // snapshot payload
  Rigidd LidarPose(int i, const Rigidd &lb) const;
  Rigidf RelTransform(int i, const Rigidd &T_pivot, const Rigidd &lb) const;
  void VectorToParams(WindowParams &P) const;
  void ParamsToVector(const WindowParams &P);
  void FillMomentArgs(MomentArgs &ma, int &max_slots) const;
  void LidarEval(const WindowParams &P, std::vector<FrameMoments> &m);
  void LidarLaunch(const WindowParams &P);             // asynchronous part: frame transforms + moments kernels
  void LidarWait(std::vector<FrameMoments> &m);        // stream sync (+ all-reduce when sharded) + unpack
  bool LidarWaitFrame(int i, FrameMoments &fm);       // per-frame form (resident kernel only; false otherwise)
  void PushCloud(DeviceCloud &&c, size_t n, int n_before);
  void PushState(int from);

  hipStream_t stream_ = nullptr, stream2_ = nullptr;
  bool owns_stream_ = true;   // false once a batch has adopted the handle: its work is enqueued on the batch's stream
  void AdoptStream(hipStream_t s);
  void ReleaseAdoptedStream();   // the batch is gone: the handle works on a stream of its own again
  void FusePivotOnce();       // A.15: frames 0 .. pivot fused into the pivot's stack, the first time a local map is built
  hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr;
  std::vector<DeviceCloud> stacks_;
  std::vector<size_t> size_surf_stack_;
  std::vector<StampedPose> imu_stamped_;
  DeviceCloud local_, local_filtered_, scratch_cloud_, upload_;
  VoxelGridDev vox_;
  KnnGrid grid_;
  // feature slots
  DBuf<uint8_t> f_valid_;
  DBuf<float4> f_coef_;
  DBuf<float> f_score_;
  std::vector<int> slot_off_, nslots_;
  size_t total_slots_ = 0;
  DBuf<float> d_transforms_;
  DBuf<OdomState> d_odom_;
  DBuf<double> d_odom_partials_, d_moment_partials_, d_moment_out_;
  int moments_form_ = 0;            // 0 by launch size, 1 MFMA, 2 VALU
  // Resident moments (solve_kernels.h): one launch per solve; every linearisation is a doorbell write + a spin on
  // the blocks' completion words.  Begun lazily by the first LidarLaunch of a SolveOptimization, stopped when it returns.
  bool resident_moments_ = true;    // configured (lio_est_config.resident_moments / LIO_RESIDENT_MOMENTS)
  bool resident_never_ = false;     // resident_moments = 3: the resident form's partition, launch pairs only (what a refused solve gets)
  int res_per_lane_ = 0;            // residuals a lane keeps in registers: 0 = chosen per window (ResidentBpf), LIO_RES_PER_LANE forces 1, 2, 4, 8
  int res_lanes_ = 4;               // ... of the launch in flight
  bool res_allowed_ = false;        // inside SolveOptimization
  bool res_active_ = false;         // a resident kernel is waiting on the doorbell
  int res_bpf_ = 0, res_nframes_ = 0;
  unsigned res_seq_ = 0;            // sequence number of the last pass rung (monotonic over the life of the handle)
  int res_relaunches_ = 0;          // launches that replaced an expired one within this solve (bounded: ResidentAwaitWord)
  unsigned res_launch_seq_ = 0;     // first sequence number of the launch in flight (its STOP value is derived from it)
  double *h_res_door_ = nullptr, *h_res_out_ = nullptr;    // coherent pinned host memory: doorbell, per-frame folded records
  unsigned *h_res_words_ = nullptr;
  long long res_timeout_ticks_ = 0;
  double res_tick_us_ = 0.01;       // microseconds per wall-clock tick
  double res_busy_us_ = 0, res_bytes_ = 0; int res_passes_ = 0, res_passes_total_ = 0;   // device-side busy time of the passes (doorbell copy seen -> sums posted), SURVEY 8(d) bytes
  MomentArgs res_args_{};
  DBuf<double> d_res_relay_, d_res_part_;   // HBM: the doorbell as republished by the relay block; the per-block records
  // lio_est_enable_kernel_timing(-1): HIP events around every launch of the resident kernel (it stays in use, unlike under
  // the per-kernel timing of on >= 1): its dispatch-to-exit span, which is what rocprofv3 reports for it
  bool res_time_launch_ = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> res_launch_events_;
  double res_launch_ms_ = 0; int res_launches_ = 0;
 public:
  void ResidentLaunchTiming(bool on) { res_time_launch_ = on; }
  int ResidentLaunchStats(double *total_ms);
 private:
  double res_diag_us_[4] = {0, 0, 0, 0}, res_polls_ = 0, res_relay_us_ = 0, res_ring_to_done_ms_ = 0, res_t_ring_ = 0, res_echo_ms_ = 0;
  int ResidentBpf(int max_slots, int nframes, int *per_lane = nullptr) const;
  bool ResidentBegin(const MomentArgs &ma);
  void ResidentRing(const MomentArgs &ma);
  void ResidentWait(std::vector<FrameMoments> &m);
  void ResidentWaitFrame(int f, FrameMoments &fm);   // frame f (0-based) of the pass in flight, as soon as its word is in
  void ResidentAwaitWord(int f);
  void ResidentUnpackFrame(int f, FrameMoments &fm);
  void ResidentPassDone();
  void ResidentLaunchKernel(unsigned first_seq);
 public:
  void ResidentEnd();
  double ResidentBusyUs(int *passes, double *bytes) const { if (passes) *passes = res_passes_total_; if (bytes) *bytes = res_bytes_; return res_busy_us_; }
 private:
  double *h_moment_out_ = nullptr;  // pinned
  OdomState *h_odom_ = nullptr;     // pinned landing zone of the laser-odom state peeks
  // Completion words (dev.h: HostSignal) in coherent pinned memory: [0, 96) one per block of k_moment_reduce, [128] the
  // newest-frame round.  The host spins on them instead of hipStreamSynchronize (LIO_HOST_SIGNAL=0 restores the synchronize calls).
  unsigned *h_signal_ = nullptr;
  unsigned signal_seq_[2] = {0, 0};
  bool host_signal_ = true;
  int device_id_ = 0;
  HostSignal moment_signal_{};
  // Restore() re-copies the per-frame containers (pre-integrations, laser transforms, stamped poses, the running pre-integration)
  // only if something touched them since they last equalled the snapshot's: a solve does not, and at hundreds of windows per
  // batch step their copies were the host's largest share of a restore + solve step
  bool frames_dirty_ = true;
  std::unique_ptr<HostState> snap_;
  std::vector<DeviceCloud> snap_stacks_;
};

}  // namespace lio
