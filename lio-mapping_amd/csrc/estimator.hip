// estimator.hip — host orchestration of the product's sliding-window step (see estimator.h).
// Reference call structure: Estimator.cc:430-774 (ProcessLaserOdom), :1361-1646 (BuildLocalMap),
// :1648-2438 (SolveOptimization), :2440-2568 (VectorToDouble/DoubleToVector), :2570-2666 (SlideWindow).
#include "estimator.h"
#include "marg_kernels.h"
#include "rccl_comm.h"

#include <atomic>
#include <cfloat>
#include <climits>
#include <chrono>
#include <cstring>

namespace lio {

static const bool g_debug_timing = std::getenv("LIO_DEBUG_TIMING") != nullptr;   // read once: the solve is a hot path
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// math_utils.h:186-232 (degrees)
static V3d R2ypr(const M3d &R) {
  V3d n(R(0, 0), R(1, 0), R(2, 0)), o(R(0, 1), R(1, 1), R(2, 1)), a(R(0, 2), R(1, 2), R(2, 2));
  double y = atan2(n.y, n.x);
  double p = atan2(-n.z, n.x * cos(y) + n.y * sin(y));
  double r = atan2(a.x * sin(y) - a.y * cos(y), -o.x * sin(y) + o.y * cos(y));
  return V3d(y, p, r) / M_PI * 180.0;
}
static M3d ypr2R(const V3d &ypr) {
  double y = ypr.x / 180.0 * M_PI, p = ypr.y / 180.0 * M_PI, r = ypr.z / 180.0 * M_PI;
  M3d Rz, Ry, Rx;
  Rz(0, 0) = cos(y); Rz(0, 1) = -sin(y); Rz(1, 0) = sin(y); Rz(1, 1) = cos(y); Rz(2, 2) = 1;
  Ry(0, 0) = cos(p); Ry(0, 2) = sin(p); Ry(1, 1) = 1; Ry(2, 0) = -sin(p); Ry(2, 2) = cos(p);
  Rx(0, 0) = 1; Rx(1, 1) = cos(r); Rx(1, 2) = -sin(r); Rx(2, 1) = sin(r); Rx(2, 2) = cos(r);
  return Rz * Ry * Rx;
}
template <typename T, typename U> static Quat<U> qcast(const Quat<T> &q) { return Quat<U>(U(q.w), U(q.x), U(q.y), U(q.z)); }
template <typename T, typename U> static Vec3<U> vcast(const Vec3<T> &v) { return Vec3<U>(U(v.x), U(v.y), U(v.z)); }
static Rigidd toDouble(const Rigidf &t) { return Rigidd(qcast<float, double>(t.rot), vcast<float, double>(t.pos)); }
static Rigidf toFloat(const Rigidd &t) { return Rigidf(qcast<double, float>(t.rot), vcast<double, float>(t.pos)); }
static Affine3f affineOf(const Rigidf &tf) {
  Mat3<float> R = linearOf(tf);
  Affine3f a;
  for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) a.m[4 * r + c] = R(r, c); }
  a.m[3] = tf.pos.x; a.m[7] = tf.pos.y; a.m[11] = tf.pos.z;
  return a;
}

struct Estimator::HostState {
  std::vector<V3d> Ps, Vs, Bas, Bgs;
  std::vector<M3d> Rs;
  V3d g_vec, acc_last, gyr_last;
  Rigidf transform_lb;
  bool inited, first_imu, init_local_map, convergence_flag;
  int cir_buf_count;
  std::vector<LaserFrame> all_laser_transforms;
  int n_state, n_frames, laser_odom_recv_count, extrinsic_stage, last_event;
  double initial_time;
  M3d R_WI;
  std::shared_ptr<MargPrior> last_marg;
  std::vector<std::shared_ptr<Preintegration>> pre_integrations;
  std::shared_ptr<Preintegration> tmp_pre_integration;
  std::vector<size_t> size_surf_stack;
  std::vector<StampedPose> imu_stamped;
};

Estimator::Estimator(const EstConfig &cfg) : cfg_(cfg), W_(cfg.W), Wo_(cfg.Wo) {
  int ndev = 0;
  LIO_HIP(hipGetDeviceCount(&ndev));
  if (ndev <= 0) throw DeviceError("no HIP device: the product has no CPU path");
  LIO_HIP(hipStreamCreate(&stream_));
  LIO_HIP(hipStreamCreate(&stream2_));
  LIO_HIP(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
  LIO_HIP(hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming));
  transform_lb_ = cfg.transform_lb;
  Ps_.assign(W_ + 1, V3d()); Vs_ = Bas_ = Bgs_ = Ps_;
  Rs_.assign(W_ + 1, M3d::identity());
  pre_integrations_.assign(W_ + 1, nullptr);
  stacks_.resize(W_ + 1);
  size_surf_stack_.assign(W_ + 1, 0);
  slot_off_.assign(W_ + 1, 0); nslots_.assign(W_ + 1, 0);
  g_vec_ = V3d(0, 0, -cfg.pim.g_norm);
  all_laser_transforms_.assign(W_ + 1, LaserFrame());
  extrinsic_stage_ = cfg.extrinsic_stage;
  R_WI_ = M3d::identity();
  // ClearState (Estimator.cc:234-288): the running pre-integration exists before the first IMU sample
  tmp_pre_integration_ = std::make_shared<Preintegration>(acc_last_, gyr_last_, Bas_[0], Bgs_[0], cfg_.pim);
  d_odom_.reserve(1);
  d_moment_out_.reserve(size_t(LIO_MAX_FRAMES) * LIO_MOMENT_OUT);
  LIO_HIP(hipMemset(d_moment_out_.p, 0, sizeof(double) * LIO_MAX_FRAMES * LIO_MOMENT_OUT));   // the two pad entries per frame stay zero under the all-reduce
  LIO_HIP(hipDeviceSynchronize());   // the memsets above run on the null stream; the kernels that read them on streams of our own
  // Execution switches: lio_est_config's trailing block, each overridable by its environment variable (A/B runs of a built host).
  async_marg_ = !cfg.inline_marg;
  host_signal_ = !cfg.stream_sync; moments_form_ = cfg.moments_form;
  resident_moments_ = cfg.resident_moments != 2;
  resident_never_ = cfg.resident_moments == 3;
  if (const char *e = std::getenv("LIO_MOMENTS")) moments_form_ = std::string(e) == "mfma" ? 1 : (std::string(e) == "valu" ? 2 : moments_form_);
  if (const char *e = std::getenv("LIO_RESIDENT_MOMENTS")) resident_moments_ = std::atoi(e) != 0;
  d_res_relay_.reserve(size_t(LIO_MAX_FRAMES) * LIO_RES_DOOR);
  LIO_HIP(hipMemset(d_res_relay_.p, 0, sizeof(double) * LIO_MAX_FRAMES * LIO_RES_DOOR));
  d_res_part_.reserve(size_t(LIO_RES_MAX_BLOCKS) * LIO_MOMENT_OUT);
  LIO_HIP(hipMemset(d_res_part_.p, 0, sizeof(double) * LIO_RES_MAX_BLOCKS * LIO_MOMENT_OUT));   // flags: no pass has sequence number 0
  if (const char *e = std::getenv("LIO_RES_PER_LANE")) { const int v = std::atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8) res_per_lane_ = v; }
  LIO_HIP(hipHostMalloc(reinterpret_cast<void **>(&h_res_door_), sizeof(double) * LIO_MAX_FRAMES * LIO_RES_DOOR, hipHostMallocCoherent));
  LIO_HIP(hipHostMalloc(reinterpret_cast<void **>(&h_res_out_), sizeof(double) * LIO_MAX_FRAMES * LIO_RES_OUT, hipHostMallocCoherent));
  LIO_HIP(hipHostMalloc(reinterpret_cast<void **>(&h_res_words_), sizeof(unsigned) * (LIO_MAX_FRAMES + 2), hipHostMallocCoherent));   // + the relay block's word + its echo
  std::memset(h_res_door_, 0, sizeof(double) * LIO_MAX_FRAMES * LIO_RES_DOOR);
  std::memset(h_res_words_, 0, sizeof(unsigned) * (LIO_MAX_FRAMES + 2));
  std::memset(h_res_out_, 0, sizeof(double) * LIO_MAX_FRAMES * LIO_RES_OUT);   // the diagnostic slots are read whether or not the kernel fills them
  {
    int khz = 0, dev = 0;
    LIO_HIP(hipGetDevice(&dev));
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;   // 100 MHz on gfx9
    res_tick_us_ = 1e3 / double(khz);
    res_timeout_ticks_ = (long long)(0.2 * 1e3 * khz);   // 200 ms without a doorbell: the block posts LIO_RES_EXPIRED and exits
  }
  if (const char *e = std::getenv("LIO_ASYNC_MARG")) async_marg_ = std::atoi(e) != 0;
  // coherent (fine-grained): kernels store results and completion words here and the host reads them while the stream is live
  LIO_HIP(hipHostMalloc(reinterpret_cast<void **>(&h_moment_out_), sizeof(double) * LIO_MAX_FRAMES * LIO_MOMENT_OUT, hipHostMallocCoherent));
  LIO_HIP(hipHostMalloc(reinterpret_cast<void **>(&h_odom_), sizeof(OdomState), hipHostMallocCoherent));
  LIO_HIP(hipHostMalloc(reinterpret_cast<void **>(&h_signal_), 256 * sizeof(unsigned), hipHostMallocCoherent));
  std::memset(h_signal_, 0, 256 * sizeof(unsigned));
  if (const char *e = std::getenv("LIO_HOST_SIGNAL")) host_signal_ = std::atoi(e) != 0;
  vox_.set_host_signal(host_signal_);
  LIO_HIP(hipGetDevice(&device_id_));
}

Estimator::~Estimator() {
  try { ResidentEnd(); if (stream_) (void)hipStreamSynchronize(stream_); } catch (...) {}
  try { JoinMarg(); } catch (...) {}
  if (h_res_door_) (void)hipHostFree(h_res_door_);
  if (h_res_out_) (void)hipHostFree(h_res_out_);
  if (h_res_words_) (void)hipHostFree(h_res_words_);
  if (h_moment_out_) (void)hipHostFree(h_moment_out_);
  if (h_signal_) (void)hipHostFree(h_signal_);
  if (h_odom_) (void)hipHostFree(h_odom_);
  if (ev_fork_) (void)hipEventDestroy(ev_fork_);
  if (ev_join_) (void)hipEventDestroy(ev_join_);
  if (stream2_) (void)hipStreamDestroy(stream2_);
  if (stream_ && owns_stream_) (void)hipStreamDestroy(stream_);
}

// A batch (est_batch.h) takes over the handle's stream: everything the handle enqueues from now on (Restore's copies, SlideWindow's
// concat, PushFrame's filter) is ordered with the batch's own launches without an event per window and solve.
void Estimator::AdoptStream(hipStream_t s) {
  ResidentEnd();
  LIO_HIP(hipStreamSynchronize(stream_));
  LIO_HIP(hipStreamSynchronize(stream2_));
  if (owns_stream_) LIO_HIP(hipStreamDestroy(stream_));
  stream_ = s; owns_stream_ = false;
}
void Estimator::ReleaseAdoptedStream() {
  if (owns_stream_) return;
  stream_ = nullptr;
  LIO_HIP(hipStreamCreate(&stream_));
  owns_stream_ = true;
}

template <typename T> static void push_full(std::vector<T> &buf, T v) {
  for (size_t i = 0; i + 1 < buf.size(); ++i) buf[i] = std::move(buf[i + 1]);
  buf.back() = std::move(v);
}
// CircularBuffer::push (include/utils/CircularBuffer.h:164-172) with `size` elements held
template <typename T> static void push_at(std::vector<T> &buf, int size, T v) {
  if (size < int(buf.size())) buf[size] = std::move(v); else push_full(buf, std::move(v));
}

void Estimator::PushState(int from) {  // Ps_.push(Ps_[from]) ... (Estimator.cc:2646-2651)
  const V3d p = Ps_[from], v = Vs_[from], ba = Bas_[from], bg = Bgs_[from];
  const M3d r = Rs_[from];
  push_at(Ps_, n_state_, p); push_at(Vs_, n_state_, v); push_at(Rs_, n_state_, r); push_at(Bas_, n_state_, ba); push_at(Bgs_, n_state_, bg);
  if (n_state_ < W_ + 1) ++n_state_;
}

void Estimator::ProcessImu(double dt, const V3d &acc, const V3d &gyr, double stamp) {
  frames_dirty_ = true;
  if (!first_imu_) {
    first_imu_ = true; acc_last_ = acc; gyr_last_ = gyr;
    if (n_state_ == 0) n_state_ = 1;  // the zero state pushed at :347-354 (the buffers already hold it)
  }
  if (cir_buf_count_ != 0) {
    if (tmp_pre_integration_) tmp_pre_integration_->push_back(dt, acc, gyr);
    const int j = cir_buf_count_;
    V3d un_acc_0 = Rs_[j] * (acc_last_ - Bas_[j]) + g_vec_;
    V3d un_gyr = 0.5 * (gyr_last_ + gyr) - Bgs_[j];
    Rs_[j] = Rs_[j] * toRot(deltaQ(un_gyr * dt));
    V3d un_acc_1 = Rs_[j] * (acc - Bas_[j]) + g_vec_;
    V3d un_acc = 0.5 * (un_acc_0 + un_acc_1);
    Ps_[j] = Ps_[j] + (dt * Vs_[j] + 0.5 * dt * dt * un_acc);
    Vs_[j] = Vs_[j] + dt * un_acc;
    StampedPose tt;
    tt.time = stamp;
    tt.T.pos = vcast<double, float>(Ps_[j]);
    Mat3<float> Rf;
    for (int k = 0; k < 9; ++k) Rf.m[k] = float(Rs_[j].m[k]);
    tt.T.rot = fromRot(Rf);
    if (imu_stamped_.size() >= 100) imu_stamped_.erase(imu_stamped_.begin());
    imu_stamped_.push_back(tt);
  }
  acc_last_ = acc; gyr_last_ = gyr;
}

void Estimator::BeginFrame(const V3d &acc, const V3d &gyr) {
  frames_dirty_ = true;
  acc_last_ = acc; gyr_last_ = gyr; first_imu_ = true;
  tmp_pre_integration_ = std::make_shared<Preintegration>(acc_last_, gyr_last_, Bas_[cir_buf_count_], Bgs_[cir_buf_count_], cfg_.pim);
}

void Estimator::SetWindow(const double *Ps, const double *Rs, const double *Vs, const double *Bas, const double *Bgs, const double g[3]) {
  for (int i = 0; i <= W_; ++i) {
    Ps_[i] = V3d(Ps[3 * i], Ps[3 * i + 1], Ps[3 * i + 2]); Vs_[i] = V3d(Vs[3 * i], Vs[3 * i + 1], Vs[3 * i + 2]);
    Bas_[i] = V3d(Bas[3 * i], Bas[3 * i + 1], Bas[3 * i + 2]); Bgs_[i] = V3d(Bgs[3 * i], Bgs[3 * i + 1], Bgs[3 * i + 2]);
    for (int k = 0; k < 9; ++k) Rs_[i].m[k] = Rs[9 * i + k];
  }
  g_vec_ = V3d(g[0], g[1], g[2]);
  inited_ = true; first_imu_ = true; cir_buf_count_ = W_;
  n_state_ = n_frames_ = W_ + 1;
}

static std::atomic<uint64_t> g_content_id{1};
// Resident kernels hold their CUs until the host (or a peer block) feeds them, so the blocks of ALL of them must be co-resident:
// a process that drives many windows admits only as many as fit (four moments kernels of ~100 blocks);
// a solve that is not admitted takes the launch path, with the same results.
static std::atomic<int> g_resident_moments{0};
// Solves in flight in this process.  A resident moments kernel holds ~100 CUs' worth of registers while it waits for the host,
// which is free when the GPU has nothing else to do and expensive when other windows' feature kernels want those CUs: measured
// on the MI355X with four windows solving on four host threads, 3290 solves/s with every solve resident, 3530 with one at a time,
// 4230 with none (launch pairs).  So a solve takes the resident form only while it is the ONLY solve in flight.
static std::atomic<int> g_active_solves{0};
// (process-wide, hence an environment knob and not a lio_est_config field; 0 = every solve takes the launch path)
static const int kMaxResidentMoments = [] { const char *e = std::getenv("LIO_MAX_RESIDENT_MOMENTS"); return e ? std::max(0, std::atoi(e)) : 4; }();

void Estimator::SetSurfStack(int frame, const float *xyzi, size_t n) {
  DeviceCloud &c = stacks_[frame];
  c.id = ++g_content_id;
  c.buf.reserve(std::max<size_t>(n, 1));
  if (n) LIO_HIP(hipMemcpyAsync(c.buf.p, xyzi, n * sizeof(float4), hipMemcpyHostToDevice, stream_));
  LIO_HIP(hipStreamSynchronize(stream_));
  c.n = n;
  size_surf_stack_[frame] = n;
  frames_dirty_ = true;
}
size_t Estimator::GetSurfStack(int frame, float *out) {
  const DeviceCloud &c = stacks_[frame];
  if (out && c.n) { LIO_HIP(hipMemcpyAsync(out, c.buf.p, c.n * sizeof(float4), hipMemcpyDeviceToHost, stream_)); LIO_HIP(hipStreamSynchronize(stream_)); }
  return c.n;
}
size_t Estimator::GetLocalMap(float *out) {
  if (out && local_filtered_.n) {
    LIO_HIP(hipMemcpyAsync(out, local_filtered_.buf.p, local_filtered_.n * sizeof(float4), hipMemcpyDeviceToHost, stream_));
    LIO_HIP(hipStreamSynchronize(stream_));
  }
  return local_filtered_.n;
}

size_t Estimator::GetFeatures(int frame, double *pt, double *co, double *sc) {
  if (frame < 0 || frame > W_ || nslots_[frame] == 0) return 0;
  const int off = slot_off_[frame], ns = nslots_[frame];
  const size_t M = stacks_[frame].n;
  std::vector<uint8_t> v(ns);
  std::vector<float4> c(ns), p(M);
  std::vector<float> s(ns);
  LIO_HIP(hipMemcpyAsync(v.data(), f_valid_.p + off, ns, hipMemcpyDeviceToHost, stream_));
  LIO_HIP(hipMemcpyAsync(c.data(), f_coef_.p + off, ns * sizeof(float4), hipMemcpyDeviceToHost, stream_));
  LIO_HIP(hipMemcpyAsync(s.data(), f_score_.p + off, ns * sizeof(float), hipMemcpyDeviceToHost, stream_));
  LIO_HIP(hipMemcpyAsync(p.data(), stacks_[frame].buf.p, M * sizeof(float4), hipMemcpyDeviceToHost, stream_));
  LIO_HIP(hipStreamSynchronize(stream_));
  size_t k = 0;
  for (int i = 0; i < ns; ++i) {
    if (!v[i]) continue;
    const float4 &pp = p[i % M];
    if (pt) { pt[3 * k] = pp.x; pt[3 * k + 1] = pp.y; pt[3 * k + 2] = pp.z; }
    if (co) { co[4 * k] = c[i].x; co[4 * k + 1] = c[i].y; co[4 * k + 2] = c[i].z; co[4 * k + 3] = c[i].w; }
    if (sc) sc[k] = s[i];
    ++k;
  }
  return k;
}

Rigidd Estimator::LidarPose(int i, const Rigidd &lb) const {
  // Quaterniond rot_li(Rs_i * transform_lb.rot.inverse()); pos_li = Ps_i - rot_li * transform_lb.pos  (Estimator.cc:1448-1449)
  Qd rot = fromRot(Rs_[i] * toRot(qinverse(lb.rot)));
  V3d pos = Ps_[i] - rotate(rot, lb.pos);
  return Rigidd(rot, pos);
}
Rigidf Estimator::RelTransform(int i, const Rigidd &T_pivot, const Rigidd &lb) const {
  return toFloat(compose(rinverse(T_pivot), LidarPose(i, lb)));
}

void Estimator::PushCloud(DeviceCloud &&c, size_t n, int n_before) {
  push_at(stacks_, n_before, std::move(c));
  push_at(size_surf_stack_, n_before, n);
}

bool Estimator::ProcessLaserOdom(const Rigidf &transform_in, const float *surf, size_t n_surf, const float *corner, size_t n_corner,
                                 double stamp, lio_solve_report *rep) {
  (void)corner; (void)n_corner;
  return ProcessLaserOdom(transform_in, reinterpret_cast<const float4 *>(surf), n_surf, false, stamp, rep);
}

// Estimator.cc:430-774
bool Estimator::ProcessLaserOdom(const Rigidf &transform_in, const float4 *surf, size_t n_surf, bool surf_on_device, double stamp,
                                 lio_solve_report *rep) {
  ++laser_odom_recv_count_;
  if (!inited_ && laser_odom_recv_count_ % cfg_.init_window_factor != 0) { last_event_ = EV_SKIPPED; return true; }  // :436-439
  if (!PushFrame(transform_in, reinterpret_cast<const float *>(surf), n_surf, nullptr, 0, stamp, surf_on_device)) return false;
  if (!inited_) {
    if (cir_buf_count_ == W_) {
      bool init_result = false;
      if (!cfg_.imu_factor) {
        init_result = true;
        SetStatesFromLaser();
      } else {
        if (extrinsic_stage_ == 2 && estimate_extrinsic_rotation(all_laser_transforms_, transform_lb_)) extrinsic_stage_ = 1;
        if (extrinsic_stage_ != 2 && (stamp - initial_time_) > 0.1) {
          init_result = RunInitialization();
          initial_time_ = stamp;
        }
      }
      if (init_result) {
        inited_ = true;
        SolveOptimization(rep);
        SlideWindow();
        last_event_ = EV_INITIALISED;
      } else {
        SlideWindow();
        last_event_ = EV_INIT_FAILED;
      }
    } else {
      SlideWindow();
      ++cir_buf_count_;
      last_event_ = EV_FILLING;
    }
    return true;
  }
  bool ok = SolveOptimization(rep);
  SlideWindow();
  last_event_ = EV_SOLVED;
  return ok;
}

void Estimator::SetStatesFromLaser() {  // :507-513, :892-906
  for (int i = 0; i <= W_; ++i) {
    const Rigidf bi = compose(all_laser_transforms_[i].transform, transform_lb_);
    Ps_[i] = vcast<float, double>(bi.pos);
    const Mat3<float> Rf = toRot(normalized(bi.rot));
    for (int k = 0; k < 9; ++k) Rs_[i].m[k] = double(Rf.m[k]);
  }
}

// Estimator.cc:858-958
bool Estimator::RunInitialization() {
  frames_dirty_ = true;
  {
    V3d sum_g;
    for (int i = 0; i < W_; ++i) {
      const Preintegration &pim = *all_laser_transforms_[i + 1].pim;
      sum_g = sum_g + pim.dv / pim.sum_dt;
    }
    const V3d aver_g = sum_g * (1.0 / W_);
    double var = 0;
    for (int i = 0; i < W_; ++i) {
      const Preintegration &pim = *all_laser_transforms_[i + 1].pim;
      const V3d d = pim.dv / pim.sum_dt - aver_g;
      var += dot(d, d);
    }
    var = std::sqrt(var / W_);
    if (var < 0.25) return false;  // "IMU excitation not enough!"
  }
  V3d g_in_laser;
  const bool init_result = imu_initialization(all_laser_transforms_, Vs_, Bgs_, g_in_laser, transform_lb_, R_WI_);
  SetStatesFromLaser();
  M3d R0 = transpose(R_WI_);
  const double yaw = R2ypr(R0 * Rs_[0]).x;
  R0 = ypr2R(V3d(-yaw, 0, 0)) * R0;
  R_WI_ = transpose(R0);
  g_vec_ = R0 * g_in_laser;
  for (int i = 0; i <= cir_buf_count_; ++i) pre_integrations_[i]->repropagate(Bas_[i], Bgs_[i]);
  for (int i = 0; i <= cir_buf_count_; ++i) { Ps_[i] = R0 * Ps_[i]; Rs_[i] = R0 * Rs_[i]; Vs_[i] = R0 * Vs_[i]; }
  return init_result;
}

bool Estimator::PushFrame(const Rigidf &transform_in, const float *surf, size_t n_surf, const float * /*corner*/, size_t /*n_corner*/,
                          double stamp, bool surf_on_device) {
  frames_dirty_ = true;
  // every precondition is checked BEFORE the window is touched: a refused frame leaves the estimator as it was
  if (inited_ && (cfg_.enable_deskew || cfg_.cutoff_deskew) && !cfg_.cutoff_deskew && imu_stamped_.empty()) return false;
  LaserFrame lf;
  lf.time = stamp; lf.transform = transform_in; lf.pim = tmp_pre_integration_;
  push_at(pre_integrations_, n_frames_, tmp_pre_integration_);
  push_at(all_laser_transforms_, n_frames_, lf);
  tmp_pre_integration_ = std::make_shared<Preintegration>(acc_last_, gyr_last_, Bas_[cir_buf_count_], Bgs_[cir_buf_count_], cfg_.pim);
  const int n_before = n_frames_;
  if (n_frames_ < W_ + 1) ++n_frames_;
  DeviceCloud fresh = std::move(stacks_[n_before < W_ + 1 ? n_before : 0]);  // recycle the buffer of the slot being (re)written
  if (!inited_) {  // :474-481: the stacks are the scan-to-map stage's down-sampled clouds, pushed as they are
    fresh.buf.reserve(std::max<size_t>(n_surf, 1));
    if (n_surf)
      LIO_HIP(hipMemcpyAsync(fresh.buf.p, surf, n_surf * sizeof(float4), surf_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream_));
    fresh.n = n_surf;
    fresh.id = ++g_content_id;
    LIO_HIP(hipStreamSynchronize(stream_));  // the source may be reused by the caller right after the call
    PushCloud(std::move(fresh), n_surf, n_before);
    return true;
  }
  // host -> HBM (the only PCIe traffic of the step besides the small state/moment exchanges)
  upload_.buf.reserve(std::max<size_t>(n_surf, 1));
  if (n_surf)
    LIO_HIP(hipMemcpyAsync(upload_.buf.p, surf, n_surf * sizeof(float4), surf_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream_));
  upload_.n = n_surf;
  if (cfg_.enable_deskew || cfg_.cutoff_deskew) {
    if (!cfg_.cutoff_deskew) {
      double time_e = imu_stamped_.back().time;
      Rigidf T_e = imu_stamped_.back().T;
      double time_s = time_e;
      Rigidf T_s = T_e;
      for (int i = int(imu_stamped_.size()) - 1; i >= 0; --i) {
        time_s = imu_stamped_[i].time;
        T_s = imu_stamped_[i].T;
        if (time_e - imu_stamped_[i].time >= 0.1) break;
      }
      Rigidf body_es = compose(rinverse(T_e), T_s);
      float s = float(0.1 / (time_e - time_s));
      Quat<float> qid;
      body_es.rot = slerp(qid, s, body_es.rot, FLT_EPSILON);
      body_es.pos = s * body_es.pos;
      Rigidf tes = compose(compose(transform_lb_, body_es), rinverse(transform_lb_));
      float q[4] = {tes.rot.x, tes.rot.y, tes.rot.z, tes.rot.w}, p[3] = {tes.pos.x, tes.pos.y, tes.pos.z};
      launch_deskew_to_end(upload_.buf.p, int(n_surf), q, p, 10.f, stream_);
    }
    // corner clouds are only consumed under USE_CORNER (off in the shipped build, Estimator.h:55): not processed
    fresh.n = vox_.run(upload_.buf.p, n_surf, cfg_.surf_filter_size, fresh.buf, stream_);
  } else {
    fresh.buf.reserve(std::max<size_t>(n_surf, 1));
    if (n_surf) LIO_HIP(hipMemcpyAsync(fresh.buf.p, upload_.buf.p, n_surf * sizeof(float4), hipMemcpyDeviceToDevice, stream_));
    fresh.n = n_surf;
  }
  size_t nfresh = fresh.n;
  fresh.id = ++g_content_id;
  PushCloud(std::move(fresh), nfresh, n_before);
  return true;
}

void Estimator::FusePivotOnce() {
  if (init_local_map_) return;
  const int pivot = W_ - Wo_;
  const Rigidd lb = toDouble(transform_lb_);
  const Rigidd T_pivot = LidarPose(pivot, lb);
  ConcatArgs ca{};
  int total = 0;
  for (int i = 0; i <= pivot; ++i) {
    ConcatSeg &sg = ca.seg[ca.nseg++];
    sg.src = stacks_[i].buf.p; sg.n = int(stacks_[i].n); sg.dst_off = total; sg.set_intensity = 0; sg.intensity = 0; sg.identity = 0;
    sg.tf = affineOf(RelTransform(i, T_pivot, lb));
    total += sg.n;
  }
  ca.total = total;
  scratch_cloud_.buf.reserve(std::max(total, 1));
  launch_transform_concat(ca, scratch_cloud_.buf.p, stream_);
  scratch_cloud_.n = size_t(total);
  scratch_cloud_.id = ++g_content_id;
  std::swap(stacks_[pivot], scratch_cloud_);
  init_local_map_ = true;
}

void Estimator::BuildLocalMap(lio_solve_report *rep) {
  const double t0 = now_ms();
  const int pivot = W_ - Wo_;
  const Rigidd lb = toDouble(transform_lb_);
  const Rigidd T_pivot = LidarPose(pivot, lb);
  FusePivotOnce();
  std::vector<Rigidf> local_transforms(W_ + 1);
  ConcatArgs ca{};
  int total = 0;
  for (int i = 0; i <= W_; ++i) {
    Rigidf tf = RelTransform(i, T_pivot, lb);
    local_transforms[i] = fromAffine(linearOf(tf), tf.pos);
    if (i < pivot || i == W_) continue;
    ConcatSeg &sg = ca.seg[ca.nseg++];
    sg.src = stacks_[i].buf.p; sg.n = int(stacks_[i].n); sg.dst_off = total;
    if (i == pivot) { sg.identity = 1; sg.set_intensity = 0; sg.intensity = 0; }
    else { sg.identity = 0; sg.set_intensity = 1; sg.intensity = float(i); sg.tf = affineOf(tf); }
    total += sg.n;
  }
  ca.total = total;
  local_.buf.reserve(std::max(total, 1));
  int th = timers_.begin(KT_CONCAT, 32.0 * total, stream_);
  launch_transform_concat(ca, local_.buf.p, stream_);
  timers_.end(th, stream_);
  local_.n = size_t(total);
  VoxParams vp;
  th = timers_.begin(KT_VOXEL, 32.0 * total, stream_);
  local_filtered_.n = vox_.run(local_.buf.p, local_.n, cfg_.surf_filter_size, local_filtered_.buf, stream_, &vp);
  timers_.end(th, stream_);
  const double t1 = now_ms();
  // K-NN grid: cell edge >= sqrt(min_match_sq_dis) so the 27-cell neighbourhood holds every point that can
  // pass the d2[4] < min_match_sq_dis gate (Estimator.cc:1021); beyond it the reference rejects anyway.
  const float cell = std::sqrt(cfg_.min_match_sq_dis) * 1.0001f + 1e-6f;
  th = timers_.begin(KT_KNN_GRID, 32.0 * double(local_filtered_.n), stream_);
  grid_.build(local_filtered_.buf.p, local_filtered_.n, vp.mn, vp.mx, cell, stream_);
  timers_.end(th, stream_);
  // slot layout
  const int keep_mult = (cfg_.keep_features && cfg_.imu_factor) ? 10 : 1;
  total_slots_ = 0;
  for (int i = 0; i <= W_; ++i) {
    slot_off_[i] = int(total_slots_);
    nslots_[i] = 0;
    if (i > pivot) { nslots_[i] = int(stacks_[i].n) * ((i == W_) ? keep_mult : 1); total_slots_ += size_t(nslots_[i]); }
  }
  f_valid_.reserve(std::max<size_t>(total_slots_, 1)); f_coef_.reserve(std::max<size_t>(total_slots_, 1)); f_score_.reserve(std::max<size_t>(total_slots_, 1));
  // feature flags cleared, local transforms and the newest frame's state on the device: one launch (cloud_kernels.h: SolveSetup)
  std::vector<float> tfs(size_t(W_ + 1) * 8, 0.f);
  SolveSetup su{};
  su.ntf = W_ + 1;
  for (int i = 0; i <= W_; ++i) {
    const Rigidf &T = local_transforms[i];
    float *o = &tfs[size_t(i) * 8];
    o[0] = T.rot.x; o[1] = T.rot.y; o[2] = T.rot.z; o[3] = T.rot.w; o[4] = T.pos.x; o[5] = T.pos.y; o[6] = T.pos.z;
    std::memcpy(su.tf[i], o, 8 * sizeof(float));
  }
  std::memcpy(su.odom_T, &tfs[size_t(W_) * 8], 8 * sizeof(float));
  su.set_odom = cfg_.imu_factor ? 1 : 0;
  d_transforms_.reserve(tfs.size());
  launch_solve_setup(su, d_transforms_.p, d_odom_.p, f_valid_.p, total_slots_, stream_);
  // frames pivot+1 .. W-1 (and W when the IMU factor is off): one batched launch
  FeatArgs fa{};
  fa.min_match_sq_dis = cfg_.min_match_sq_dis; fa.min_plane_dis = cfg_.min_plane_dis;
  const int last_static = cfg_.imu_factor ? W_ - 1 : W_;
  for (int i = pivot + 1; i <= last_static; ++i) {
    FeatFrame &f = fa.fr[fa.nframes++];
    f.stack = stacks_[i].buf.p; f.M = int(stacks_[i].n); f.slot_off = slot_off_[i]; f.tf_index = i;
    fa.max_M = std::max(fa.max_M, f.M);
  }
  {
    double mq = 0;
    for (int k = 0; k < fa.nframes; ++k) mq += fa.fr[k].M;
    // 16(M+N) + 8*K*M + 32*M bytes per call, K = 5
    // The Wo-1 older frames do not depend on the newest frame's Gauss-Newton rounds: their batched launch goes to a second
    // stream and fills the CUs the serial rows/update kernels of that loop leave idle; joined before the solve.
    hipStream_t sf = cfg_.imu_factor ? stream2_ : stream_;
    if (sf != stream_) {
      LIO_HIP(hipEventRecord(ev_fork_, stream_));
      LIO_HIP(hipStreamWaitEvent(sf, ev_fork_, 0));
    }
    th = timers_.begin(KT_FEATURES, 16.0 * (mq + double(local_filtered_.n)) + 40.0 * mq + 32.0 * mq, sf);
    launch_features(fa, d_transforms_.p, grid_.sorted(), grid_.cells(), grid_.desc(), f_valid_.p, f_coef_.p, f_score_.p, nullptr, sf);
    timers_.end(th, sf);
    if (sf != stream_) LIO_HIP(hipEventRecord(ev_join_, sf));
  }
  laser_odom_iters_ = 0; laser_odom_kz_ = 0;
  if (cfg_.imu_factor) {
    // CalculateLaserOdom: <= 10 dependent rounds, no host round trip inside (the device carries the
    // transform and the convergence flag; later launches turn into no-ops)
    OdomState st{};
    bool have_state = false;  // a converged peek already brought the final state to the host
    std::memcpy(st.T, &tfs[size_t(W_) * 8], 8 * sizeof(float));   // (on the device since launch_solve_setup)
    const int M = int(stacks_[W_].n);
    const bool mail = host_signal_ && !timers_.on;
    HostSignal sig{};
    if (M > 0) {
      const int lpq = 8;   // lanes per query (the K-NN result does not depend on it; the row partition does)
      const int nb = odom_round_blocks(M, lpq);
      d_odom_partials_.reserve(size_t(nb) * 28);
      FeatArgs fo{};
      fo.min_match_sq_dis = cfg_.min_match_sq_dis; fo.min_plane_dis = cfg_.min_plane_dis;
      fo.nframes = 1; fo.max_M = M;
      fo.fr[0].stack = stacks_[W_].buf.p; fo.fr[0].M = M; fo.fr[0].tf_index = 0; fo.fr[0].slot_off = slot_off_[W_];
      // Launch in chunks and peek at the device-side convergence flag between them: a peek costs one small
      // D2H (~10 us) and saves the no-op launches of every skipped round.
      const int chunk_end[4] = {3, 5, 7, 10};
      int chunk = 0;
      for (int iter = 0; iter < 10; ++iter) {
        if (iter == chunk_end[chunk]) {
          if (mail) {
            wait_host_signal(sig, stream_);   // the round before this one has posted its state
          } else {
            LIO_HIP(hipMemcpyAsync(h_odom_, d_odom_.p, sizeof(st), hipMemcpyDeviceToHost, stream_));  // pinned: a pageable target costs ~10 us more
            LIO_HIP(hipStreamSynchronize(stream_));
          }
          st = *h_odom_;
          if (st.converged) { have_state = true; break; }
          ++chunk;
        }
        if (mail) { sig.flag = h_signal_ + 128; sig.seq = ++signal_seq_[1]; }
        // one round = search + plane fit + rows (k_odom_round) and fold + 6x6 step (k_odom_update_wide)
        const double ns = keep_mult > 1 ? double(iter + 1) * M : double(M);
        int t1h = timers_.begin(KT_ODOM_FEATURES, 16.0 * (double(M) + double(local_filtered_.n)) + 72.0 * M + 33.0 * ns, stream_);
        launch_odom_round(fo, slot_off_[W_], iter, keep_mult > 1 ? 1 : 0, d_odom_.p, grid_.sorted(), grid_.cells(), grid_.desc(), f_valid_.p, f_coef_.p,
                          f_score_.p, d_odom_partials_.p, stream_, mail ? h_odom_ : nullptr, sig, lpq);
        timers_.end(t1h, stream_);
      }
    }
    // the older frames' features (second stream) must be complete before anything later on stream_ reads them; the host
    // itself only needs the final state, which a converged peek has already delivered
    LIO_HIP(hipStreamWaitEvent(stream_, ev_join_, 0));
    if (!have_state) {
      if (sig.flag) {
        wait_host_signal(sig, stream_);
      } else {
        LIO_HIP(hipMemcpyAsync(h_odom_, d_odom_.p, sizeof(st), hipMemcpyDeviceToHost, stream_));
        LIO_HIP(hipStreamSynchronize(stream_));
      }
      st = *h_odom_;
      timers_.resolve();
    }
    laser_odom_iters_ = st.iters;
    laser_odom_kz_ = st.degenerate ? st.kz : 0;
    laser_odom_transform_ = Rigidf(Quat<float>(st.T[3], st.T[0], st.T[1], st.T[2]), Vec3<float>(st.T[4], st.T[5], st.T[6]));
    if (keep_mult > 1) nslots_[W_] = int(stacks_[W_].n) * std::max(1, st.iters);
  } else {
    LIO_HIP(hipStreamSynchronize(stream_));
    timers_.resolve();
  }
  const double t2 = now_ms();
  if (rep) {
    rep->ms_build_map = t1 - t0; rep->ms_features = t2 - t1; rep->n_local_map = int(local_filtered_.n);
    rep->laser_odom_iterations = laser_odom_iters_; rep->laser_odom_kz = laser_odom_kz_;
  }
}

void Estimator::VectorToParams(WindowParams &P) const {
  const int pivot = W_ - Wo_;
  P.Wo = Wo_;
  P.pose.resize(Wo_ + 1); P.sb.resize(Wo_ + 1);
  for (int i = 0, oi = pivot; i <= Wo_; ++i, ++oi) {
    Qd q = fromRot(Rs_[oi]);
    P.pose[i] = {Ps_[oi].x, Ps_[oi].y, Ps_[oi].z, q.x, q.y, q.z, q.w};
    P.sb[i] = {Vs_[oi].x, Vs_[oi].y, Vs_[oi].z, Bas_[oi].x, Bas_[oi].y, Bas_[oi].z, Bgs_[oi].x, Bgs_[oi].y, Bgs_[oi].z};
  }
  P.ex = {transform_lb_.pos.x, transform_lb_.pos.y, transform_lb_.pos.z, transform_lb_.rot.x, transform_lb_.rot.y, transform_lb_.rot.z,
          transform_lb_.rot.w};
}

void Estimator::ParamsToVector(const WindowParams &P) {  // DoubleToVector with yaw re-anchoring (Estimator.cc:2479-2568)
  const int pivot = W_ - Wo_;
  const V3d origin_P0 = Ps_[pivot];
  const V3d origin_R0 = R2ypr(Rs_[pivot]);
  const M3d R00 = toRot(normalized(Qd(P.pose[0][6], P.pose[0][3], P.pose[0][4], P.pose[0][5])));
  const V3d origin_R00 = R2ypr(R00);
  const double y_diff = origin_R0.x - origin_R00.x;
  M3d rot_diff = ypr2R(V3d(y_diff, 0, 0));
  if (std::fabs(std::fabs(origin_R0.y) - 90) < 1.0 || std::fabs(std::fabs(origin_R00.y) - 90) < 1.0) rot_diff = Rs_[pivot] * transpose(R00);
  {
    Rigidd trans_pivot(fromRot(Rs_[pivot]), Ps_[pivot]);
    Rigidd trans_opt_pivot(fromRot(rot_diff * R00), origin_P0);
    for (int idx = 0; idx < pivot; ++idx) {
      Rigidd trans_idx(fromRot(Rs_[idx]), Ps_[idx]);
      Rigidd t = compose(compose(trans_opt_pivot, rinverse(trans_pivot)), trans_idx);
      Ps_[idx] = t.pos;
      Rs_[idx] = toRot(normalized(t.rot));
    }
  }
  for (int i = 0, oi = pivot; i <= Wo_; ++i, ++oi) {
    Qd qi(P.pose[i][6], P.pose[i][3], P.pose[i][4], P.pose[i][5]);
    Rs_[oi] = rot_diff * toRot(normalized(qi));
    Ps_[oi] = rot_diff * V3d(P.pose[i][0] - P.pose[0][0], P.pose[i][1] - P.pose[0][1], P.pose[i][2] - P.pose[0][2]) + origin_P0;
    Vs_[oi] = rot_diff * V3d(P.sb[i][0], P.sb[i][1], P.sb[i][2]);
    Bas_[oi] = V3d(P.sb[i][3], P.sb[i][4], P.sb[i][5]);
    Bgs_[oi] = V3d(P.sb[i][6], P.sb[i][7], P.sb[i][8]);
  }
  transform_lb_.pos = Vec3<float>(float(P.ex[0]), float(P.ex[1]), float(P.ex[2]));
  transform_lb_.rot = Quat<float>(float(P.ex[6]), float(P.ex[3]), float(P.ex[4]), float(P.ex[5]));
}

void Estimator::LidarEval(const WindowParams &P, std::vector<FrameMoments> &m) {
  LidarLaunch(P);
  LidarWait(m);
}

void Estimator::FillMomentArgs(MomentArgs &ma, int &max_slots) const {
  const int pivot = W_ - Wo_;
  ma = MomentArgs{};
  max_slots = 0;
  for (int i = 1; i <= Wo_; ++i) {
    MomentFrame &f = ma.fr[ma.nframes++];
    const int idx = pivot + i;
    f.stack = stacks_[idx].buf.p; f.M = std::max<int>(1, int(stacks_[idx].n)); f.slot_off = slot_off_[idx]; f.nslots = nslots_[idx];
    f.slot_begin = 0; f.slot_end = f.nslots;
    if (Sharded()) {  // contiguous share of this frame's factor slots
      f.slot_begin = int((long long)f.nslots * shard_rank_ / shard_world_);
      f.slot_end = int((long long)f.nslots * (shard_rank_ + 1) / shard_world_);
    }
    max_slots = std::max(max_slots, f.nslots);
  }
  ma.blocks_per_frame = moment_blocks_per_frame(max_slots);
  ma.form = moments_form_;
  // With the resident form configured, BOTH paths use its partition (blocks per frame so that a lane holds <= per_lane
  // 64-slot chunks per wave, fp64-MFMA form): the launch path — taken when a pass cannot use the resident kernel (kernel timing, factor
  // sharding, stream_sync) — then yields bit-identical moments.
  const int rb = ResidentBpf(max_slots, ma.nframes);
  if (rb > 0) { ma.blocks_per_frame = rb; ma.form = 1; }
}

// Blocks per frame of the resident form's partition (0: the window does not fit) and, in *per_lane, the residuals a lane keeps.
// Fewer residuals per lane = more blocks = a shorter accumulate phase (1.7 us of MFMA per wave at four, 0.85 at two); the frame
// fold costs one memory round trip as long as a frame's blocks fit one batch of loads (RES_FOLD_BATCH = 64).  So: the smallest
// per-lane count whose blocks are all co-resident (<= 256) with at most 64 per frame.  A pure function of the window's slot
// counts, so the partition — and with it every bit of the result — does not depend on how a pass is executed.
int Estimator::ResidentBpf(int max_slots, int nframes, int *per_lane) const {
  if (per_lane) *per_lane = 0;
  if (!resident_moments_ || moments_form_ == 2) return 0;
  for (int r : {1, 2, 4, 8}) {
    if (res_per_lane_ > 0 && r != res_per_lane_) continue;
    const int b = resident_blocks_per_frame(max_slots, nframes, r);
    if (b > 0 && (b <= 64 || r == 8 || res_per_lane_ > 0)) { if (per_lane) *per_lane = r; return b; }
  }
  return 0;
}

void Estimator::ResidentLaunchKernel(unsigned first_seq) {
  ResidentArgs ra{h_res_door_, h_res_out_, h_res_words_, first_seq, res_timeout_ticks_, d_res_relay_.p, d_res_part_.p, g_debug_timing ? 1 : 0};
  res_launch_seq_ = first_seq;   // (a launch's STOP value is derived from it; see ResidentAwaitWord for the one case where the HBM copy must be cleared)
  launch_lidar_moments_resident(res_args_, ra, res_lanes_, f_valid_.p, f_coef_.p, stream_);
}

// The resident kernel of this solve: launched behind everything the feature stage enqueued on stream_; it returns when the host
// writes LIO_RES_STOP (ResidentEnd) or after res_timeout_ticks_ without a doorbell.
bool Estimator::ResidentBegin(const MomentArgs &ma) {
  if (!res_allowed_ || resident_never_ || !host_signal_ || timers_.on || Sharded() || rccl_comm_) return false;
  int max_slots = 0;
  for (int k = 0; k < ma.nframes; ++k) max_slots = std::max(max_slots, ma.fr[k].nslots);
  int per_lane = 0;
  if (ResidentBpf(max_slots, ma.nframes, &per_lane) != ma.blocks_per_frame || ma.blocks_per_frame <= 0) return false;
  if (g_active_solves.load(std::memory_order_relaxed) > 1) return false;
  if (res_seq_ > 0xF0000000u) {   // 32-bit sequence numbers: start over long before they wrap (no launch is in flight here)
    LIO_HIP(hipStreamSynchronize(stream_));
    LIO_HIP(hipMemset(d_res_part_.p, 0, sizeof(double) * LIO_RES_MAX_BLOCKS * LIO_MOMENT_OUT));
    std::memset(h_res_words_, 0, sizeof(unsigned) * (LIO_MAX_FRAMES + 2));
    res_seq_ = 0;
  }
  if (g_resident_moments.fetch_add(1) >= kMaxResidentMoments) { g_resident_moments.fetch_sub(1); return false; }
  struct Admission { bool keep = false; ~Admission() { if (!keep) g_resident_moments.fetch_sub(1); } } admission;   // released if the launch throws
  res_args_ = ma; res_lanes_ = per_lane;
  res_bpf_ = ma.blocks_per_frame; res_nframes_ = ma.nframes;
  for (int f = 0; f < res_nframes_; ++f) {   // idle doorbell: neither the expected sequence number nor STOP
    __atomic_store_n(reinterpret_cast<unsigned long long *>(h_res_door_ + f * LIO_RES_DOOR + 7), 0ull, __ATOMIC_RELEASE);
    __atomic_store_n(reinterpret_cast<unsigned long long *>(h_res_door_ + f * LIO_RES_DOOR + 15), 0ull, __ATOMIC_RELEASE);
  }
  if (res_time_launch_) {
    hipEvent_t a, b;
    LIO_HIP(hipEventCreate(&a)); LIO_HIP(hipEventCreate(&b));
    LIO_HIP(hipEventRecord(a, stream_));
    res_launch_events_.push_back({a, b});
  }
  res_relaunches_ = 0;
  ResidentLaunchKernel(res_seq_ + 1);
  res_active_ = true; admission.keep = true;
  return true;
}

int Estimator::ResidentLaunchStats(double *total_ms) {
  LIO_HIP(hipStreamSynchronize(stream_));
  for (auto &ev : res_launch_events_) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) { res_launch_ms_ += ms; ++res_launches_; }
    (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second);
  }
  res_launch_events_.clear();
  if (total_ms) *total_ms = res_launch_ms_;
  return res_launches_;
}

void Estimator::ResidentRing(const MomentArgs &ma) {
  const unsigned seq = ++res_seq_;
  const double sd = double(seq);
  for (int f = 0; f < res_nframes_; ++f) {
    double *d = h_res_door_ + f * LIO_RES_DOOR;
    const MomentFrame &fr = ma.fr[f];
    // payload first, the sequence slot of each cache line last (x86 keeps the order of stores; the GPU reads a line at a time)
    for (int k = 0; k < 7; ++k) d[k] = fr.R[k];
    __atomic_store_n(reinterpret_cast<unsigned long long *>(d + 7), *reinterpret_cast<const unsigned long long *>(&sd), __ATOMIC_RELEASE);
    d[8] = fr.R[7]; d[9] = fr.R[8]; d[10] = fr.t[0]; d[11] = fr.t[1]; d[12] = fr.t[2]; d[13] = 0.0; d[14] = 0.0;
    __atomic_store_n(reinterpret_cast<unsigned long long *>(d + 15), *reinterpret_cast<const unsigned long long *>(&sd), __ATOMIC_RELEASE);
  }
}

// Waits for frame f's completion word of the pass in flight.  A relay timeout (this host thread was held up for > 200 ms before
// it rang) can only show while NO frame of the pass has been posted: the relay gives up between passes, and a pass that was
// started is posted whole.
void Estimator::ResidentAwaitWord(int f) {
  const unsigned seq = res_seq_;
  const volatile unsigned *w = h_res_words_;
  for (unsigned long it = 1;; ++it) {
    if (__atomic_load_n(w + f, __ATOMIC_ACQUIRE) == seq) return;
    if (__atomic_load_n(w + LIO_MAX_FRAMES, __ATOMIC_ACQUIRE) == LIO_RES_EXPIRED) {
      if (__atomic_load_n(w + f, __ATOMIC_ACQUIRE) == seq) return;
      if (f > 0 && __atomic_load_n(w + 0, __ATOMIC_ACQUIRE) == seq) throw DeviceError("resident moments kernel gave up in the middle of a pass");
      // let that launch drain and start a new one for the pass that is pending; its doorbell is still rung.  Twice at most: a
      // kernel that keeps expiring is not being scheduled whole (its blocks are not co-resident) and no retry will change that.
      LIO_HIP(hipStreamSynchronize(stream_));
      if (++res_relaunches_ > 2) throw DeviceError("resident moments kernel expired three times within one solve (its blocks are not co-resident?)");
      h_res_words_[LIO_MAX_FRAMES] = 0;
      // The expired relay left ITS stop value in the HBM copy of the doorbell.  If that launch never served a pass, the one that
      // replaces it starts at the same sequence number and has the same stop value: clear the copy, or the new workers leave on it
      // before the new relay republishes the pending pass.
      LIO_HIP(hipMemsetAsync(d_res_relay_.p, 0, sizeof(double) * LIO_MAX_FRAMES * LIO_RES_DOOR, stream_));
      ResidentLaunchKernel(seq);
      continue;
    }
    __builtin_ia32_pause();
    if ((it & 0xFFFFu) == 0) {
      const hipError_t e = hipStreamQuery(stream_);
      if (e != hipErrorNotReady && e != hipSuccess) throw DeviceError(std::string("resident moments pass failed: ") + hipGetErrorString(e));
      if (e == hipSuccess && h_res_words_[LIO_MAX_FRAMES] != LIO_RES_EXPIRED && __atomic_load_n(w + f, __ATOMIC_ACQUIRE) != seq)
        throw DeviceError("resident moments kernel ended without posting its pass");   // the kernel is gone although nobody stopped it
    }
  }
}

void Estimator::ResidentUnpackFrame(int f, FrameMoments &fm) {
  // the device posts the upper triangle of the 13 x 13 tile (it is symmetric bit for bit); S is the padded 16 x 16 tile
  static const struct TriMap { int at[256]; TriMap() { for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { const int a = std::min(i, j), b = std::max(i, j); at[i * 16 + j] = (b < 13) ? a * 13 - a * (a - 1) / 2 + (b - a) : -1; } } } tri;
  const double *rec = h_res_out_ + size_t(f) * LIO_RES_OUT;
  for (int k = 0; k < 256; ++k) fm.S[k] = tri.at[k] >= 0 ? rec[tri.at[k]] : 0.0;
  fm.cost = rec[LIO_RES_NTRI]; fm.count = rec[LIO_RES_NTRI + 1];
  double *o = h_moment_out_ + size_t(f) * LIO_MOMENT_OUT;   // the landing zone of the launch path doubles as "the last moments"
  o[256] = fm.cost; o[257] = fm.count;
}

// bookkeeping of a finished pass (all frames in): device-side phase stamps, busy time, algorithmic bytes
void Estimator::ResidentPassDone() {
  const int nf = res_nframes_;
  double busy = 0, polls = 0;
  for (int f = 0; f < nf; ++f) {
    const double *rec = h_res_out_ + size_t(f) * LIO_RES_OUT;
    for (int q = 0; q < 4; ++q) res_diag_us_[q] += rec[258 + q] * res_tick_us_ / nf;
    polls += rec[262] / nf;
    res_relay_us_ += rec[263] * res_tick_us_ / nf;
    busy = std::max(busy, rec[261]);
  }
  res_busy_us_ += busy * res_tick_us_;   // doorbell copy seen -> sums posted, slowest frame
  { double nres = 0; for (int f = 0; f < nf; ++f) nres += res_args_.fr[f].nslots; res_bytes_ += 60.0 * nres; }   // 60 B per lidar residual
  res_polls_ += polls; ++res_passes_; ++res_passes_total_;
}

void Estimator::ResidentWaitFrame(int f, FrameMoments &fm) {
  ResidentAwaitWord(f);
  ResidentUnpackFrame(f, fm);
  if (f == res_nframes_ - 1) ResidentPassDone();
}

void Estimator::ResidentWait(std::vector<FrameMoments> &m) {
  for (int f = 0; f < res_nframes_; ++f) ResidentAwaitWord(f);
  for (int f = 0; f < res_nframes_; ++f) ResidentUnpackFrame(f, m[f + 1]);
  ResidentPassDone();
}

void Estimator::ResidentEnd() {
  res_allowed_ = false;
  if (!res_active_) return;
  const double stop = LIO_RES_STOP(res_launch_seq_);
  const unsigned long long bits = *reinterpret_cast<const unsigned long long *>(&stop);
  for (int f = 0; f < res_nframes_; ++f) {
    __atomic_store_n(reinterpret_cast<unsigned long long *>(h_res_door_ + f * LIO_RES_DOOR + 7), bits, __ATOMIC_RELEASE);
    __atomic_store_n(reinterpret_cast<unsigned long long *>(h_res_door_ + f * LIO_RES_DOOR + 15), bits, __ATOMIC_RELEASE);
  }
  res_active_ = false;   // the kernel leaves within one poll; whatever is enqueued on stream_ next is ordered behind it
  g_resident_moments.fetch_sub(1);
  if (res_time_launch_ && !res_launch_events_.empty()) (void)hipEventRecord(res_launch_events_.back().second, stream_);
}

void Estimator::LidarLaunch(const WindowParams &P) {
  const double t_dbg0 = now_ms();
  struct DbgAcc { Estimator *e; double t0; ~DbgAcc() { e->dbg_eval_ms_ += now_ms() - t0; } } dbg_acc{this, t_dbg0};
  MomentArgs ma;
  int max_slots = 0;
  FillMomentArgs(ma, max_slots);
  for (int i = 1; i <= Wo_; ++i) relative_lidar_pose(P.pose[0].data(), P.pose[i].data(), P.ex.data(), ma.fr[i - 1].R, ma.fr[i - 1].t);
  if (res_active_ || ResidentBegin(ma)) {
    res_t_ring_ = now_ms();
    ResidentRing(ma);
    if (g_debug_timing) {   // ring -> the relay's echo of the sequence number: the inbound PCIe leg + one word back
      const volatile unsigned *echo = h_res_words_ + LIO_MAX_FRAMES + 1;
      for (unsigned long it = 0; it < 2000000ul && __atomic_load_n(echo, __ATOMIC_ACQUIRE) != res_seq_; ++it) __builtin_ia32_pause();
      res_echo_ms_ += now_ms() - res_t_ring_;
    }
    return;
  }
  d_moment_partials_.reserve(size_t(ma.nframes) * ma.blocks_per_frame * LIO_MOMENT_OUT);
  d_moment_out_.reserve(size_t(LIO_MAX_FRAMES) * LIO_MOMENT_OUT);
  double nres = 0;
  for (int k = 0; k < ma.nframes; ++k) nres += ma.fr[k].nslots;
  int th = timers_.begin(KT_MOMENTS, 60.0 * nres, stream_);  // 60 B read per lidar residual
  // k_moment_reduce stores its Wo x 260 doubles directly into pinned, device-mapped host memory: no copy
  // command, only the kernel-completion wait (kernel end = system-scope release, so the host sees the data).
  if (rccl_comm_) {
    // per-shard moments -> whole-window moments without leaving HBM: fold into a device buffer, SUM all-reduce over xGMI on the
    // same stream, then the 10 KB result goes to the pinned landing zone
    launch_lidar_moments(ma, f_valid_.p, f_coef_.p, d_moment_partials_.p, d_moment_out_.p, stream_);
    rccl_all_reduce_sum_f64(rccl_comm_, d_moment_out_.p, size_t(Wo_) * LIO_MOMENT_OUT, stream_);
    LIO_HIP(hipMemcpyAsync(h_moment_out_, d_moment_out_.p, sizeof(double) * Wo_ * LIO_MOMENT_OUT, hipMemcpyDeviceToHost, stream_));
  } else {
    moment_signal_ = HostSignal();
    if (host_signal_ && !timers_.on) {
      // one completion word per block of the kernel that writes the result: k_moment_reduce's (frames, 3) grid
      moment_signal_.flag = h_signal_; moment_signal_.seq = ++signal_seq_[0]; moment_signal_.nslots = 3 * ma.nframes;
    }
    launch_lidar_moments(ma, f_valid_.p, f_coef_.p, d_moment_partials_.p, h_moment_out_, stream_, moment_signal_);
  }
  timers_.end(th, stream_);
}

bool Estimator::BenchBatchedMoments(int B, int reps, double *avg_ms, double *bytes) {
  if (B < 1 || reps < 1 || total_slots_ == 0 || !init_local_map_) return false;
  const int pivot = W_ - Wo_;
  WindowParams P;
  VectorToParams(P);
  // replicate the feature slots and the stacks B times (distinct addresses: no cache reuse across windows)
  DBuf<uint8_t> valid_b; DBuf<float4> coef_b, stack_b;
  size_t stack_pts = 0;
  for (int i = pivot + 1; i <= W_; ++i) stack_pts += stacks_[i].n;
  valid_b.reserve(size_t(B) * total_slots_); coef_b.reserve(size_t(B) * total_slots_); stack_b.reserve(std::max<size_t>(size_t(B) * stack_pts, 1));
  std::vector<MomentFrame> frames;
  int max_slots = 0;
  double nres = 0;
  for (int b = 0; b < B; ++b) {
    LIO_HIP(hipMemcpyAsync(valid_b.p + size_t(b) * total_slots_, f_valid_.p, total_slots_, hipMemcpyDeviceToDevice, stream_));
    LIO_HIP(hipMemcpyAsync(coef_b.p + size_t(b) * total_slots_, f_coef_.p, total_slots_ * sizeof(float4), hipMemcpyDeviceToDevice, stream_));
    size_t off = size_t(b) * stack_pts;
    for (int i = 1; i <= Wo_; ++i) {
      const int idx = pivot + i;
      if (stacks_[idx].n) LIO_HIP(hipMemcpyAsync(stack_b.p + off, stacks_[idx].buf.p, stacks_[idx].n * sizeof(float4), hipMemcpyDeviceToDevice, stream_));
      MomentFrame f{};
      f.stack = stack_b.p + off; f.M = std::max<int>(1, int(stacks_[idx].n)); f.slot_off = int(size_t(b) * total_slots_) + slot_off_[idx];
      f.nslots = nslots_[idx]; f.slot_begin = 0; f.slot_end = f.nslots;
      relative_lidar_pose(P.pose[0].data(), P.pose[i].data(), P.ex.data(), f.R, f.t);
      frames.push_back(f);
      off += stacks_[idx].n;
      max_slots = std::max(max_slots, f.nslots);
      nres += f.nslots;
    }
  }
  if (size_t(B) * total_slots_ > size_t(INT_MAX)) return false;
  const int nf = int(frames.size()), bpf = moment_blocks_per_frame_batched(max_slots, nf);
  DBuf<MomentFrame> d_frames; DBuf<double> partials, out;
  d_frames.reserve(nf); partials.reserve(size_t(nf) * bpf * LIO_MOMENT_OUT); out.reserve(size_t(nf) * LIO_MOMENT_OUT);
  LIO_HIP(hipMemcpyAsync(d_frames.p, frames.data(), sizeof(MomentFrame) * nf, hipMemcpyHostToDevice, stream_));
  hipEvent_t e0, e1;
  LIO_HIP(hipEventCreate(&e0)); LIO_HIP(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) launch_lidar_moments_batched(d_frames.p, nf, bpf, max_slots, valid_b.p, coef_b.p, partials.p, out.p, stream_, moments_form_);
  LIO_HIP(hipEventRecord(e0, stream_));
  for (int r = 0; r < reps; ++r) launch_lidar_moments_batched(d_frames.p, nf, bpf, max_slots, valid_b.p, coef_b.p, partials.p, out.p, stream_, moments_form_);
  LIO_HIP(hipEventRecord(e1, stream_));
  LIO_HIP(hipStreamSynchronize(stream_));
  float ms = 0;
  LIO_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (avg_ms) *avg_ms = double(ms) / reps;
  if (bytes) *bytes = 60.0 * nres;  // 60 B read per lidar residual
  return true;
}

// Frame i (1-based) of the pass in flight, as soon as its completion word is in — only the resident kernel posts per frame.
bool Estimator::LidarWaitFrame(int i, FrameMoments &fm) {
  if (!res_active_) return false;
  const double t_dbg0 = now_ms();
  ResidentWaitFrame(i - 1, fm);
  const double t1 = now_ms();
  dbg_eval_ms_ += t1 - t_dbg0; dbg_sync_ms_ += t1 - t_dbg0;
  if (i == res_nframes_) { ++dbg_eval_n_; res_ring_to_done_ms_ += t1 - res_t_ring_; }
  return true;
}

void Estimator::LidarWait(std::vector<FrameMoments> &m) {
  const double t_dbg0 = now_ms();
  struct DbgAcc { Estimator *e; double t0; ~DbgAcc() { e->dbg_eval_ms_ += now_ms() - t0; e->dbg_eval_n_++; } } dbg_acc{this, t_dbg0};
  if (res_active_) { ResidentWait(m); dbg_sync_ms_ += now_ms() - t_dbg0; res_ring_to_done_ms_ += now_ms() - res_t_ring_; return; }
  if (moment_signal_.flag && !rccl_comm_) wait_host_signal(moment_signal_, stream_);
  else LIO_HIP(hipStreamSynchronize(stream_));
  dbg_sync_ms_ += now_ms() - t_dbg0;
  timers_.resolve();
  if (shard_world_ > 1 && allreduce_ && !rccl_comm_) {
    // per-shard moments -> whole-window moments through the caller's callback (gloo on CPU hosts; the RCCL form never gets here)
    if (allreduce_(h_moment_out_, Wo_ * LIO_MOMENT_OUT, allreduce_user_) != 0) throw std::runtime_error("factor-sharding all-reduce failed");
  }
  for (int i = 1; i <= Wo_; ++i) {
    const double *src = h_moment_out_ + size_t(i - 1) * LIO_MOMENT_OUT;
    std::memcpy(m[i].S, src, 256 * sizeof(double));
    m[i].cost = src[256]; m[i].count = src[257];
  }
}

bool Estimator::SolveOptimizationHost(lio_solve_report *rep) {
  if (cir_buf_count_ < W_ && cfg_.imu_factor) return false;
  const double t_total0 = now_ms();
  lio_solve_report local{};
  lio_solve_report &R = rep ? *rep : local;
  std::memset(&R, 0, sizeof(R));
  bool turn_off = true;
  struct ActiveSolve { ActiveSolve() { g_active_solves.fetch_add(1); } ~ActiveSolve() { g_active_solves.fetch_sub(1); } } active_solve;
  BuildLocalMap(&R);
  // from here to the end of the solve the lidar passes may come from ONE resident kernel (begun by the first LidarLaunch)
  struct ResidentScope { Estimator *e; ~ResidentScope() { e->ResidentEnd(); } } resident_scope{this};
  res_allowed_ = true;
  const double t_prep0 = now_ms();
  const int pivot = W_ - Wo_;
  WindowParams P;
  VectorToParams(P);
  P.ex_constant = (cfg_.extrinsic_stage == 0 || !cfg_.opt_extrinsic);
  WindowSystem sys;
  sys.Wo = Wo_;
  sys.use_lidar = cfg_.point_distance_factor;
  sys.pim.assign(Wo_, nullptr);
  if (cfg_.imu_factor)
    for (int i = 0; i < Wo_; ++i) {
      auto &pi = pre_integrations_[pivot + i + 1];
      if (pi && pi->sum_dt <= 10.0) sys.pim[i] = pi;
    }
  JoinMarg();  // the previous solve's marginalization has had the map + feature stages to finish
  if (cfg_.marginalization_factor && last_marg_) sys.prior = last_marg_;
  if (cfg_.prior_factor) {
    sys.use_prior_factor = true;
    Rigidd t = toDouble(transform_lb_);
    sys.prior_pos = t.pos; sys.prior_rot = t.rot;
  }
  sys.lidar_eval = [this](const WindowParams &Pq, std::vector<FrameMoments> &m) { LidarEval(Pq, m); };
  sys.lidar_launch = [this](const WindowParams &Pq) { LidarLaunch(Pq); };
  sys.lidar_wait = [this](std::vector<FrameMoments> &m) { LidarWait(m); };
  sys.lidar_wait_frame = [this](int i, FrameMoments &fm) { return LidarWaitFrame(i, fm); };
  R.ms_prepare = now_ms() - t_prep0;
  // Group costs at the initial point (Estimator.cc:1924-1954) and the convergence_flag_ logic (:1956-1984).
  // The reference evaluates the three groups, then Ceres linearises again at the same point; here ONE device
  // pass yields both — unless the flag logic changes the problem (prior dropped / extrinsic frozen).
  SolveSummary s;
  Linearization first;
  {
    Layout lay = WindowSystem::solve_layout(P);
    first.costs = sys.evaluate(P, lay, 1 | 2 | 4 | 8, false, &first.H, &first.g, &first.m);
    first.valid = true;
    const WindowSystem::Costs &gc = first.costs;
    R.cost_pim_before = gc.pim; R.cost_ppp_before = gc.ppp; R.cost_marg_before = gc.marg;
    if (cfg_.imu_factor) turn_off = gc.pim > 1e3;
    const double ratio = gc.marg / (gc.ppp + gc.pim);
    if (!convergence_flag_ && !turn_off && ratio <= 2 && ratio != 0) convergence_flag_ = true;
    if (!convergence_flag_) {
      if (!P.ex_constant || sys.prior) first.valid = false;
      P.ex_constant = true;
      last_marg_.reset();
      sys.prior.reset();
    }
  }
  const double t_opt0 = now_ms();
  {
  // Factor sharding: every linearisation is a collective, so every rank must take the same number of them.  A per-rank
  // wall-clock cap (Estimator.cc:1921) could stop one rank an iteration earlier than its peers and leave an unmatched
  // all-reduce behind; the sharded mode therefore terminates on the iteration / tolerance rules only.
  const double time_cap = Sharded() ? -1.0 : cfg_.max_solver_time;
  s = solve_dogleg(sys, P, cfg_.max_num_iterations, time_cap, &first);
  R.ms_opt = now_ms() - t_opt0;
  if (g_debug_timing)
    std::fprintf(stderr, "[lio_hip timing] dogleg: chol %.3f ms, candidate evaluate %.3f ms | evaluate x%d: launch %.3f prior %.3f imu %.3f wait %.3f assemble %.3f\n",
                 s.ms_chol, s.ms_eval, sys.eclk.n, sys.eclk.launch, sys.eclk.prior, sys.eclk.imu, sys.eclk.wait, sys.eclk.assemble);
  }
  R.iterations = s.iterations; R.successful_steps = s.successful; R.termination = s.termination;
  R.initial_cost = s.initial_cost; R.final_cost = s.final_cost;
  for (size_t k = 0; k < s.trace.size() && k < 32; ++k) R.cost_trace[k] = s.trace[k];
  ParamsToVector(P);
  R.turn_off = turn_off; R.convergence_flag = convergence_flag_;
  if (cfg_.marginalization_factor && !turn_off) {
    const double tm0 = now_ms();
    WindowParams M;
    VectorToParams(M);
    M.ex_constant = false;
    auto msys = std::make_shared<WindowSystem>();
    msys->Wo = Wo_;
    msys->use_lidar = cfg_.point_distance_factor;
    msys->pim.assign(Wo_, nullptr);
    if (cfg_.imu_factor) {
      auto &pi = pre_integrations_[pivot + 1];
      if (pi && pi->sum_dt < 10.0) msys->pim[0] = pi;
    }
    msys->prior = last_marg_;
    const bool have_moments = msys->use_lidar && !s.final_moments.empty();
    if (async_marg_ && (have_moments || !msys->use_lidar)) {
      // host-only from here (the lidar moments at the final point come from the solve): hand it to the worker
      auto moments = std::make_shared<std::vector<FrameMoments>>(std::move(s.final_moments));
      auto Mp = std::make_shared<WindowParams>(std::move(M));
      marg_task_epoch_ = marg_epoch_;
      marg_worker_.submit([msys, moments, Mp, have_moments] {
        if (have_moments) msys->preset_moments = moments.get();
        return marginalize(*msys, *Mp);
      });
    } else {
      msys->lidar_eval = sys.lidar_eval; msys->lidar_launch = sys.lidar_launch; msys->lidar_wait = sys.lidar_wait;
      if (have_moments) msys->preset_moments = &s.final_moments;  // no second device pass at the same point
      last_marg_ = marginalize(*msys, M);
    }
    R.marginalized = 1;
    R.ms_marg = now_ms() - tm0;
  }
  // residual count of the last device evaluation (valid feature slots of frames 1..Wo)
  {
    double cnt = 0;
    for (int i = 1; i <= Wo_; ++i) cnt += h_moment_out_[size_t(i - 1) * LIO_MOMENT_OUT + 257];
    R.n_lidar_residuals = cfg_.point_distance_factor ? int(cnt) : 0;
  }
  R.ms_total = now_ms() - t_total0;
  if (g_debug_timing) {
    std::fprintf(stderr, "[lio_hip timing] total %.3f map %.3f feat %.3f opt %.3f marg %.3f | lidar_eval %d calls %.3f ms (%.1f us each)\n", R.ms_total,
                 R.ms_build_map, R.ms_features, R.ms_opt, R.ms_marg, dbg_eval_n_, dbg_eval_ms_, dbg_eval_n_ ? 1e3 * dbg_eval_ms_ / dbg_eval_n_ : 0.0);
  }
  if (g_debug_timing && res_passes_) {
    std::fprintf(stderr, "[lio_hip timing] resident moments: %d passes; folding block, from the doorbell copy seen (us): accumulated %.2f, parked %.2f, all flags in %.2f, sums posted %.2f; relay detect -> copy seen %.2f; host ring -> moments unpacked %.2f; HBM polls %.1f; %d worker blocks\n",
                 res_passes_, res_diag_us_[0] / res_passes_, res_diag_us_[1] / res_passes_, res_diag_us_[2] / res_passes_, res_diag_us_[3] / res_passes_,
                 res_relay_us_ / res_passes_, 1e3 * res_ring_to_done_ms_ / res_passes_, res_polls_ / res_passes_, res_bpf_ * res_nframes_);
    std::fprintf(stderr, "[lio_hip timing] resident moments: host ring -> relay's echo seen %.2f us (the host waits for it only under LIO_DEBUG_TIMING)\n",
                 1e3 * res_echo_ms_ / res_passes_);
    res_echo_ms_ = 0;
    res_diag_us_[0] = res_diag_us_[1] = res_diag_us_[2] = res_diag_us_[3] = res_polls_ = res_relay_us_ = res_ring_to_done_ms_ = 0; res_passes_ = 0;
  }
  if (g_debug_timing) std::fprintf(stderr, "[lio_hip timing] of which hipStreamSynchronize %.3f ms\n", dbg_sync_ms_);
  dbg_eval_ms_ = 0; dbg_eval_n_ = 0; dbg_sync_ms_ = 0;
  return true;
}

void Estimator::SlideWindow() {
  if (init_local_map_) {
    const int pivot = W_ - Wo_;
    const Rigidd lb = toDouble(transform_lb_);
    const Rigidd T_pivot = LidarPose(pivot, lb);
    const int i = pivot + 1;
    const Rigidd T_li = LidarPose(i, lb);
    const Rigidf tf = toFloat(compose(rinverse(T_li), T_pivot));
    const size_t drop = std::min(size_surf_stack_[0], stacks_[pivot].n);
    ConcatArgs ca{};
    ca.nseg = 2;
    ca.seg[0].src = stacks_[pivot].buf.p + drop; ca.seg[0].n = int(stacks_[pivot].n - drop); ca.seg[0].dst_off = 0;
    ca.seg[0].identity = 0; ca.seg[0].set_intensity = 0; ca.seg[0].intensity = 0; ca.seg[0].tf = affineOf(tf);
    ca.seg[1].src = stacks_[i].buf.p; ca.seg[1].n = int(stacks_[i].n); ca.seg[1].dst_off = ca.seg[0].n;
    ca.seg[1].identity = 1; ca.seg[1].set_intensity = 0; ca.seg[1].intensity = 0;
    ca.total = ca.seg[0].n + ca.seg[1].n;
    scratch_cloud_.buf.reserve(std::max(ca.total, 1));
    launch_transform_concat(ca, scratch_cloud_.buf.p, stream_);
    scratch_cloud_.n = size_t(ca.total);
    scratch_cloud_.id = ++g_content_id;
    // no host wait: the swap exchanges host-side handles only, and every reader or writer of either buffer (the next solve's
    // BuildLocalMap, PushFrame's recycling, Restore's copies, lio_est_get_stack) is enqueued on stream_ behind this kernel
    std::swap(stacks_[i], scratch_cloud_);
  }
  PushState(cir_buf_count_);
}

void Estimator::Snapshot() {
  JoinMarg();
  frames_dirty_ = true;   // (the next Restore copies everything once, then the containers equal the snapshot's)
  snap_.reset(new HostState{Ps_, Vs_, Bas_, Bgs_, Rs_, g_vec_, acc_last_, gyr_last_, transform_lb_, inited_, first_imu_, init_local_map_,
                            convergence_flag_, cir_buf_count_, all_laser_transforms_, n_state_, n_frames_, laser_odom_recv_count_,
                            extrinsic_stage_, last_event_, initial_time_, R_WI_, last_marg_, pre_integrations_,
                            tmp_pre_integration_ ? std::make_shared<Preintegration>(*tmp_pre_integration_) : nullptr, size_surf_stack_,
                            imu_stamped_});
  snap_stacks_.resize(stacks_.size());
  for (size_t i = 0; i < stacks_.size(); ++i) {
    snap_stacks_[i].buf.reserve(std::max<size_t>(stacks_[i].n, 1));
    if (stacks_[i].n)
      LIO_HIP(hipMemcpyAsync(snap_stacks_[i].buf.p, stacks_[i].buf.p, stacks_[i].n * sizeof(float4), hipMemcpyDeviceToDevice, stream_));
    snap_stacks_[i].n = stacks_[i].n;
    snap_stacks_[i].id = stacks_[i].id;
  }
  LIO_HIP(hipStreamSynchronize(stream_));
}

bool Estimator::CopySnapshotOf(Estimator &src) {
  if (!src.snap_ || src.W_ != W_ || src.Wo_ != Wo_ || &src == this) return false;
  JoinMarg();
  frames_dirty_ = true;
  snap_.reset(new HostState(*src.snap_));   // pre-integrations and the prior are immutable once pushed: shared
  snap_stacks_.resize(src.snap_stacks_.size());
  for (size_t i = 0; i < src.snap_stacks_.size(); ++i) {
    const DeviceCloud &c = src.snap_stacks_[i];
    snap_stacks_[i].buf.reserve(std::max<size_t>(c.n, 1), stream_);
    if (c.n) LIO_HIP(hipMemcpyAsync(snap_stacks_[i].buf.p, c.buf.p, c.n * sizeof(float4), hipMemcpyDeviceToDevice, stream_));
    snap_stacks_[i].n = c.n;
    snap_stacks_[i].id = ++g_content_id;
  }
  LIO_HIP(hipStreamSynchronize(stream_));
  return true;
}

bool Estimator::Restore() {
  if (!snap_) return false;
  ++marg_epoch_;  // a marginalization still in flight belongs to the state being discarded: its result is dropped at the next join
  const HostState &h = *snap_;
  Ps_ = h.Ps; Vs_ = h.Vs; Bas_ = h.Bas; Bgs_ = h.Bgs; Rs_ = h.Rs; g_vec_ = h.g_vec; acc_last_ = h.acc_last; gyr_last_ = h.gyr_last;
  transform_lb_ = h.transform_lb; inited_ = h.inited; first_imu_ = h.first_imu; init_local_map_ = h.init_local_map;
  convergence_flag_ = h.convergence_flag; cir_buf_count_ = h.cir_buf_count; last_marg_ = h.last_marg;
  n_state_ = h.n_state; n_frames_ = h.n_frames; laser_odom_recv_count_ = h.laser_odom_recv_count;
  extrinsic_stage_ = h.extrinsic_stage; last_event_ = h.last_event; initial_time_ = h.initial_time; R_WI_ = h.R_WI;
  if (frames_dirty_) {
    pre_integrations_ = h.pre_integrations; all_laser_transforms_ = h.all_laser_transforms;
    tmp_pre_integration_ = h.tmp_pre_integration ? std::make_shared<Preintegration>(*h.tmp_pre_integration) : nullptr;
    size_surf_stack_ = h.size_surf_stack; imu_stamped_ = h.imu_stamped;
    frames_dirty_ = false;
  }
  for (size_t i = 0; i < stacks_.size(); ++i) {
    if (stacks_[i].id == snap_stacks_[i].id && stacks_[i].n == snap_stacks_[i].n) continue;  // untouched since the snapshot
    stacks_[i].id = snap_stacks_[i].id;
    stacks_[i].buf.reserve(std::max<size_t>(snap_stacks_[i].n, 1));
    if (snap_stacks_[i].n)
      LIO_HIP(hipMemcpyAsync(stacks_[i].buf.p, snap_stacks_[i].buf.p, snap_stacks_[i].n * sizeof(float4), hipMemcpyDeviceToDevice, stream_));
    stacks_[i].n = snap_stacks_[i].n;
  }
  // no host wait: every consumer of the stacks is ordered behind these copies on stream_ (or behind an event recorded on it)
  return true;
}

// ================================================================================================
// Batched solve: the per-window host halves (est_batch.hip drives them)
// ================================================================================================
bool Estimator::BatchEligible() const {
  if (!inited_ || cir_buf_count_ < W_ || false) return false;
  if (!cfg_.imu_factor || !cfg_.point_distance_factor || Sharded() || rccl_comm_) return false;
  if (Wo_ < 1 || Wo_ > DS_MAX_WO || Wo_ > LIO_BW_MAX_STATIC + 1 || Wo_ > LIO_BW_MAX_SEG) return false;
  const int dim = 15 * (Wo_ + 1) + 6;
  if ((dim + DS_NB - 1) / DS_NB * DS_NB > DS_MAX_NPAD || 6 * Wo_ + 15 > MARG_MAX_N) return false;
  return true;
}

void Estimator::BatchDescribe(BatchWin &bw) {
  FusePivotOnce();
  const int pivot = W_ - Wo_;
  const Rigidd lb = toDouble(transform_lb_);
  const Rigidd T_pivot = LidarPose(pivot, lb);
  std::memset(&bw, 0, sizeof(bw));
  bw.inv_leaf = 1.0f / cfg_.surf_filter_size;
  bw.min_match_sq_dis = cfg_.min_match_sq_dis; bw.min_plane_dis = cfg_.min_plane_dis;
  const int keep_mult = cfg_.keep_features ? 10 : 1;
  bw.keep = cfg_.keep_features ? 1 : 0;
  int total = 0;
  total_slots_ = 0;
  for (int i = 0; i <= W_; ++i) { slot_off_[i] = 0; nslots_[i] = 0; }
  for (int i = pivot; i <= W_; ++i) {
    const Rigidf tf = RelTransform(i, T_pivot, lb);
    const Rigidf lt = fromAffine(linearOf(tf), tf.pos);
    if (i < W_) {   // a segment of the local map (Estimator.cc:1480-1507)
      BwSeg &sg = bw.seg[bw.nseg++];
      sg.src = stacks_[i].buf.p; sg.n = int(stacks_[i].n); sg.dst_off = total;
      if (i == pivot) { sg.identity = 1; sg.set_intensity = 0; sg.intensity = 0; }
      else { sg.identity = 0; sg.set_intensity = 1; sg.intensity = float(i); sg.tf = affineOf(tf); }
      total += sg.n;
    }
    if (i == pivot) continue;
    const int k = (i == W_) ? LIO_BW_MAX_STATIC : bw.nstatic;
    float *o = bw.tf[k];
    o[0] = lt.rot.x; o[1] = lt.rot.y; o[2] = lt.rot.z; o[3] = lt.rot.w; o[4] = lt.pos.x; o[5] = lt.pos.y; o[6] = lt.pos.z; o[7] = 0.f;
    slot_off_[i] = int(total_slots_);
    nslots_[i] = int(stacks_[i].n) * ((i == W_) ? keep_mult : 1);
    total_slots_ += size_t(nslots_[i]);
    FeatFrame &f = (i == W_) ? bw.newest : bw.fr[bw.nstatic];
    f.stack = stacks_[i].buf.p; f.M = int(stacks_[i].n); f.slot_off = slot_off_[i]; f.tf_index = k;
    if (i < W_) ++bw.nstatic;
  }
  bw.n_local = total;
  bw.n_slots = int(total_slots_);
  bw.nb_round = bw.newest.M > 0 ? bw_round_blocks(bw.newest.M) : 0;
}

void Estimator::BatchSetOdom(const OdomState &st) {
  laser_odom_iters_ = st.iters;
  laser_odom_kz_ = st.degenerate ? st.kz : 0;
  laser_odom_transform_ = Rigidf(Quat<float>(st.T[3], st.T[0], st.T[1], st.T[2]), Vec3<float>(st.T[4], st.T[5], st.T[6]));
  if (cfg_.keep_features) nslots_[W_] = int(stacks_[W_].n) * std::max(1, st.iters);
}

bool Estimator::BatchPackProblem(int bpf, DevProblem &pb, DevState &st, std::shared_ptr<MargPrior> *prior) {
  const int pivot = W_ - Wo_;
  WindowParams P;
  VectorToParams(P);
  P.ex_constant = (cfg_.extrinsic_stage == 0 || !cfg_.opt_extrinsic);
  WindowSystem sys;
  sys.Wo = Wo_;
  sys.use_lidar = cfg_.point_distance_factor;
  sys.pim.assign(Wo_, nullptr);
  for (int i = 0; i < Wo_; ++i) {
    auto &pi = pre_integrations_[pivot + i + 1];
    if (pi && pi->sum_dt <= 10.0) sys.pim[i] = pi;
  }
  if (cfg_.marginalization_factor && last_marg_) sys.prior = last_marg_;
  if (cfg_.prior_factor) {
    sys.use_prior_factor = true;
    Rigidd t = toDouble(transform_lb_);
    sys.prior_pos = t.pos; sys.prior_rot = t.rot;
  }
  if (prior) *prior = sys.prior;
  if (!ds_pack_problem(sys, P, cfg_.max_num_iterations, bpf, convergence_flag_, cfg_.imu_factor, pb, nullptr)) return false;
  if (ds_lds_doubles(pb.n_pad, Wo_) * sizeof(double) > 160 * 1024) return false;
  ds_init_state(P, st);
  return true;
}

bool Estimator::BatchFinish(const DevState &st, const std::shared_ptr<MargPrior> &prior_used, lio_solve_report &R, DevMarg &mg,
                            std::shared_ptr<MargPrior> *shell) {
  const int pivot = W_ - Wo_;
  R.cost_marg_before = st.costs0[0]; R.cost_pim_before = st.costs0[1]; R.cost_ppp_before = st.costs0[2];
  const bool turn_off = st.turn_off != 0;
  convergence_flag_ = st.conv_flag_out != 0;
  std::shared_ptr<MargPrior> prior = prior_used;
  if (!convergence_flag_) { last_marg_.reset(); prior.reset(); }   // Estimator.cc:1962-1975 (the device loop only runs when that changes nothing)
  WindowParams P;
  VectorToParams(P);   // sizes; every entry is overwritten
  ds_unpack_params(st.x, P);
  R.iterations = st.it; R.successful_steps = st.successful; R.termination = st.termination;
  R.initial_cost = st.ntrace > 0 ? st.trace[0] : 0.0; R.final_cost = st.x_cost;
  for (int k = 0; k < st.ntrace && k < 32; ++k) R.cost_trace[k] = st.trace[k];
  ParamsToVector(P);
  R.turn_off = turn_off; R.convergence_flag = convergence_flag_;
  R.n_lidar_residuals = int(st.n_lidar);
  R.laser_odom_iterations = laser_odom_iters_; R.laser_odom_kz = laser_odom_kz_;
  std::memset(&mg, 0, sizeof(mg));
  if (!cfg_.marginalization_factor || turn_off) return false;
  // MarginalizationInfo::{AddResidualBlockInfo, PreMarginalize, Marginalize} (host_solver.h: marginalize): the layout
  WindowParams M;
  VectorToParams(M);
  ds_pack_params(M, mg.x);
  auto &pi = pre_integrations_[pivot + 1];
  const bool has_imu = pi && pi->sum_dt < 10.0 && pi->sqrt_info() != nullptr;
  const bool sb0_present = has_imu || prior != nullptr;
  mg.active = 1; mg.Wo = Wo_; mg.has_imu = has_imu ? 1 : 0; mg.have_prior = prior ? 1 : 0;
  for (int i = 0; i <= DS_MAX_WO; ++i) mg.pose_col[i] = -1;
  mg.sb_col[0] = mg.sb_col[1] = -1;
  int pos = 0;
  mg.pose_col[0] = pos; pos += 6;
  if (sb0_present) { mg.sb_col[0] = pos; pos += 9; }
  const int m = pos;
  std::vector<KeepBlock> keep;
  mg.pose_col[1] = pos; keep.push_back({0, 0, 7, pos - m}); pos += 6;
  if (has_imu) { mg.sb_col[1] = pos; keep.push_back({1, 0, 9, pos - m}); pos += 9; }
  for (int i = 2; i <= Wo_; ++i) { mg.pose_col[i] = pos; keep.push_back({0, i - 1, 7, pos - m}); pos += 6; }
  mg.ex_col = pos; keep.push_back({2, 0, 7, pos - m}); pos += 6;
  mg.m = m; mg.n = pos - m;
  for (int i = 0; i < DS_MAX_NPAD; ++i) mg.prior_col[i] = -1;
  if (prior) {
    for (const KeepBlock &kb : prior->keep) {
      const int col = kb.kind == 0 ? mg.pose_col[kb.index] : (kb.kind == 1 ? (kb.index < 2 ? mg.sb_col[kb.index] : -1) : mg.ex_col);
      if (col < 0) continue;
      const int la = kb.size == 7 ? 6 : kb.size;
      for (int i = 0; i < la; ++i) mg.prior_col[col + i] = kb.idx + i;
    }
  }
  auto pr = std::make_shared<MargPrior>();
  pr->n = mg.n; pr->keep = keep;
  for (const KeepBlock &kb : keep) {
    const double *src = kb.kind == 0 ? M.pose[kb.index + 1].data() : (kb.kind == 1 ? M.sb[kb.index + 1].data() : M.ex.data());
    pr->x0.emplace_back(src, src + kb.size);
  }
  if (shell) *shell = pr;
  R.marginalized = 1;
  return true;
}

}  // namespace lio
