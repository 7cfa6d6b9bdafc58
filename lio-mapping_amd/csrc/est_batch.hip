// est_batch.hip — host orchestration of the batched solve (see est_batch.h).  The kernels: batch_kernels.hip (map / feature stages),
// solve_kernels.hip (launch A / launch B of the trust-region loop), marg_kernels.hip (marginalization).
#include "est_batch.h"

#include <chrono>
#include <exception>
#include <string>
#include <thread>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace lio {

static double bnow_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static int round_up(int v, int m) { return (v + m - 1) / m * m; }
template <typename T> static void pinned(T *&p, size_t n) { LIO_HIP(hipHostMalloc(reinterpret_cast<void **>(&p), sizeof(T) * std::max<size_t>(n, 1))); std::memset(static_cast<void *>(p), 0, sizeof(T) * std::max<size_t>(n, 1)); }

EstimatorBatch::EstimatorBatch(const std::vector<Estimator *> &members) : m_(members), knobs_(batch_knobs_from_env()) {
  if (m_.empty()) throw std::runtime_error("EstimatorBatch: no windows");
  for (Estimator *e : m_) if (!e) throw std::runtime_error("EstimatorBatch: null window");
  const size_t B = m_.size();
  // one device per batch: the members' buffers, this batch's streams and the kernels' per-device attributes (hipFuncSetAttribute
  // applies to the current device only) all belong to the device the windows were created on
  device_id_ = m_[0]->device_id_;
  for (Estimator *e : m_) if (e->device_id_ != device_id_) throw std::runtime_error("EstimatorBatch: windows of different devices");
  LIO_HIP(hipSetDevice(device_id_));
  prepare_bw_step_kernel();
  prepare_bw_marg_kernel();
  ok_.assign(B, 0);
  LIO_HIP(hipStreamCreate(&stream_));
  for (hipEvent_t &e : ev_) LIO_HIP(hipEventCreate(&e));
  for (hipEvent_t &e : ev_wait_) LIO_HIP(hipEventCreate(&e));
  for (hipStream_t &g : stream_grp_) LIO_HIP(hipStreamCreate(&g));
  for (hipStream_t &g : stream_aux_) LIO_HIP(hipStreamCreate(&g));
  LIO_HIP(hipStreamCreate(&stream_marg_));
  LIO_HIP(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
  LIO_HIP(hipEventCreateWithFlags(&ev_marg_, hipEventDisableTiming));
  for (hipEvent_t &e : ev_grp_) LIO_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (hipEvent_t &e : ev_aux_) LIO_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (hipEvent_t &e : ev_step_) LIO_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  win_.resize(B);
  for (size_t w = 0; w < B; ++w) { win_[w].e = m_[w]; m_[w]->AdoptStream(stream_); }
  pinned(h_win_, B); pinned(h_grid_, B); pinned(h_vout_, B); pinned(h_odom_, B); pinned(h_bs_, B); pinned(h_pb_, B); pinned(h_st_, B); pinned(h_mg_, B);
  pinned(h_prior_, B * ds_prior_mats_size(MARG_MAX_N)); pinned(h_nconv_, 4); pinned(h_seg_, 2 * B);
  d_win_.reserve(B); d_grid_.reserve(B); d_vout_.reserve(B); d_odom_.reserve(B); d_bs_.reserve(B); d_pb_.reserve(B); d_st_.reserve(B); d_mg_.reserve(B);
  range_overflow_.reserve(B); vparams_.reserve(B); nconv_.reserve(4); d_seg_.reserve(2 * B); d_layout_.reserve(B);
  LIO_HIP(hipMemsetAsync(range_overflow_.p, 0, sizeof(int) * range_overflow_.cap, stream_));
  LIO_HIP(hipMemsetAsync(d_mg_.p, 0, sizeof(DevMarg) * d_mg_.cap, stream_));
  // the scratch slab of a window (doubles)
  size_t o = 0;
  auto take = [&](size_t n) { const size_t at = o; o += (n + 7) & ~size_t(7); return at; };
  lay_.prior[0] = take(ds_prior_mats_size(MARG_MAX_N)); lay_.prior[1] = take(ds_prior_mats_size(MARG_MAX_N));
  lay_.imu = take(size_t(DS_MAX_WO) * DS_IMU_OUT); lay_.lmap = take(size_t(DS_MAX_WO) * DS_LMAP_OUT);
  lay_.prior_out = take(MARG_MAX_N + 8); lay_.exprior = take(DS_EXP_OUT);
  lay_.Hcur = take(size_t(DS_MAX_NPAD) * (DS_MAX_NPAD + 1)); lay_.Sbuf = take(size_t(2) * DS_MAX_WO * LIO_MOMENT_OUT); lay_.prof = take(96);
  lay_.marg_imu = take(DS_IMU_OUT); lay_.marg_lmap = take(size_t(DS_MAX_WO) * DS_LMAP_OUT); lay_.marg_prior_out = take(MARG_MAX_N + 8);
  { const size_t N = MARG_MAX_M + MARG_MAX_N; lay_.marg_A = take(N * N + N); }
  lay_.marg_info = take(MARG_MAX_N + 8 + 16);   // sweeps (2) | eigenvalues | 8 phase stamps | 4 per-phase sums of the eigensolver (LIO_MARG_PROF builds)
  lay_.total = o;
  slab_.reserve(B * lay_.total);
  LIO_HIP(hipMemsetAsync(slab_.p, 0, sizeof(double) * slab_.cap, stream_));
  LIO_HIP(hipStreamSynchronize(stream_));
}

EstimatorBatch::~EstimatorBatch() {
  try {
    if (stream_) (void)hipStreamSynchronize(stream_);
    if (stream_marg_) (void)hipStreamSynchronize(stream_marg_);
    for (size_t w = 0; w < win_.size(); ++w)
      for (int k = 0; k < 2; ++k) if (win_[w].dev_prior[k]) win_[w].dev_prior[k]->materialize();   // nobody may be left holding a shell
    for (Estimator *e : m_) { e->solve_hook_ = nullptr; e->ReleaseAdoptedStream(); }
  } catch (...) {}
  for (void *p : {static_cast<void *>(h_win_), static_cast<void *>(h_grid_), static_cast<void *>(h_vout_), static_cast<void *>(h_odom_), static_cast<void *>(h_bs_),
                  static_cast<void *>(h_pb_), static_cast<void *>(h_st_), static_cast<void *>(h_mg_), static_cast<void *>(h_prior_), static_cast<void *>(h_nconv_),
                  static_cast<void *>(h_seg_)})
    if (p) (void)hipHostFree(p);
  for (hipEvent_t e : ev_) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : ev_wait_) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : ev_k_) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : ev_grp_) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : ev_aux_) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : ev_step_) if (e) (void)hipEventDestroy(e);
  if (ev_fork_) (void)hipEventDestroy(ev_fork_);
  if (ev_marg_) (void)hipEventDestroy(ev_marg_);
  for (hipStream_t g : stream_grp_) if (g) (void)hipStreamDestroy(g);
  for (hipStream_t g : stream_aux_) if (g) (void)hipStreamDestroy(g);
  if (stream_marg_) (void)hipStreamDestroy(stream_marg_);
  if (stream_) (void)hipStreamDestroy(stream_);
}

bool EstimatorBatch::SetOption(const char *name, int v) {
  if (!name) return false;
  const std::string n(name);
  if (n == "lanes_per_query") { if (v != 0 && v != 1 && v != 2 && v != 4 && v != 8) return false; knobs_.lanes_per_query = v; }
  else if (n == "occupancy") { if (v != -1 && v != 0 && v != 6 && v != 8) return false; knobs_.occupancy = v; }
  else if (n == "loop_groups") { if (v < 0 || v > kGroups) return false; knobs_.loop_groups = v; }
  else if (n == "aux_threads") { if (v != 0 && v != 64 && v != 128 && v != 256) return false; knobs_.aux_threads = v; }
  else if (n == "aux_stream") { if (v != 0 && v != 1) return false; knobs_.aux_stream = v; }
  else if (n == "finish_threads") { if (v < 0 || v > 8) return false; knobs_.finish_threads = v; }
  else if (n == "time_kernels") { if (v != 0 && v != 1) return false; knobs_.time_kernels = v; }
  else return false;
  return true;
}

const BatchClock &EstimatorBatch::clock() {
  if (ev_valid_) {
    LIO_HIP(hipStreamSynchronize(stream_));
    LIO_HIP(hipStreamSynchronize(stream_marg_));
    for (int k = 0; k < 6; ++k) {
      float ms = 0;
      clk_.dev[k] = hipEventElapsedTime(&ms, ev_[k], ev_[k + 1]) == hipSuccess ? double(ms) : 0.0;
    }
    clk_.dev_marg_wait = 0;
    if (ev_wait_valid_) {   // the rounds' stage minus the wait for the previous marginalization: what the rounds themselves took
      float ms = 0;
      clk_.dev_marg_wait = hipEventElapsedTime(&ms, ev_wait_[0], ev_wait_[1]) == hipSuccess ? double(ms) : 0.0;
      clk_.dev[3] = std::max(0.0, clk_.dev[3] - clk_.dev_marg_wait);
    }
    for (int k = 0; k < 3; ++k) { clk_.kernel_ms[k] = 0; clk_.kernel_launches[k] = 0; }
    for (int q = 0; q + 3 < ev_k_used_; q += 4)
      for (int k = 0; k < 3; ++k) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ev_k_[size_t(q + k)], ev_k_[size_t(q + k + 1)]) == hipSuccess) { clk_.kernel_ms[k] += double(ms); ++clk_.kernel_launches[k]; }
      }
    ev_k_used_ = 0;
    ev_valid_ = false; ev_wait_valid_ = false;
  }
  return clk_;
}

void EstimatorBatch::Sync() {
  LIO_HIP(hipStreamSynchronize(stream_));
  LIO_HIP(hipStreamSynchronize(stream_marg_));
}

// the matrices of the prior held in device buffer `buf` of window w -> pr (the fetch of a MargPrior shell)
void EstimatorBatch::FetchPrior(int w, int buf, MargPrior &pr) {
  const size_t n = size_t(pr.n);
  std::vector<double> h(ds_prior_mats_size(pr.n));
  Sync();
  LIO_HIP(hipMemcpy(h.data(), slab_.p + size_t(w) * lay_.total + lay_.prior[buf], h.size() * sizeof(double), hipMemcpyDeviceToHost));
  pr.JtJ = DMat(pr.n, pr.n); pr.lin_jac = DMat(pr.n, pr.n); pr.lin_res.assign(n, 0.0); pr.Jtr0.assign(n, 0.0);
  std::memcpy(pr.JtJ.a.data(), h.data(), sizeof(double) * n * n);
  std::memcpy(pr.lin_jac.a.data(), h.data() + n * n, sizeof(double) * n * n);
  std::memcpy(pr.lin_res.data(), h.data() + 2 * n * n, sizeof(double) * n);
  std::memcpy(pr.Jtr0.data(), h.data() + 2 * n * n + n, sizeof(double) * n);
}
// buffer `buf` of window w is about to be overwritten: whoever else still holds its prior as a shell gets the matrices first
void EstimatorBatch::Materialize(int w, int buf) {
  std::shared_ptr<MargPrior> &p = win_[w].dev_prior[buf];
  if (p && p->on_device && p.use_count() > 1) p->materialize();
  p.reset();
}

// ---- test hook: per-window digests of a stage's device arrays (est_batch.h)
static unsigned long long fnv1a(const void *p, size_t n, unsigned long long h = 1469598103934665603ull) {
  const unsigned char *b = static_cast<const unsigned char *>(p);
  for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}
static unsigned long long mix64(unsigned long long x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}
template <typename T> static std::vector<T> fetch(const T *d, size_t n) {
  std::vector<T> h(std::max<size_t>(n, 1));
  if (n) LIO_HIP(hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost));
  return h;
}
void EstimatorBatch::StageDigest(int stage, unsigned long long *out) {
  Sync();
  for (hipStream_t g : stream_grp_) LIO_HIP(hipStreamSynchronize(g));
  const int B = size();
  for (int w = 0; w < B; ++w) {
    const BatchWin &bw = h_win_[w];
    const BatchGrid &G = h_grid_[w];
    unsigned long long h = 0;
    switch (stage) {
      case 0: {
        const std::vector<float4> p = fetch(filtered_all_.p + bw.loc_off, size_t(G.n_filtered));
        h = fnv1a(p.data(), size_t(G.n_filtered) * sizeof(float4), mix64(G.n_filtered));
        break;
      }
      case 1: {
        size_t ncells = size_t(G.g.dims[0]) * G.g.dims[1] * G.g.dims[2];
        const std::vector<int> c = fetch(cells_all_.p + G.cell_off, ncells + 1);
        const std::vector<float4> p = fetch(sorted_all_.p + c[0], size_t(G.n_filtered));
        h = mix64(ncells);
        for (size_t k = 0; k < ncells; ++k) {
          unsigned long long run = 0;   // a cell's points as a multiset
          for (int j = c[k] - c[0]; j < c[k + 1] - c[0]; ++j) run += mix64(fnv1a(&p[size_t(j)], sizeof(float4)));
          const int rel[2] = {c[k] - c[0], c[k + 1] - c[0]};
          if (rel[1] != rel[0]) h = mix64(h ^ fnv1a(rel, sizeof(rel)) ^ run ^ mix64(k));
        }
        break;
      }
      case 2: {
        const std::vector<uint8_t> v = fetch(valid_all_.p + bw.slot_base, size_t(bw.n_slots));
        h = fnv1a(v.data(), size_t(bw.n_slots), mix64(bw.n_slots));
        break;
      }
      case 3: {
        const std::vector<uint8_t> v = fetch(valid_all_.p + bw.slot_base, size_t(bw.n_slots));
        const std::vector<float4> c = fetch(coef_all_.p + bw.slot_base, size_t(bw.n_slots));
        h = mix64(bw.n_slots);
        for (int k = 0; k < bw.n_slots; ++k) if (v[size_t(k)]) h = fnv1a(&c[size_t(k)], sizeof(float4), h ^ mix64(k));
        break;
      }
      case 4: { const std::vector<OdomState> o = fetch(d_odom_.p + w, 1); h = fnv1a(o.data(), sizeof(OdomState)); break; }
      case 5: { const std::vector<DevState> o = fetch(d_st_.p + w, 1); h = fnv1a(o.data(), sizeof(DevState)); break; }
      case 8: {   // scaled H at the accepted point (upper triangle), + on stderr whether it is positive definite
        const DevProblem &pb = h_pb_[w];
        const int n = pb.n, ld = pb.ld;
        const std::vector<double> Hc = fetch(slab_.p + size_t(w) * lay_.total + lay_.Hcur, size_t(pb.n_pad) * ld);
        h = mix64(n);
        std::vector<double> M(size_t(n) * n);
        for (int r = 0; r < n; ++r) { h = fnv1a(&Hc[size_t(r) * ld + r], sizeof(double) * (n - r), h); for (int c = r; c < n; ++c) M[size_t(r) * n + c] = M[size_t(c) * n + r] = Hc[size_t(r) * ld + c]; }
        if (std::getenv("LIO_DEBUG_DIGEST")) {
          int bad = -1; double dmin = 1e300, dmax = 0; bool finite = true;
          for (double v : M) if (!std::isfinite(v)) finite = false;
          for (int j = 0; j < n && bad < 0; ++j) {   // plain Cholesky of H + 1e-8 diag(H)
            double d = M[size_t(j) * n + j] * (1.0 + 1e-8);
            for (int k = 0; k < j; ++k) d -= M[size_t(j) * n + k] * M[size_t(j) * n + k];
            if (!(d > 0)) { bad = j; break; }
            dmin = std::min(dmin, d); dmax = std::max(dmax, d);
            const double l = std::sqrt(d);
            M[size_t(j) * n + j] = l;
            for (int r = j + 1; r < n; ++r) { double v = M[size_t(r) * n + j]; for (int k = 0; k < j; ++k) v -= M[size_t(r) * n + k] * M[size_t(j) * n + k]; M[size_t(r) * n + j] = v / l; }
          }
          std::fprintf(stderr, "[digest] window %d: H_cur %016llx finite %d cholesky %s (pivot %d) pivots in [%.3e, %.3e]\n", w, h, int(finite), bad < 0 ? "ok" : "FAILS", bad, dmin, dmax);
        }
        break;
      }
      case 9: {   // marginalization: sweeps and phase stamps of the last launch (stderr under LIO_DEBUG_DIGEST); digest of the new prior
        const std::vector<double> mi = fetch(slab_.p + size_t(w) * lay_.total + lay_.marg_info, MARG_MAX_N + 24);
        const std::vector<double> pm = fetch(slab_.p + size_t(w) * lay_.total + lay_.prior[1 - win_[w].cur], ds_prior_mats_size(h_mg_[w].n));
        h = fnv1a(pm.data(), pm.size() * sizeof(double), mix64(h_mg_[w].n));
        if (std::getenv("LIO_DEBUG_DIGEST") && (w == 0 || w == B - 1))
          std::fprintf(stderr, "[digest] window %d marginalization: m %d n %d QL sweeps %g / %g; shader clocks: assembly %.0f, Amm eig %.0f, pinv + T + S %.0f, S eig %.0f, factors out %.0f, J^T J %.0f\n",
                       w, h_mg_[w].m, h_mg_[w].n, mi[0], mi[1], mi[MARG_MAX_N + 9] - mi[MARG_MAX_N + 8], mi[MARG_MAX_N + 10] - mi[MARG_MAX_N + 9], mi[MARG_MAX_N + 11] - mi[MARG_MAX_N + 10], mi[MARG_MAX_N + 12] - mi[MARG_MAX_N + 11],
                       mi[MARG_MAX_N + 13] - mi[MARG_MAX_N + 12], mi[MARG_MAX_N + 14] - mi[MARG_MAX_N + 13]);
        if (std::getenv("LIO_DEBUG_DIGEST") && w == 0 && mi[MARG_MAX_N + 16] != 0.0)
          std::fprintf(stderr, "[digest] S eig, thread 0: reduction %.0f, QL %.0f of which wave 0's recurrence %.0f, its waits at the sweeps' barriers %.0f\n", mi[MARG_MAX_N + 16], mi[MARG_MAX_N + 19], mi[MARG_MAX_N + 17], mi[MARG_MAX_N + 18]);
        break;
      }
      case 7: { const std::vector<DevState> o = fetch(d_st_.p + w, 1); h = fnv1a(o[0].scale, sizeof(o[0].scale)); break; }
      case 6: {
        const Win &Wn = win_[w];
        const size_t n = Wn.device ? size_t(Wn.e->Wo_) * Wn.bpf * LIO_MOMENT_OUT : 0;
        const std::vector<double> o = fetch(partials_.p + Wn.part_off, n);
        h = mix64(n);
        for (size_t r = 0; r + LIO_MOMENT_OUT <= n; r += LIO_MOMENT_OUT) h = fnv1a(o.data() + r, 258 * sizeof(double), h);   // (the last two words of a row are never written)
        break;
      }
      default: throw std::runtime_error("EstimatorBatch::StageDigest: unknown stage");
    }
    out[w] = h;
  }
}

int EstimatorBatch::Solve(lio_solve_report *reps) {
  const double t0 = bnow_ms();
  const int B = size();
  hipStream_t s = stream_;
  std::vector<lio_solve_report> local;
  if (!reps) { local.resize(B); reps = local.data(); }
  std::memset(static_cast<void *>(reps), 0, sizeof(lio_solve_report) * B);
  clk_ = BatchClock();
  ev_valid_ = false;
  ev_k_used_ = 0;
  // ------------------------------------------------------------------------------------------------ describe
  int off = 0, slot = 0, part_rows = 0, max_cap = 0, max_slots = 0, max_M = 0, max_static = 0, max_nb = 0;
  long long q_static = 0, q_newest = 0;
  for (int w = 0; w < B; ++w) {
    Win &Wn = win_[w];
    Estimator *e = Wn.e;
    BatchWin &bw = h_win_[w];
    Wn.device = e->BatchEligible();
    Wn.prior_used.reset();
    if (Wn.device) {
      e->JoinMarg(false);
      e->BatchDescribe(bw);
    } else {
      std::memset(&bw, 0, sizeof(bw));
      bw.inv_leaf = 1.f;
    }
    bw.loc_off = off; bw.loc_cap = round_up(std::max(bw.n_local, 1), 256); off += bw.loc_cap;
    bw.slot_base = slot; slot += round_up(bw.n_slots, 16);
    for (int k = 0; k < bw.nstatic; ++k) { bw.fr[k].slot_off += bw.slot_base; max_M = std::max(max_M, bw.fr[k].M); q_static += bw.fr[k].M; }
    bw.newest.slot_off += bw.slot_base; q_newest += bw.newest.M;
    bw.part_off = part_rows; part_rows += bw.nb_round;
    max_cap = std::max(max_cap, bw.loc_cap); max_slots = std::max(max_slots, bw.n_slots); max_static = std::max(max_static, bw.nstatic);
    max_nb = std::max(max_nb, bw.nb_round);
  }
  const size_t N = size_t(off);
  if (N > size_t(INT_MAX) / 2) throw std::runtime_error("EstimatorBatch: the batch's local maps exceed 2^30 points");
  local_all_.reserve(N, s); filtered_all_.reserve(N, s); sorted_all_.reserve(N, s);
  keys_.reserve(N, s); keysb_.reserve(N, s); vals_.reserve(N, s); valsb_.reserve(N, s); ckeys_.reserve(N, s);
  bounds_partial_.reserve(N / 256 * 8, s); tile_heads_.reserve(N / 256, s);
  valid_all_.reserve(std::max(slot, 16), s); coef_all_.reserve(std::max(slot, 16), s); score_all_.reserve(std::max(slot, 16), s);
  odom_partials_.reserve(size_t(std::max(part_rows, 1)) * 28, s);
  const double t1 = bnow_ms();
  clk_.describe = t1 - t0;
  // ------------------------------------------------------------------------------------------------ BuildLocalMap
  LIO_HIP(hipEventRecord(ev_[0], s));
  LIO_HIP(hipMemcpyAsync(d_win_.p, h_win_, sizeof(BatchWin) * B, hipMemcpyHostToDevice, s));
  LIO_HIP(hipMemsetAsync(nconv_.p, 0, sizeof(int), s));
  launch_bw_setup(d_win_.p, B, max_slots, valid_all_.p, d_odom_.p, nconv_.p, s);
  // The filter's order: 9-bit passes over the keys relative to each window's bounds (pass 0 converts the stored absolute keys on the fly
  // and numbers the values); every window sorts its own range.  Three passes order 27 bits — 26 of key and "no point" above them — which
  // is what a 200 m x 200 m x 25 m map at a 0.4 m leaf takes; the bounds are only known on the device, so the host goes by what each
  // window's keys took in its previous solve (4 passes while unknown).  A window that outgrows the guess is flagged by
  // the layout kernel and re-done by the single-window path.
  int vox_passes = 3;
  for (int w = 0; w < B; ++w) if (win_[w].device && (win_[w].key_bits == 0 || win_[w].key_bits > 26)) vox_passes = 4;
  launch_bw_concat_keys(d_win_.p, B, max_cap, local_all_.p, ckeys_.p, bounds_partial_.p, vparams_.p, d_layout_.p, range_overflow_.p,
                        std::min(31, SS_MAX_BITS * vox_passes - 1), s);
  const uint32_t *keys_sorted = nullptr, *vals_sorted = nullptr;
  {
    SegDesc *seg = h_seg_;
    for (int w = 0; w < B; ++w) seg[w] = SegDesc{h_win_[w].loc_off, h_win_[w].n_local, 0};
    const SegSortPlan plan = seg_sort_plan(seg, B, SS_MAX_BITS);
    sort_hist_.reserve(std::max<size_t>(plan.hist_entries, 1), s);
    LIO_HIP(hipMemcpyAsync(d_seg_.p, seg, sizeof(SegDesc) * B, hipMemcpyHostToDevice, s));
    const uint32_t *ki = ckeys_.p, *vi = nullptr;
    uint32_t *ko = keys_.p, *vo = vals_.p, *ko2 = keysb_.p, *vo2 = valsb_.p;
    for (int p = 0; p < vox_passes; ++p) {
      const int shift = p * SS_MAX_BITS, bits = std::min(SS_MAX_BITS, 32 - shift);
      seg_sort_pass(d_seg_.p, B, plan, ki, vi, ko, vo, sort_hist_.p, shift, bits, p == 0 ? d_layout_.p : nullptr, s);
      ki = ko; vi = vo;
      std::swap(ko, ko2); std::swap(vo, vo2);
    }
    keys_sorted = ki; vals_sorted = vi;
  }
  launch_bw_vox_finish(d_win_.p, B, max_cap, local_all_.p, keys_sorted, vals_sorted, tile_heads_.p, filtered_all_.p, vparams_.p, range_overflow_.p, d_vout_.p, s);
  LIO_HIP(hipMemcpyAsync(h_vout_, d_vout_.p, sizeof(BwVoxOut) * B, hipMemcpyDeviceToHost, s));
  LIO_HIP(hipEventRecord(ev_[1], s));
  LIO_HIP(hipStreamSynchronize(s));
  const double t2 = bnow_ms();
  clk_.map = t2 - t1;
  // ------------------------------------------------------------------------------------------------ K-NN grids, features, rounds
  size_t cell_total = 0;
  int max_filtered = 0;
  for (int w = 0; w < B; ++w) {
    Win &Wn = win_[w];
    BatchGrid &G = h_grid_[w];
    std::memset(&G, 0, sizeof(G));
    const BwVoxOut &vo = h_vout_[w];
    size_t ncells = 1;
    G.g.dims[0] = G.g.dims[1] = G.g.dims[2] = 1;
    G.g.inv_cell = 1.f;
    {   // what the window's relative keys took: the next solve's guess for the number of sort passes
      int kb = 0;
      for (int d = 0; d < 3; ++d) { int b = 0; while ((1 << b) < vo.params.divb[d]) ++b; kb += b; }
      Wn.key_bits = vo.params.n_valid > 0 ? std::max(kb, 1) : Wn.key_bits;
    }
    if (Wn.device && (vo.params.overflow || vo.range_overflow)) Wn.device = false;   // PCL's own index / "leaf too small" / keys beyond the passes: the single-window path has those forms
    if (Wn.device && vo.count > 0) {
      const float cell = std::sqrt(Wn.e->cfg_.min_match_sq_dis) * 1.0001f + 1e-6f;
      G.g.inv_cell = 1.0f / cell;
      for (int d = 0; d < 3; ++d) {
        const int lo = int(std::floor(vo.params.mn[d] * G.g.inv_cell)) - 1, hi = int(std::floor(vo.params.mx[d] * G.g.inv_cell)) + 1;
        G.g.origin[d] = lo; G.g.dims[d] = hi - lo + 1;
        ncells *= size_t(G.g.dims[d]);
      }
      if (ncells > (size_t(1) << 28)) { Wn.device = false; ncells = 1; G.g.dims[0] = G.g.dims[1] = G.g.dims[2] = 1; }
      else { G.n_filtered = vo.count; G.g.n_points = vo.count; }
    }
    G.cell_off = int(cell_total);
    cell_total += ncells + 1;
    max_filtered = std::max(max_filtered, G.n_filtered);
    reps[w].n_local_map = G.n_filtered;
  }
  if (cell_total > size_t(INT_MAX)) throw std::runtime_error("EstimatorBatch: the batch's cell tables exceed 2^31 entries");
  cells_all_.reserve(cell_total, s);
  LIO_HIP(hipMemcpyAsync(d_grid_.p, h_grid_, sizeof(BatchGrid) * B, hipMemcpyHostToDevice, s));
  launch_bw_cell_keys(d_win_.p, d_grid_.p, B, max_filtered, filtered_all_.p, ckeys_.p, s);
  {
    // a window's filtered points ordered by cell: as many 9-bit passes as the largest table's index needs
    SegDesc *seg = h_seg_ + B;
    int bits = 1;
    for (int w = 0; w < B; ++w) {
      seg[w] = SegDesc{h_win_[w].loc_off, h_grid_[w].n_filtered, 0};
      const long long nc = (long long)h_grid_[w].g.dims[0] * h_grid_[w].g.dims[1] * h_grid_[w].g.dims[2];
      while ((1ll << bits) < nc) ++bits;
    }
    const int passes = std::max(1, (bits + SS_MAX_BITS - 1) / SS_MAX_BITS);
    const SegSortPlan plan = seg_sort_plan(seg, B, SS_MAX_BITS);
    sort_hist_.reserve(std::max<size_t>(plan.hist_entries, 1), s);
    LIO_HIP(hipMemcpyAsync(d_seg_.p + B, seg, sizeof(SegDesc) * B, hipMemcpyHostToDevice, s));
    const uint32_t *ki = ckeys_.p, *vi = nullptr;
    uint32_t *ko = keys_.p, *vo = vals_.p, *ko2 = keysb_.p, *vo2 = valsb_.p;
    for (int p = 0; p < passes; ++p) {
      seg_sort_pass(d_seg_.p + B, B, plan, ki, vi, ko, vo, sort_hist_.p, p * SS_MAX_BITS, SS_MAX_BITS, nullptr, s);
      ki = ko; vi = vo;
      std::swap(ko, ko2); std::swap(vo, vo2);
    }
    launch_bw_cell_table(d_win_.p, d_grid_.p, B, max_filtered, filtered_all_.p, ki, vi, cells_all_.p, sorted_all_.p, s);
  }
  LIO_HIP(hipEventRecord(ev_[2], s));
  launch_bw_features(d_win_.p, d_grid_.p, B, max_M, max_static, q_static, knobs_, sorted_all_.p, cells_all_.p, valid_all_.p, coef_all_.p, score_all_.p, s);
  LIO_HIP(hipEventRecord(ev_[3], s));
  int round = 0;
  for (; round < 3; ++round)
    launch_bw_odom_round(d_win_.p, d_grid_.p, B, max_nb, q_newest, knobs_, round, d_odom_.p, sorted_all_.p, cells_all_.p, valid_all_.p, coef_all_.p, score_all_.p, odom_partials_.p,
                         nconv_.p, s);
  // ---- while the device searches: the problems of Estimator.cc:1660-1921, packed for the device loop.  Their uploads overwrite
  // what the previous solve's marginalization (on its own stream) still reads: everything enqueued from here on waits for it —
  // it has had this solve's filter, grids, features and first rounds to finish.
  const double t3a = bnow_ms();
  ev_wait_valid_ = false;
  if (marg_in_flight_) {
    LIO_HIP(hipEventRecord(ev_wait_[0], s));
    LIO_HIP(hipStreamWaitEvent(s, ev_marg_, 0));
    LIO_HIP(hipEventRecord(ev_wait_[1], s));   // (reached when the first three rounds are done AND the marginalization is: the difference is rounds + wait;
    ev_wait_valid_ = true;                     //  the rounds in front of it end at ev_wait_[0], so [0] -> [1] is the wait alone)
    marg_in_flight_ = false;
  }
  int max_bpf = 1, max_wo = 1, max_npad = DS_NB;
  size_t part_total = 0;
  for (int w = 0; w < B; ++w) {
    Win &Wn = win_[w];
    if (!Wn.device) continue;
    Estimator *e = Wn.e;
    int ms = 0;
    for (int i = e->W_ - e->Wo_ + 1; i <= e->W_; ++i) ms = std::max(ms, e->nslots_[i]);
    Wn.max_slots = ms;
    Wn.bpf = batch_blocks_per_frame(ms);
    if (e->total_slots_ == 0 || !e->BatchPackProblem(Wn.bpf, h_pb_[w], h_st_[w], &Wn.prior_used)) { Wn.device = false; continue; }
    Wn.part_off = part_total;
    part_total += size_t(e->Wo_) * Wn.bpf * LIO_MOMENT_OUT;
    max_bpf = std::max(max_bpf, Wn.bpf); max_wo = std::max(max_wo, e->Wo_); max_npad = std::max(max_npad, h_pb_[w].n_pad);
    // the prior's matrices on the device: already there (the previous solve's marginalization left them, or an earlier upload), or sent now
    if (Wn.prior_used) {
      int k = -1;
      for (int q = 0; q < 2; ++q) if (Wn.dev_prior[q] == Wn.prior_used) k = q;
      if (k < 0) {
        k = 0;
        Materialize(w, k);
        MargPrior &pr = *Wn.prior_used;
        pr.materialize();
        const size_t n = size_t(pr.n);
        double *h = h_prior_ + size_t(w) * ds_prior_mats_size(MARG_MAX_N);
        std::memcpy(h, pr.JtJ.a.data(), sizeof(double) * n * n);
        std::memcpy(h + n * n, pr.lin_jac.a.data(), sizeof(double) * n * n);
        std::memcpy(h + 2 * n * n, pr.lin_res.data(), sizeof(double) * n);
        std::memcpy(h + 2 * n * n + n, pr.Jtr0.data(), sizeof(double) * n);
        LIO_HIP(hipMemcpyAsync(slab_.p + size_t(w) * lay_.total + lay_.prior[k], h, sizeof(double) * ds_prior_mats_size(pr.n), hipMemcpyHostToDevice, s));
        Wn.dev_prior[k] = Wn.prior_used;
      }
      Wn.cur = k;
    } else {
      Wn.cur = 0;
    }
  }
  LIO_HIP(hipMemcpyAsync(d_pb_.p, h_pb_, sizeof(DevProblem) * B, hipMemcpyHostToDevice, s));
  LIO_HIP(hipMemcpyAsync(d_st_.p, h_st_, sizeof(DevState) * B, hipMemcpyHostToDevice, s));
  clk_.pack = bnow_ms() - t3a;
  // ---- the remaining rounds, with a look at the number of converged windows every second round
  for (; round < 10; ++round) {
    if (round >= 3 && round % 2 == 1) {
      LIO_HIP(hipMemcpyAsync(h_nconv_, nconv_.p, sizeof(int), hipMemcpyDeviceToHost, s));
      LIO_HIP(hipStreamSynchronize(s));
      if (*h_nconv_ >= B) break;
    }
    launch_bw_odom_round(d_win_.p, d_grid_.p, B, max_nb, q_newest, knobs_, round, d_odom_.p, sorted_all_.p, cells_all_.p, valid_all_.p, coef_all_.p, score_all_.p, odom_partials_.p,
                         nconv_.p, s);
  }
  clk_.rounds = round;
  LIO_HIP(hipEventRecord(ev_[4], s));
  LIO_HIP(hipMemcpyAsync(h_odom_, d_odom_.p, sizeof(OdomState) * B, hipMemcpyDeviceToHost, s));
  LIO_HIP(hipStreamSynchronize(s));
  const double t3 = bnow_ms();
  clk_.grid_features = t3 - t2;
  // ------------------------------------------------------------------------------------------------ the trust-region loop
  partials_.reserve(std::max<size_t>(part_total, 1), s);
  int max_it = 0, n_dev = 0;
  for (int w = 0; w < B; ++w) {
    Win &Wn = win_[w];
    BatchSolve &S = h_bs_[w];
    std::memset(static_cast<void *>(&S), 0, sizeof(S));
    S.marg = d_mg_.p + w;
    if (!Wn.device) continue;
    Estimator *e = Wn.e;
    e->BatchSetOdom(h_odom_[w]);
    const int pivot = e->W_ - e->Wo_;
    double *slab = slab_.p + size_t(w) * lay_.total;
    S.active = 1; S.nframes = e->Wo_; S.bpf = Wn.bpf;
    for (int i = 1; i <= e->Wo_; ++i) {
      MomentFrame &f = S.fr[i - 1];
      const int idx = pivot + i;
      f.stack = e->stacks_[idx].buf.p; f.M = std::max<int>(1, int(e->stacks_[idx].n));
      f.slot_off = h_win_[w].slot_base + e->slot_off_[idx]; f.nslots = e->nslots_[idx]; f.slot_begin = 0; f.slot_end = f.nslots;
    }
    S.pb = d_pb_.p + w; S.st = d_st_.p + w;
    S.prior_mats = slab + lay_.prior[Wn.cur]; S.next_prior_mats = slab + lay_.prior[1 - Wn.cur];
    S.partials = partials_.p + Wn.part_off;
    S.imu_out = slab + lay_.imu; S.lmap = slab + lay_.lmap; S.prior_out = slab + lay_.prior_out; S.exprior_out = slab + lay_.exprior;
    static const bool dbg_prof = std::getenv("LIO_DEBUG_TIMING") != nullptr;
    S.Hcur = slab + lay_.Hcur; S.S_buf = slab + lay_.Sbuf; S.prof = (dbg_prof && w == 0) ? reinterpret_cast<long long *>(slab + lay_.prof) : nullptr;
    S.marg_imu = slab + lay_.marg_imu; S.marg_lmap = slab + lay_.marg_lmap; S.marg_prior_out = slab + lay_.marg_prior_out;
    S.marg_A = slab + lay_.marg_A; S.marg_info = slab + lay_.marg_info;
    max_it = std::max(max_it, e->cfg_.max_num_iterations);
    ++n_dev;
  }
  LIO_HIP(hipMemcpyAsync(d_bs_.p, h_bs_, sizeof(BatchSolve) * B, hipMemcpyHostToDevice, s));
  if (n_dev > 0) {
    // iteration k evaluates candidate k (k = 0: the initial point); a window that is done costs its blocks one load each.
    // Groups of windows run their chains side by side (see est_batch.h); one group below 32 windows.
    // Groups by size, measured with the interleaved enqueue below (profiles/r6_m_loop_groups.txt; loop ms at 1 / 2 / 4 / 8 groups): 32 windows
    // 1.86 / 1.84 / 1.78 / 1.79, 64: 2.51 / 2.10 / 2.08 / 2.15, 128: 3.46 / 3.23 / 2.77 / 3.21 (four chains = four hardware queues), 256: 5.60 /
    // 5.05 / 5.28 / 5.04, 512: 10.3 / 9.6 / 9.6 / 9.6.
    const int G = knobs_.loop_groups ? std::min(knobs_.loop_groups, B) : (B >= 32 ? (B < 256 ? 4 : 2) : 1);
    // knobs_.aux_stream: the aux row on a side stream beside the moments.  Measured slower on the MI355X (two events per iteration cost
    // more than the 41 us they hide: loop 2.92 ms against 2.68 at 64 windows, profiles/r5_e_aux_stream_ab_and_step_phases.txt): off.
    const BatchBases bases{slab_.p, partials_.p, d_st_.p, d_pb_.p, d_mg_.p};
    const bool side_aux = knobs_.aux_stream != 0;
    static const int prof_it = [] { const char *e = std::getenv("LIO_DEBUG_TIMING_IT"); return e ? std::atoi(e) : 3; }();
    if (G > 1 || side_aux) LIO_HIP(hipEventRecord(ev_fork_, s));
    // The groups' chains are enqueued INTERLEAVED, iteration by iteration: a group's chain is 3 x (max_iterations + 1) launches, ~130 us of
    // host enqueue time — enqueued one whole chain after the other, group g started that much behind group g - 1 and the loop ended that
    // much later (at 64 windows the loop is one chain's latency, not throughput).
    struct Grp { int w0, w1, bpf, wo, npad, it, n; hipStream_t sg, sa; };
    Grp grp[kGroups];
    int it_max = 0;
    for (int g = 0; g < G; ++g) {
      Grp &q = grp[g];
      q.w0 = int((long long)B * g / G); q.w1 = int((long long)B * (g + 1) / G);
      q.sg = G > 1 ? stream_grp_[g] : s; q.sa = stream_aux_[g];
      if (G > 1) LIO_HIP(hipStreamWaitEvent(q.sg, ev_fork_, 0));
      q.bpf = 1; q.wo = 1; q.npad = DS_NB; q.it = 0; q.n = 0;
      for (int w = q.w0; w < q.w1; ++w) {
        if (!win_[w].device) continue;
        const Estimator *e = win_[w].e;
        q.bpf = std::max(q.bpf, win_[w].bpf); q.wo = std::max(q.wo, e->Wo_); q.npad = std::max(q.npad, h_pb_[w].n_pad);
        q.it = std::max(q.it, e->cfg_.max_num_iterations); ++q.n;
      }
      if (q.n > 0) it_max = std::max(it_max, q.it);
    }
    for (int k = 0; k <= it_max; ++k)
      for (int g = 0; g < G; ++g) {
        const Grp &q = grp[g];
        if (q.n == 0 || k > q.it) continue;
        const int w0 = q.w0, w1 = q.w1, g_bpf = q.bpf, g_wo = q.wo, g_npad = q.npad, g_it = q.it;
        hipStream_t sg = q.sg, sa = q.sa;
        const BatchSolve *gb = d_bs_.p + w0;
        if (side_aux) {   // aux row beside the moments; the step kernel joins the two
          LIO_HIP(hipStreamWaitEvent(sa, k == 0 ? ev_fork_ : ev_step_[g], 0));
          launch_bw_aux(gb, bases, w1 - w0, g_wo, knobs_.aux_threads, sa);
          LIO_HIP(hipEventRecord(ev_aux_[g], sa));
          launch_bw_moments(gb, bases, w1 - w0, g_bpf, g_wo, valid_all_.p, coef_all_.p, sg);
          LIO_HIP(hipStreamWaitEvent(sg, ev_aux_[g], 0));
          launch_bw_step(gb, bases, w1 - w0, g_wo, g_npad, sg);
          if (k < g_it) LIO_HIP(hipEventRecord(ev_step_[g], sg));
        } else if (knobs_.time_kernels) {   // measurement run: the three launches bracketed by events on their stream
          while (ev_k_.size() < size_t(ev_k_used_ + 4)) { hipEvent_t e = nullptr; LIO_HIP(hipEventCreate(&e)); ev_k_.push_back(e); }
          hipEvent_t *ek = ev_k_.data() + ev_k_used_;
          ev_k_used_ += 4;
          LIO_HIP(hipEventRecord(ek[0], sg));
          launch_bw_aux(gb, bases, w1 - w0, g_wo, knobs_.aux_threads, sg);
          LIO_HIP(hipEventRecord(ek[1], sg));
          launch_bw_moments(gb, bases, w1 - w0, g_bpf, g_wo, valid_all_.p, coef_all_.p, sg);
          LIO_HIP(hipEventRecord(ek[2], sg));
          launch_bw_step(gb, bases, w1 - w0, g_wo, g_npad, sg);
          LIO_HIP(hipEventRecord(ek[3], sg));
        } else {
          launch_bw_solve_iteration(gb, bases, w1 - w0, g_bpf, g_wo, g_npad, knobs_.aux_threads, valid_all_.p, coef_all_.p, sg);
        }
        if (h_bs_[0].prof && w0 == 0 && k == prof_it)   // LIO_DEBUG_TIMING: keep the stamps of this iteration's launch B beside the last one's
          LIO_HIP(hipMemcpyAsync(h_bs_[0].prof + 32, h_bs_[0].prof, 32 * sizeof(long long), hipMemcpyDeviceToDevice, sg));
      }
    if (G > 1)
      for (int g = 0; g < G; ++g) { LIO_HIP(hipEventRecord(ev_grp_[g], grp[g].sg)); LIO_HIP(hipStreamWaitEvent(s, ev_grp_[g], 0)); }
    LIO_HIP(hipEventRecord(ev_[5], s));
    LIO_HIP(hipMemcpyAsync(h_st_, d_st_.p, sizeof(DevState) * B, hipMemcpyDeviceToHost, s));
    LIO_HIP(hipStreamSynchronize(s));
  } else {
    LIO_HIP(hipEventRecord(ev_[5], s));
  }
  if (h_bs_[0].prof) {   // LIO_DEBUG_TIMING: the phase stamps of window 0's last launch B (shader clock, 100 MHz wall clock is not used here)
    long long pr[96];
    LIO_HIP(hipMemcpy(pr, h_bs_[0].prof, sizeof(pr), hipMemcpyDeviceToHost));
    for (int half = 1; half >= 0; --half) {
      std::fprintf(stderr, "[lio_hip timing] launch B of window 0, %s, clock64 ticks from its start:", half ? "iteration LIO_DEBUG_TIMING_IT (default 3)" : "last launch");
      for (int k = 0; k < 32; ++k) std::fprintf(stderr, " P%d %lld", k, pr[32 * half + k] ? pr[32 * half + k] - pr[32 * half] : -1);
      std::fprintf(stderr, "\n");
    }
    std::fprintf(stderr, "[lio_hip timing] aux row of window 0, block 0 (ImuFactor 0 + lidar map 1), last launch, clock64 ticks: raw Jacobians + residual %lld, whitening %lld, J^T J %lld, lidar map %lld\n",
                 pr[65] - pr[64], pr[66] - pr[65], pr[67] - pr[66], pr[68] - pr[67]);
  }
  clk_.iterations = max_it + 1;
  const double t4 = bnow_ms();
  clk_.solve = t4 - t3;
  // ------------------------------------------------------------------------------------------------ write-back, marginalization
  int max_n = 1, n_marg = 0;
  std::vector<int> host_path;
  // per window: the state back into the estimator (DoubleToVector), the report, the marginalization's layout.  Window-local host
  // work with the device idle behind it (1.3 ms at 512 windows on one thread): from 128 windows on, four threads share it.
  std::vector<std::shared_ptr<MargPrior>> shells(B);
  std::vector<char> margs(B, 0);
  auto finish_range = [&](int w0, int w1) {
    for (int w = w0; w < w1; ++w) {
      Win &Wn = win_[w];
      DevMarg &mg = h_mg_[w];
      std::memset(&mg, 0, sizeof(mg));
      if (!Wn.device) continue;
      const DevState &st = h_st_[w];
      if (st.need_host || !st.started) continue;
      margs[w] = Wn.e->BatchFinish(st, Wn.prior_used, reps[w], mg, &shells[w]) ? 1 : 0;
    }
  };
  {
    const int T = knobs_.finish_threads ? std::min(knobs_.finish_threads, B) : (B >= 256 ? 8 : (B >= 128 ? 4 : 1));   // (512 windows: 1.31 -> 1.03 ms with eight, profiles/r6_h_*)
    if (T == 1) {
      finish_range(0, B);
    } else {
      std::vector<std::thread> pool;
      std::vector<std::exception_ptr> errs(T);
      for (int t = 0; t < T; ++t)
        pool.emplace_back([&, t] {
          try { finish_range(int((long long)B * t / T), int((long long)B * (t + 1) / T)); } catch (...) { errs[t] = std::current_exception(); }
        });
      for (std::thread &th : pool) th.join();
      for (const std::exception_ptr &e : errs) if (e) std::rethrow_exception(e);
    }
  }
  for (int w = 0; w < B; ++w) {
    Win &Wn = win_[w];
    if (!Wn.device) { host_path.push_back(w); continue; }
    const DevState &st = h_st_[w];
    if (st.need_host || !st.started) { Wn.device = false; host_path.push_back(w); continue; }
    if (margs[w]) {
      const int nb = 1 - Wn.cur;
      Materialize(w, nb);
      std::shared_ptr<MargPrior> &shell = shells[w];
      shell->on_device = true;
      shell->fetch = [this, w, nb](MargPrior &pr) { FetchPrior(w, nb, pr); };
      Wn.dev_prior[nb] = shell;
      Wn.e->last_marg_ = shell;
      max_n = std::max(max_n, h_mg_[w].n);
      ++n_marg;
    }
    Wn.prior_used.reset();
  }
  clk_.n_device = B - int(host_path.size()); clk_.n_host = int(host_path.size());
  if (n_marg > 0) {
    // (the host has waited for the loop; the marginalization's inputs — final moments, states, problems — are complete)
    LIO_HIP(hipMemcpyAsync(d_mg_.p, h_mg_, sizeof(DevMarg) * B, hipMemcpyHostToDevice, stream_marg_));
    launch_bw_marginalize(d_bs_.p, BatchBases{slab_.p, partials_.p, d_st_.p, d_pb_.p, d_mg_.p}, B, max_wo, max_n, stream_marg_);
    LIO_HIP(hipEventRecord(ev_marg_, stream_marg_));
    marg_in_flight_ = true;
  }
  LIO_HIP(hipEventRecord(ev_[6], stream_marg_));
  ev_valid_ = true;
  const double t5 = bnow_ms();
  clk_.finish = t5 - t4;
  // ------------------------------------------------------------------------------------------------ windows the device loop did not take
  for (int w = 0; w < B; ++w) ok_[size_t(w)] = 1;
  for (int w : host_path) ok_[size_t(w)] = win_[w].e->SolveOptimizationHost(&reps[w]) ? 1 : 0;
  const double t6 = bnow_ms();
  clk_.fallback = t6 - t5;
  clk_.total = t6 - t0;
  for (int w = 0; w < B; ++w) {
    if (!win_[w].device) continue;
    lio_solve_report &R = reps[w];
    R.ms_build_map = clk_.map; R.ms_features = clk_.grid_features; R.ms_prepare = clk_.describe + clk_.pack; R.ms_opt = clk_.solve; R.ms_marg = clk_.finish;
    R.ms_total = t5 - t0;
  }
  return B;
}

}  // namespace lio
