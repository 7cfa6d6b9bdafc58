// capi.hip — include/lio_c.h implemented by the product (liblio_hip.so).  Every entry point that touches
// point data runs on the GPU; a missing device or a HIP failure is reported as LIO_ERR_DEVICE — there is
// no CPU fallback.  Pure-host entry points (pre-integration, single-factor evaluation) are the host
// half of the estimator and also back the `-m "not gpu"` ABI tests.
#include <cstring>
#include <new>

#include "../../include/lio_c.h"
#include "est_batch.h"
#include "seg_sort.h"
#include "estimator.h"
#include "rccl_comm.h"
#include "host_init.h"
#include "kf_batch.h"
#include <mutex>
#include <thread>
#include <atomic>

#include "mapping.h"
#include "odometry.h"
#include "pointproc.h"

using namespace lio;

struct lio_pim { std::shared_ptr<Preintegration> p; };
// (members are destroyed in reverse order: the batch of one that serves lio_est_config.device_solve goes before the estimator it adopted)
struct lio_est { std::unique_ptr<Estimator> e; EstConfig cfg; std::unique_ptr<MappingDev> map; lio_map_config map_cfg; std::unique_ptr<EstimatorBatch> solo; bool adopted = false; struct lio_est_batch *owner = nullptr; };
// A batch of at least kBatchSplitFrom windows is TWO EstimatorBatch objects (the first and the second half of the windows) solved side by
// side from two host threads: one part's host phases (describe, pack, write-back), syncs and latency-bound stages fill with the other
// part's kernels (tools/batches_in_flight.py: 2 x 256 windows 20.1 k solves/s against 19.0 k for 1 x 512, 2 x 64 17.4 k against 16.1 k).
// Every window's results are those of the window alone whatever the partition (the batch's contract), so the split is an execution
// choice like the others ("parts", lio_est_batch_set_option).
struct lio_est_batch {
  std::unique_ptr<EstimatorBatch> b, b2;   // b2: the second part (null: one part)
  int n1 = 0;                              // windows of b
  int parts_opt = 0;                       // 0: by size, 1, 2
  std::vector<std::pair<std::string, int>> options;   // what lio_est_batch_set_option has set, re-applied when the partition changes
  std::vector<lio_est *> members;
};
static constexpr int kBatchSplitFrom = 96;    // measured (profiles/r6_n_batch_parts.txt): 16 .. 64 windows within noise, 96: + 4 %, 128 .. 512: + 7-8 %
// lio_pp_process_batch runs its sweeps through ONE multi-sweep processor shared by the handles of the call (`pool`); a handle whose last
// sweep went that way reads its results from sweep `pool_sweep` of it.  The pool is shared state: its users take `mu`.
// `gen` counts the pool's batches: a handle whose results a LATER batch of other handles overwrote (same pool, its own sweep not among them)
// is told so instead of being handed somebody else's sweep.
struct lio_pp_pool { PointProcessorDev pp; std::mutex mu; unsigned long gen = 0; lio_pp_pool(float lo, float up, int r, const lio_pp_config &c) : pp(lo, up, r, c) {} };
struct lio_pp { std::unique_ptr<PointProcessorDev> pp; std::shared_ptr<lio_pp_pool> pool; int pool_sweep = -1; unsigned long pool_gen = 0; };
struct lio_odom { std::unique_ptr<OdometryDev> o; };
struct lio_map { std::unique_ptr<MappingDev> m; };

static V3d v3(const double *p) { return V3d(p[0], p[1], p[2]); }
static Rigidf toT(const lio_transform_f &t) { return Rigidf(Quat<float>(t.q[3], t.q[0], t.q[1], t.q[2]), Vec3<float>(t.p[0], t.p[1], t.p[2])); }
static void fromT(const Rigidf &T, lio_transform_f *o) {
  o->q[0] = T.rot.x; o->q[1] = T.rot.y; o->q[2] = T.rot.z; o->q[3] = T.rot.w; o->p[0] = T.pos.x; o->p[1] = T.pos.y; o->p[2] = T.pos.z;
}

static thread_local char g_last_error[512];
template <typename F> static int guarded(F &&f) {
  try {
    return f();
  } catch (const DeviceError &e) {
    std::snprintf(g_last_error, sizeof(g_last_error), "%s", e.what());
    std::fprintf(stderr, "[lio_hip] %s\n", e.what());
    return LIO_ERR_DEVICE;
  } catch (const std::exception &e) {
    std::snprintf(g_last_error, sizeof(g_last_error), "%s", e.what());
    std::fprintf(stderr, "[lio_hip] %s\n", e.what());
    return LIO_ERR_STATE;
  }
}

// the processor that holds the handle's last sweep, ready to be read: f(processor) under the pool's lock when that is the shared one
template <typename F> static int pp_read(const lio_pp *h, F &&f) {
  return guarded([&] {
    if (h->pool && h->pool_sweep >= 0) {
      std::lock_guard<std::mutex> lk(h->pool->mu);
      if (h->pool_gen != h->pool->gen) throw std::runtime_error("lio_pp: this handle's batch results were overwritten by a later lio_pp_process_batch on the storage it shared");
      h->pool->pp.ProcessFinish();
      h->pool->pp.SelectSweep(h->pool_sweep);
      f(h->pool->pp);
    } else {
      h->pp->ProcessFinish();   // a sweep still in flight (lio_pp_process_async) is waited for
      f(*h->pp);
    }
    return LIO_OK;
  });
}

// f(part, first window of the part) on every part — two parts on two host threads
template <typename F> static void batch_for_parts(lio_est_batch *h, F &&f) {
  if (!h->b2) { f(*h->b, 0); return; }
  std::exception_ptr err2;
  std::thread t2([&] { try { f(*h->b2, h->n1); } catch (...) { err2 = std::current_exception(); } });
  std::exception_ptr err1;
  try { f(*h->b, 0); } catch (...) { err1 = std::current_exception(); }
  t2.join();
  if (err1) std::rethrow_exception(err1);
  if (err2) std::rethrow_exception(err2);
}

extern "C" {

const char *lio_backend(void) { return "hip-gfx950"; }

// ---------------------------------------------------------------- PointProcessor
void lio_pp_default_config(lio_pp_config *c) {
  if (!c) return;
  c->scan_period = 0.1; c->num_scan_subregions = 8; c->num_curvature_regions = 5; c->surf_curv_th = 0.1f;
  c->max_corner_sharp = 2; c->max_corner_less_sharp = 20; c->max_surf_flat = 4; c->less_flat_filter_size = 0.2f;
  c->infer_start_ori = 0; c->rad_diff = 0.2;
}
int lio_pp_check_config(float lo, float up, int rings, const lio_pp_config *c) {
  if (rings <= 0 || rings > LIO_PP_MAX_RINGS || !(up > lo)) return LIO_ERR_ARG;
  lio_pp_config cfg;
  if (c) cfg = *c; else lio_pp_default_config(&cfg);
  if (cfg.num_scan_subregions < 1 || cfg.num_scan_subregions > 16 || cfg.num_curvature_regions < 1 || cfg.num_curvature_regions > 8) return LIO_ERR_ARG;
  // pick caps size LDS tables and device reserves; the leaf is inverted: keep them in sane ranges
  const int caps[3] = {cfg.max_corner_sharp, cfg.max_corner_less_sharp, cfg.max_surf_flat};
  for (int v : caps)
    if (v < 0 || v > 4096) return LIO_ERR_ARG;
  if (cfg.max_corner_sharp > cfg.max_corner_less_sharp) return LIO_ERR_ARG;
  if (!(cfg.less_flat_filter_size > 1e-4f && cfg.less_flat_filter_size < 1e4f)) return LIO_ERR_ARG;
  if (!(cfg.scan_period > 0.0f) || !std::isfinite(cfg.scan_period)) return LIO_ERR_ARG;
  if (cfg.infer_start_ori && !(cfg.rad_diff >= 0.0)) return LIO_ERR_ARG;
  // k_ring_pick keeps the picks of a subregion one per lane of a wave (corner picks + flat picks in 64 slots): a fixed capacity of
  // the device path, reported as such (the reference accepts any quota; its defaults are 20 + 4)
  if (cfg.max_corner_less_sharp + cfg.max_surf_flat > 64) return LIO_ERR_CAPACITY;
  return LIO_OK;
}
lio_pp *lio_pp_create(float lo, float up, int rings, const lio_pp_config *c) {
  const int chk = lio_pp_check_config(lo, up, rings, c);
  if (chk != LIO_OK) {
    if (chk == LIO_ERR_CAPACITY)
      std::fprintf(stderr, "lio_pp_create: LIO_ERR_CAPACITY — max_corner_less_sharp + max_surf_flat > 64 picks per subregion is beyond k_ring_pick's per-wave pick table "
                           "(lio_pp_check_config reports the code)\n");
    return nullptr;
  }
  lio_pp_config cfg;
  if (c) cfg = *c; else lio_pp_default_config(&cfg);
  lio_pp *h = new (std::nothrow) lio_pp;
  if (!h) return nullptr;
  int rc = guarded([&] { h->pp.reset(new PointProcessorDev(lo, up, rings, cfg)); return LIO_OK; });
  if (rc != LIO_OK) { delete h; return nullptr; }
  return h;
}
void lio_pp_destroy(lio_pp *h) { delete h; }
float lio_pp_start_ori(const lio_pp *h) {
  if (!h) return std::nanf("");
  float v = std::nanf("");
  pp_read(h, [&](PointProcessorDev &p) { v = p.StartOri(); });
  return v;
}
int lio_pp_process(lio_pp *h, const float *xyzi, size_t n) {
  if (!h || (!xyzi && n)) return LIO_ERR_ARG;
  h->pool_sweep = -1;
  return guarded([&] { h->pp->Process(xyzi, n); return LIO_OK; });
}
int lio_pp_process_async(lio_pp *h, const float *xyzi, size_t n) {
  if (!h || (!xyzi && n)) return LIO_ERR_ARG;
  h->pool_sweep = -1;
  return guarded([&] { h->pp->ProcessLaunch(xyzi, n); return LIO_OK; });
}
int lio_pp_wait(lio_pp *h) {
  if (!h) return LIO_ERR_ARG;
  return guarded([&] { h->pp->ProcessFinish(); return LIO_OK; });
}
static int pp_process_batch(lio_pp *const *handles, const float *const *xyzi, const size_t *n, int n_sweeps, bool on_device,
                            const uint16_t *const *ring = nullptr) {
  if (n_sweeps < 0 || (n_sweeps > 0 && (!handles || !xyzi || !n))) return LIO_ERR_ARG;
  bool same = true;
  for (int k = 0; k < n_sweeps; ++k) {
    if (!handles[k] || (!xyzi[k] && n[k]) || (ring && !ring[k] && n[k])) return LIO_ERR_ARG;
    for (int j = 0; j < k; ++j) if (handles[j] == handles[k]) return LIO_ERR_ARG;
    same = same && handles[k]->pp->SameSensor(*handles[0]->pp);
  }
  if (n_sweeps == 0) return LIO_OK;
  if (n_sweeps >= 2 && same) {
    // ONE launch chain over all sweeps: the handles share a multi-sweep processor (kept for the next call), every handle reads its own
    // sweep of it; a sweep's start azimuth still goes through ITS handle's ten-sweep history (infer_start_ori)
    return guarded([&] {
      std::shared_ptr<lio_pp_pool> pool = handles[0]->pool;
      if (!pool || !pool->pp.SameSensor(*handles[0]->pp)) {
        const PointProcessorDev &a = *handles[0]->pp;
        pool = std::make_shared<lio_pp_pool>(a.lower(), a.upper(), a.rings(), a.config());
      }
      std::vector<StartOriFilter *> filters(static_cast<size_t>(n_sweeps));
      std::lock_guard<std::mutex> lk(pool->mu);
      ++pool->gen;
      for (int k = 0; k < n_sweeps; ++k) {
        handles[k]->pp->ProcessFinish();
        handles[k]->pool = pool; handles[k]->pool_sweep = k; handles[k]->pool_gen = pool->gen;
        filters[size_t(k)] = &handles[k]->pp->start_ori_filter();
      }
      pool->pp.ProcessLaunchBatch(xyzi, ring, n, n_sweeps, on_device, filters.data());
      pool->pp.ProcessFinish();
      return LIO_OK;
    });
  }
  // different sensors (or one sweep): every handle's own chain, all enqueued before the first is waited for
  int rc = LIO_OK, launched = 0;
  for (; launched < n_sweeps && rc == LIO_OK; ++launched) {
    lio_pp *h = handles[launched];
    h->pool_sweep = -1;
    rc = guarded([&] {
      StartOriFilter *f = &h->pp->start_ori_filter();
      h->pp->ProcessLaunchBatch(&xyzi[launched], ring ? &ring[launched] : nullptr, &n[launched], 1, on_device, &f);
      return LIO_OK;
    });
  }
  if (rc != LIO_OK) --launched;   // (the failing handle has nothing in flight)
  for (int k = 0; k < launched; ++k) { const int r = lio_pp_wait(handles[k]); if (rc == LIO_OK) rc = r; }
  return rc;
}
int lio_pp_process_batch(lio_pp *const *handles, const float *const *xyzi, const size_t *n, int n_sweeps) {
  return pp_process_batch(handles, xyzi, n, n_sweeps, false);
}
int lio_pp_process_batch_device(lio_pp *const *handles, const float *const *d_xyzi, const size_t *n, int n_sweeps) {
  return pp_process_batch(handles, d_xyzi, n, n_sweeps, true);
}
int lio_pp_process_rings_batch(lio_pp *const *handles, const float *const *xyzi, const uint16_t *const *ring, const size_t *n, int n_sweeps) {
  if (n_sweeps > 0 && !ring) return LIO_ERR_ARG;
  return pp_process_batch(handles, xyzi, n, n_sweeps, false, ring);
}
int lio_pp_process_rings(lio_pp *h, const float *xyzi, const uint16_t *ring, size_t n) {
  if (!h || ((!xyzi || !ring) && n)) return LIO_ERR_ARG;
  h->pool_sweep = -1;
  return guarded([&] { h->pp->Process(xyzi, n, n ? ring : nullptr); return LIO_OK; });
}
size_t lio_pp_count(const lio_pp *h, int which) {
  if (!h || which < 0 || which > 4) return 0;
  size_t n = 0;
  pp_read(h, [&](PointProcessorDev &p) { n = p.Count(which); });
  return n;
}
int lio_pp_get_cloud(const lio_pp *h, int which, float *out) {
  if (!h || which < 0 || which > 4 || !out) return LIO_ERR_ARG;
  return pp_read(h, [&](PointProcessorDev &p) { p.GetCloud(which, out); });
}
int lio_pp_get_indices(const lio_pp *h, int which, int32_t *ring, int32_t *idx) {
  if (!h || which < 1 || which > 3 || !ring || !idx) return LIO_ERR_ARG;
  return pp_read(h, [&](PointProcessorDev &p) { p.GetIndices(which, ring, idx); });
}
int lio_pp_get_ring_offsets(const lio_pp *h, int32_t *out) {
  if (!h || !out) return LIO_ERR_ARG;
  return pp_read(h, [&](PointProcessorDev &p) { p.GetRingOffsets(out); });
}
int lio_pp_get_curvature(const lio_pp *h, float *curv, int32_t *mask) {
  if (!h) return LIO_ERR_ARG;
  return pp_read(h, [&](PointProcessorDev &p) { p.GetCurvature(curv, mask); });
}

int lio_pp_get_ring_intensity(const lio_pp *h, float *out) {
  if (!h || !out) return LIO_ERR_ARG;
  return pp_read(h, [&](PointProcessorDev &p) { p.GetRingIntensity(out); });
}

// ---------------------------------------------------------------- PointOdometry
lio_odom *lio_odom_create(float scan_period, int io_ratio, int max_iter, int no_deskew) {
  if (!(scan_period > 0) || max_iter < 1) return nullptr;
  lio_odom *h = new (std::nothrow) lio_odom;
  if (!h) return nullptr;
  int rc = guarded([&] { h->o.reset(new OdometryDev(scan_period, io_ratio, max_iter, no_deskew != 0)); return LIO_OK; });
  if (rc != LIO_OK) { delete h; return nullptr; }
  return h;
}
void lio_odom_destroy(lio_odom *h) { delete h; }
int lio_odom_process(lio_odom *h, const float *sharp, size_t n_sharp, const float *less_sharp, size_t n_ls, const float *flat, size_t n_flat,
                     const float *less_flat, size_t n_lf, lio_transform_f *Tsum, lio_transform_f *Tes, int *iters, int *nsel) {
  if (!h || (!sharp && n_sharp) || (!less_sharp && n_ls) || (!flat && n_flat) || (!less_flat && n_lf)) return LIO_ERR_ARG;
  return guarded([&] {
    h->o->Process(sharp, n_sharp, less_sharp, n_ls, flat, n_flat, less_flat, n_lf);
    if (Tsum) fromT(h->o->transform_sum_, Tsum);
    if (Tes) fromT(h->o->transform_es_, Tes);
    if (iters) *iters = h->o->iterations_done_;
    if (nsel) *nsel = h->o->last_num_sel_;
    return LIO_OK;
  });
}
int lio_odom_get_iteration_trace(const lio_odom *h, lio_transform_f *trace, int capacity, int *kz) {
  if (!h || capacity < 0 || (!trace && capacity)) return LIO_ERR_ARG;
  const int n = h->o->iterations_done_;
  for (int k = 0; k < n && k < capacity && k < int(h->o->es_trace_.size()); ++k) fromT(h->o->es_trace_[size_t(k)], &trace[k]);
  if (kz) *kz = h->o->last_kz_;
  return n;
}
int lio_odom_enable(lio_odom *h, int on) {
  if (!h) return LIO_ERR_ARG;
  h->o->enable_odom_ = on != 0;
  return LIO_OK;
}
size_t lio_odom_get_last_cloud(const lio_odom *h, int which, float *out) {
  if (!h || which < 0 || which > 1) return 0;
  size_t n = 0;
  guarded([&] { n = h->o->GetLastCloud(which, out); return LIO_OK; });
  return n;
}

// ---------------------------------------------------------------- ImuInitializer (host)
static bool gatherLaserFrames(size_t n, const lio_transform_f *T, lio_pim *const *pims, std::vector<LaserFrame> &all) {
  all.resize(n);
  for (size_t i = 0; i < n; ++i) {
    all[i].transform = toT(T[i]);
    if (pims[i]) all[i].pim = pims[i]->p;
    else if (i != 0) return false;
  }
  return true;
}
int lio_imu_estimate_extrinsic_rotation(size_t n, const lio_transform_f *T, lio_pim *const *pims, lio_transform_f *lb) {
  if (n < 2 || !T || !pims || !lb) return LIO_ERR_ARG;
  std::vector<LaserFrame> all;
  if (!gatherLaserFrames(n, T, pims, all)) return LIO_ERR_ARG;
  return guarded([&] {
    Rigidf tlb = toT(*lb);
    const bool ok = estimate_extrinsic_rotation(all, tlb);
    fromT(tlb, lb);
    return ok ? 1 : 0;
  });
}
int lio_imu_initialization(size_t n, const lio_transform_f *T, lio_pim *const *pims, const lio_transform_f *lb, double *Vs, double *Bgs, double g[3],
                           double R_WI[9]) {
  if (n < 2 || !T || !pims || !lb || !Vs || !Bgs || !g || !R_WI) return LIO_ERR_ARG;
  std::vector<LaserFrame> all;
  if (!gatherLaserFrames(n, T, pims, all)) return LIO_ERR_ARG;
  return guarded([&] {
    std::vector<V3d> vs(n), bgs(n);
    for (size_t i = 0; i < n; ++i) bgs[i] = v3(Bgs + 3 * i);
    V3d gv;
    M3d R;
    R(0, 0) = R(1, 1) = R(2, 2) = 1.0;
    const bool ok = imu_initialization(all, vs, bgs, gv, toT(*lb), R);
    for (size_t i = 0; i < n; ++i) {
      Vs[3 * i] = vs[i].x; Vs[3 * i + 1] = vs[i].y; Vs[3 * i + 2] = vs[i].z;
      Bgs[3 * i] = bgs[i].x; Bgs[3 * i + 1] = bgs[i].y; Bgs[3 * i + 2] = bgs[i].z;
    }
    g[0] = gv.x; g[1] = gv.y; g[2] = gv.z;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R_WI[r * 3 + c] = R(r, c);
    return ok ? 1 : 0;
  });
}

// ---------------------------------------------------------------- PointMapping
void lio_map_default_config(lio_map_config *c) {
  if (!c) return;
  c->corner_filter_size = 0.2f; c->surf_filter_size = 0.4f; c->min_match_sq_dis = 1.0f; c->min_plane_dis = 0.2f; c->num_max_iterations = 10;
  c->map_builder = 0; c->enable_4d = 1; c->skip_count = 2;
}
lio_map *lio_map_create(const lio_map_config *c) {
  lio_map_config cfg;
  if (c) cfg = *c; else lio_map_default_config(&cfg);
  if (!(cfg.corner_filter_size > 0) || !(cfg.surf_filter_size > 0) || cfg.num_max_iterations < 1) return nullptr;
  if (cfg.map_builder && cfg.skip_count < 1) return nullptr;
  lio_map *h = new (std::nothrow) lio_map;
  if (!h) return nullptr;
  int rc = guarded([&] { h->m.reset(new MappingDev(cfg)); return LIO_OK; });
  if (rc != LIO_OK) { delete h; return nullptr; }
  return h;
}
void lio_map_destroy(lio_map *h) { delete h; }
int lio_map_process(lio_map *h, const float *corner, size_t nc, const float *surf, size_t ns, const lio_transform_f *Tsum, lio_transform_f *Taft,
                    int *iters, int *nsel) {
  if (!h || !Tsum || (!corner && nc) || (!surf && ns)) return LIO_ERR_ARG;
  return guarded([&] {
    h->m->Process(corner, nc, surf, ns, toT(*Tsum));
    if (Taft) fromT(h->m->transform_aft_mapped_, Taft);
    if (iters) *iters = h->m->iterations_;
    if (nsel) *nsel = h->m->num_selected_;
    return LIO_OK;
  });
}
int lio_map_get_degeneracy(const lio_map *h, int *kz) {
  if (!h) return LIO_ERR_ARG;
  if (kz) *kz = h->m->kz_;
  return h->m->degenerate_ ? 1 : 0;
}
int lio_map_set_init_flag(lio_map *h, int on) {
  if (!h) return LIO_ERR_ARG;
  h->m->imu_inited_ = on != 0;
  return LIO_OK;
}
int lio_map_set_transform_tobe_mapped(lio_map *h, const lio_transform_f *T) {
  if (!h || !T) return LIO_ERR_ARG;
  h->m->transform_tobe_mapped_ = toT(*T);
  return LIO_OK;
}
int lio_map_get_transform_tobe_mapped(const lio_map *h, lio_transform_f *T) {
  if (!h || !T) return LIO_ERR_ARG;
  fromT(h->m->transform_tobe_mapped_, T);
  return LIO_OK;
}
int lio_map_update_map_database(lio_map *h, const float *corner, size_t nc, const float *surf, size_t ns, const uint32_t *valid, size_t nv,
                                const lio_transform_f *T, const int cen[3]) {
  if (!h || !T || !cen || (!corner && nc) || (!surf && ns) || (!valid && nv)) return LIO_ERR_ARG;
  for (size_t i = 0; i < nv; ++i)
    if (valid[i] >= uint32_t(MappingDev::L * MappingDev::Wd * MappingDev::H)) return LIO_ERR_ARG;
  return guarded([&] { h->m->UpdateMapDatabase(corner, nc, surf, ns, valid, nv, toT(*T), cen); return LIO_OK; });
}
size_t lio_map_get_cloud(const lio_map *h, int which, float *out) {
  if (!h || which < 0 || which > 3) return 0;
  size_t n = 0;
  guarded([&] { n = h->m->GetCloud(which, out); return LIO_OK; });
  return n;
}
size_t lio_map_get_cube(const lio_map *h, int cls, uint32_t idx, float *out) {
  if (!h || cls < 0 || cls > 1 || idx >= uint32_t(MappingDev::L * MappingDev::Wd * MappingDev::H)) return 0;
  size_t n = 0;
  guarded([&] { n = h->m->GetCube(cls, idx, out); return LIO_OK; });
  return n;
}
size_t lio_map_get_cube_state(const lio_map *h, int cen[3], uint32_t *valid) {
  if (!h) return 0;
  if (cen) for (int d = 0; d < 3; ++d) cen[d] = h->m->cen_[d];
  if (valid) for (size_t i = 0; i < h->m->valid_idx_.size(); ++i) valid[i] = h->m->valid_idx_[i];
  return h->m->valid_idx_.size();
}
size_t lio_map_get_score_point_coeff(const lio_map *h, float *score, float *point, float *coeff) {
  if (!h) return 0;
  size_t n = 0;
  guarded([&] { n = h->m->GetScorePointCoeff(score, point, coeff); return LIO_OK; });
  return n;
}

// ---------------------------------------------------------------- batched keyframe refinement
struct lio_kf_batch { std::unique_ptr<KfBatchDev> b; };
lio_kf_batch *lio_kf_batch_create(const lio_map_config *c) {
  lio_map_config cfg;
  if (c) cfg = *c; else lio_map_default_config(&cfg);
  if (cfg.num_max_iterations < 1) return nullptr;
  lio_kf_batch *h = new (std::nothrow) lio_kf_batch;
  if (!h) return nullptr;
  int rc = guarded([&] { h->b.reset(new KfBatchDev(cfg)); return LIO_OK; });
  if (rc != LIO_OK) { delete h; return nullptr; }
  return h;
}
void lio_kf_batch_destroy(lio_kf_batch *h) { delete h; }
int lio_kf_batch_add_map(lio_kf_batch *h, const float *corner, size_t nc, const float *surf, size_t ns) {
  if (!h || (!corner && nc) || (!surf && ns)) return LIO_ERR_ARG;
  int idx = -1;
  int rc = guarded([&] { idx = h->b->AddMap(corner, nc, surf, ns); return LIO_OK; });
  return rc == LIO_OK ? idx : rc;
}
int lio_kf_batch_add_keyframe(lio_kf_batch *h, int map, const float *corner, size_t nc, const float *surf, size_t ns, const lio_transform_f *T) {
  if (!h || !T || (!corner && nc) || (!surf && ns) || map < 0 || size_t(map) >= h->b->n_maps()) return LIO_ERR_ARG;
  int idx = -1;
  int rc = guarded([&] { idx = h->b->AddKeyframe(map, corner, nc, surf, ns, toT(*T)); return LIO_OK; });
  return rc == LIO_OK ? idx : rc;
}
int lio_kf_batch_clear_keyframes(lio_kf_batch *h) {
  if (!h) return LIO_ERR_ARG;
  h->b->ClearKeyframes();
  return LIO_OK;
}
size_t lio_kf_batch_size(const lio_kf_batch *h) { return h ? h->b->n_keyframes() : 0; }
int lio_kf_batch_refine(lio_kf_batch *h, lio_transform_f *T_out, int32_t *iters, int32_t *rows, double *device_ms) {
  if (!h) return LIO_ERR_ARG;
  return guarded([&] {
    h->b->Refine();
    const auto &st = h->b->states();
    for (size_t k = 0; k < st.size(); ++k) {
      if (T_out) fromT(Rigidf(Quat<float>(st[k].T[3], st[k].T[0], st[k].T[1], st[k].T[2]), Vec3<float>(st[k].T[4], st[k].T[5], st[k].T[6])), &T_out[k]);
      if (iters) iters[k] = st[k].iters;
      if (rows) rows[k] = st[k].nsel;
    }
    if (device_ms) *device_ms = h->b->device_ms_;
    return LIO_OK;
  });
}
int lio_kf_batch_get_degeneracy(const lio_kf_batch *h, int32_t *kz_out) {
  if (!h || !kz_out) return LIO_ERR_ARG;
  const auto &st = h->b->states();
  if (st.size() != h->b->n_keyframes()) return LIO_ERR_STATE;   // no refine since the keyframe list changed
  for (size_t k = 0; k < st.size(); ++k) kz_out[k] = st[k].kz;
  return LIO_OK;
}
int lio_kf_batch_refine_gather(lio_kf_batch *h, lio_rccl *comm, int slots_per_rank, float *packed_all, double *device_ms) {
  if (!h || !comm || !packed_all || slots_per_rank < 1) return LIO_ERR_ARG;
  // collective precondition, checked before any device work: slots_per_rank is the SAME on every rank and >= every rank's
  // keyframe count (a rank that returned here would leave its peers waiting in ncclAllGather — the caller sizes it from the
  // global maximum, dist_util.refine_keyframes_sharded)
  if (size_t(slots_per_rank) < h->b->n_keyframes()) return LIO_ERR_CAPACITY;
  return guarded([&] {
    h->b->RefineGather(rccl_raw_comm(comm), lio_rccl_world(comm), slots_per_rank, packed_all);
    if (device_ms) *device_ms = h->b->device_ms_;
    return LIO_OK;
  });
}

// ---------------------------------------------------------------- /compact_data codec (host: a memcpy-class wire format)
size_t lio_compact_encode(const lio_transform_f *T, const float *corner, size_t nc, const float *surf, size_t ns, const float *full, size_t nf,
                          float *out) {
  if (!T || !out || (!corner && nc) || (!surf && ns) || (!full && nf)) return 0;
  float *o = out;
  *o++ = T->p[0]; *o++ = T->p[1]; *o++ = T->p[2]; *o++ = 0.f;
  *o++ = T->q[0]; *o++ = T->q[1]; *o++ = T->q[2]; *o++ = T->q[3];
  *o++ = float(nc); *o++ = float(ns); *o++ = float(nf); *o++ = T->q[3];  // the reused PointT keeps intensity = qw
  const float *src[3] = {corner, surf, full};
  const size_t cnt[3] = {nc, ns, nf};
  for (int k = 0; k < 3; ++k) { if (cnt[k]) std::memcpy(o, src[k], cnt[k] * 4 * sizeof(float)); o += 4 * cnt[k]; }
  return size_t(o - out) / 4;
}
int lio_compact_decode(const float *d, size_t n, lio_transform_f *T, size_t *nc, size_t *ns, size_t *nf) {
  if (!d || !nc || !ns || !nf || n < 4) return LIO_ERR_ARG;
  // the three sizes are untrusted wire floats: reject NaN / Inf / negative / larger-than-the-message values BEFORE any
  // float -> integer cast (undefined behaviour otherwise) and before they are summed
  const float lim = float(n);
  for (int k = 8; k <= 10; ++k)
    if (!(d[k] >= 0.0f && d[k] <= lim)) return LIO_ERR_ARG;
  const unsigned long long c = (unsigned long long)d[8], s = (unsigned long long)d[9], f = (unsigned long long)d[10];
  if (3ull + c + s + f != (unsigned long long)n) return LIO_ERR_ARG;
  if (T) { for (int k = 0; k < 3; ++k) T->p[k] = d[k]; for (int k = 0; k < 4; ++k) T->q[k] = d[4 + k]; }
  *nc = size_t(c); *ns = size_t(s); *nf = size_t(f);
  return LIO_OK;
}

// ---------------------------------------------------------------- stateless blocks
struct Scratch {
  hipStream_t s = nullptr;
  DBuf<float4> a, b, c;
  VoxelGridDev vox;
  KnnGrid grid;
  DBuf<uint8_t> valid;
  DBuf<float4> coef;
  DBuf<float> score, tf;
  DBuf<int32_t> idx;
  DBuf<float> sqd;
  DBuf<float> bounds;
};
static Scratch &scratch() {
  static thread_local Scratch sc;
  if (!sc.s) {
    int nd = 0;
    LIO_HIP(hipGetDeviceCount(&nd));
    if (nd <= 0) throw DeviceError("no HIP device: the product has no CPU path");
    LIO_HIP(hipStreamCreate(&sc.s));
  }
  return sc;
}
static void host_bounds(const float *xyzi, size_t n, float mn[3], float mx[3]) {
  for (int d = 0; d < 3; ++d) { mn[d] = 3.4e38f; mx[d] = -3.4e38f; }
  for (size_t i = 0; i < n; ++i)
    for (int d = 0; d < 3; ++d) { float v = xyzi[4 * i + d]; if (std::isfinite(v)) { mn[d] = std::min(mn[d], v); mx[d] = std::max(mx[d], v); } }
}

int lio_voxel_grid(const float *xyzi, size_t n, float leaf, float *out, size_t *n_out) {
  if ((!xyzi && n) || !out || !n_out || !(leaf > 0)) return LIO_ERR_ARG;
  return guarded([&] {
    Scratch &sc = scratch();
    sc.a.reserve(std::max<size_t>(n, 1));
    if (n) LIO_HIP(hipMemcpyAsync(sc.a.p, xyzi, n * sizeof(float4), hipMemcpyHostToDevice, sc.s));
    size_t m = sc.vox.run(sc.a.p, n, leaf, sc.b, sc.s);
    if (m) LIO_HIP(hipMemcpyAsync(out, sc.b.p, m * sizeof(float4), hipMemcpyDeviceToHost, sc.s));
    LIO_HIP(hipStreamSynchronize(sc.s));
    *n_out = m;
    return LIO_OK;
  });
}

int lio_knn(const float *map, size_t n_map, const float *query, size_t m, int k, float radius_sq, int32_t *idx, float *sqd) {
  if ((!map && n_map) || (!query && m) || !(k == 1 || k == 5) || !idx || !sqd) return LIO_ERR_ARG;
  if (!(radius_sq > 0)) return LIO_ERR_ARG;  // the GPU search is radius-bounded by construction
  return guarded([&] {
    Scratch &sc = scratch();
    sc.a.reserve(std::max<size_t>(n_map, 1)); sc.b.reserve(std::max<size_t>(m, 1));
    sc.idx.reserve(std::max<size_t>(m * k, 1)); sc.sqd.reserve(std::max<size_t>(m * k, 1));
    if (n_map) LIO_HIP(hipMemcpyAsync(sc.a.p, map, n_map * sizeof(float4), hipMemcpyHostToDevice, sc.s));
    if (m) LIO_HIP(hipMemcpyAsync(sc.b.p, query, m * sizeof(float4), hipMemcpyHostToDevice, sc.s));
    float mn[3], mx[3];
    host_bounds(map, n_map, mn, mx);
    if (n_map == 0) { mn[0] = mn[1] = mn[2] = 0; mx[0] = mx[1] = mx[2] = 0; }
    sc.grid.build(sc.a.p, n_map, mn, mx, std::sqrt(radius_sq) * 1.0001f + 1e-6f, sc.s);
    launch_knn(sc.b.p, int(m), k, radius_sq, sc.grid.sorted(), sc.grid.cells(), sc.grid.desc(), sc.idx.p, sc.sqd.p, sc.s);
    if (m) {
      LIO_HIP(hipMemcpyAsync(idx, sc.idx.p, m * k * sizeof(int32_t), hipMemcpyDeviceToHost, sc.s));
      LIO_HIP(hipMemcpyAsync(sqd, sc.sqd.p, m * k * sizeof(float), hipMemcpyDeviceToHost, sc.s));
    }
    LIO_HIP(hipStreamSynchronize(sc.s));
    return LIO_OK;
  });
}

int lio_calculate_features(const float *map, size_t n_map, const float *stack, size_t m, const lio_transform_f *T, float mm, float mp,
                           uint8_t *valid, float *coeff, float *score) {
  if ((!map && n_map) || (!stack && m) || !T || !valid || !coeff || !score || !(mm > 0)) return LIO_ERR_ARG;
  return guarded([&] {
    Scratch &sc = scratch();
    sc.a.reserve(std::max<size_t>(n_map, 1)); sc.b.reserve(std::max<size_t>(m, 1));
    sc.valid.reserve(std::max<size_t>(m, 1)); sc.coef.reserve(std::max<size_t>(m, 1)); sc.score.reserve(std::max<size_t>(m, 1)); sc.tf.reserve(8);
    if (n_map) LIO_HIP(hipMemcpyAsync(sc.a.p, map, n_map * sizeof(float4), hipMemcpyHostToDevice, sc.s));
    if (m) LIO_HIP(hipMemcpyAsync(sc.b.p, stack, m * sizeof(float4), hipMemcpyHostToDevice, sc.s));
    float tf[8] = {T->q[0], T->q[1], T->q[2], T->q[3], T->p[0], T->p[1], T->p[2], 0.f};
    LIO_HIP(hipMemcpyAsync(sc.tf.p, tf, sizeof(tf), hipMemcpyHostToDevice, sc.s));
    float mn[3], mx[3];
    host_bounds(map, n_map, mn, mx);
    if (n_map == 0) { mn[0] = mn[1] = mn[2] = 0; mx[0] = mx[1] = mx[2] = 0; }
    sc.grid.build(sc.a.p, n_map, mn, mx, std::sqrt(mm) * 1.0001f + 1e-6f, sc.s);
    FeatArgs fa{};
    fa.nframes = 1; fa.max_M = int(m); fa.min_match_sq_dis = mm; fa.min_plane_dis = mp;
    fa.fr[0].stack = sc.b.p; fa.fr[0].M = int(m); fa.fr[0].slot_off = 0; fa.fr[0].tf_index = 0;
    launch_features(fa, sc.tf.p, sc.grid.sorted(), sc.grid.cells(), sc.grid.desc(), sc.valid.p, sc.coef.p, sc.score.p, nullptr, sc.s);
    if (m) {
      LIO_HIP(hipMemcpyAsync(valid, sc.valid.p, m, hipMemcpyDeviceToHost, sc.s));
      LIO_HIP(hipMemcpyAsync(coeff, sc.coef.p, m * sizeof(float4), hipMemcpyDeviceToHost, sc.s));
      LIO_HIP(hipMemcpyAsync(score, sc.score.p, m * sizeof(float), hipMemcpyDeviceToHost, sc.s));
    }
    LIO_HIP(hipStreamSynchronize(sc.s));
    return LIO_OK;
  });
}

// ---------------------------------------------------------------- pre-integration (host)
lio_pim *lio_pim_create(const double acc0[3], const double gyr0[3], const double ba[3], const double bg[3], double acc_n, double gyr_n,
                        double acc_w, double gyr_w, double g_norm) {
  if (!acc0 || !gyr0 || !ba || !bg) return nullptr;
  PimNoise n; n.acc_n = acc_n; n.gyr_n = gyr_n; n.acc_w = acc_w; n.gyr_w = gyr_w; n.g_norm = g_norm;
  lio_pim *h = new (std::nothrow) lio_pim;
  if (h) h->p = std::make_shared<Preintegration>(v3(acc0), v3(gyr0), v3(ba), v3(bg), n);
  return h;
}
void lio_pim_destroy(lio_pim *h) { delete h; }
int lio_pim_push_back(lio_pim *h, double dt, const double acc[3], const double gyr[3]) {
  if (!h || !acc || !gyr) return LIO_ERR_ARG;
  h->p->push_back(dt, v3(acc), v3(gyr));
  return LIO_OK;
}
int lio_pim_repropagate(lio_pim *h, const double ba[3], const double bg[3]) {
  if (!h || !ba || !bg) return LIO_ERR_ARG;
  h->p->repropagate(v3(ba), v3(bg));
  return LIO_OK;
}
int lio_pim_get(const lio_pim *h, double *sum_dt, double *dp, double *dq, double *dv, double *jac, double *cov) {
  if (!h) return LIO_ERR_ARG;
  const Preintegration &p = *h->p;
  if (sum_dt) *sum_dt = p.sum_dt;
  if (dp) { dp[0] = p.dp.x; dp[1] = p.dp.y; dp[2] = p.dp.z; }
  if (dq) { dq[0] = p.dq.x; dq[1] = p.dq.y; dq[2] = p.dq.z; dq[3] = p.dq.w; }
  if (dv) { dv[0] = p.dv.x; dv[1] = p.dv.y; dv[2] = p.dv.z; }
  if (jac) std::memcpy(jac, p.jac, sizeof(p.jac));
  if (cov) std::memcpy(cov, p.cov, sizeof(p.cov));
  return LIO_OK;
}
int lio_pim_evaluate(const lio_pim *h, const double *pi, const double *sbi, const double *pj, const double *sbj, double *res) {
  if (!h || !pi || !sbi || !pj || !sbj || !res) return LIO_ERR_ARG;
  V3d Pi, Pj; Qd Qi, Qj;
  unpack_pose(pi, Pi, Qi); unpack_pose(pj, Pj, Qj);
  h->p->evaluate(Pi, Qi, v3(sbi), v3(sbi + 3), v3(sbi + 6), Pj, Qj, v3(sbj), v3(sbj + 3), v3(sbj + 6), res);
  return LIO_OK;
}

// ---------------------------------------------------------------- factors (host)
int lio_factor_imu(const lio_pim *h, const double *pi, const double *sbi, const double *pj, const double *sbj, double *res, double *j0,
                   double *j1, double *j2, double *j3) {
  if (!h || !pi || !sbi || !pj || !sbj || !res) return LIO_ERR_ARG;
  return imu_factor(*h->p, pi, sbi, pj, sbj, res, j0, j1, j2, j3) ? LIO_OK : LIO_ERR_STATE;
}
int lio_factor_pivot_point_plane(const double point[3], const double coeff[4], const double *pp, const double *pi, const double *pex,
                                 double *res, double *j0, double *j1, double *j2) {
  if (!point || !coeff || !pp || !pi || !pex || !res) return LIO_ERR_ARG;
  ppp_factor(v3(point), coeff, pp, pi, pex, res, j0, j1, j2);
  return LIO_OK;
}
int lio_factor_prior(const double pos0[3], const double rot0[4], const double *pose, double *res, double *j) {
  if (!pos0 || !rot0 || !pose || !res) return LIO_ERR_ARG;
  prior_factor(v3(pos0), Qd(rot0[3], rot0[0], rot0[1], rot0[2]), pose, res, j);
  return LIO_OK;
}
int lio_pose_plus(const double *pose, const double *d, double *out) {
  if (!pose || !d || !out) return LIO_ERR_ARG;
  pose_plus(pose, d, out);
  return LIO_OK;
}

// ---------------------------------------------------------------- estimator
void lio_est_default_config(lio_est_config *c) {
  if (!c) return;
  std::memset(c, 0, sizeof(*c));
  c->window_size = 15; c->opt_window_size = 5; c->corner_filter_size = 0.2f; c->surf_filter_size = 0.4f;
  c->min_match_sq_dis = 1.0f; c->min_plane_dis = 0.2f;
  c->transform_lb.q[3] = 1.f; c->transform_lb.p[2] = -0.1f;
  c->opt_extrinsic = 0; c->imu_factor = 1; c->point_distance_factor = 0; c->prior_factor = 0; c->marginalization_factor = 1;
  c->enable_deskew = 1; c->cutoff_deskew = 0; c->keep_features = 0;
  c->acc_n = 0.1; c->gyr_n = 0.01; c->acc_w = 0.0002; c->gyr_w = 2.0e-5; c->g_norm = 9.805;
  c->max_num_iterations = 10; c->max_solver_time = 0.10; c->extrinsic_stage = 2; c->init_window_factor = 3;
}
lio_est *lio_est_create(const lio_est_config *c) {
  if (!c || c->window_size < 1 || c->opt_window_size < 1 || c->opt_window_size > c->window_size || c->window_size + 1 > LIO_MAX_FRAMES) return nullptr;
  lio_est *h = new (std::nothrow) lio_est;
  if (!h) return nullptr;
  EstConfig &e = h->cfg;
  e.W = c->window_size; e.Wo = c->opt_window_size;
  e.corner_filter_size = c->corner_filter_size; e.surf_filter_size = c->surf_filter_size;
  e.min_match_sq_dis = c->min_match_sq_dis; e.min_plane_dis = c->min_plane_dis;
  e.transform_lb = toT(c->transform_lb);
  e.opt_extrinsic = c->opt_extrinsic; e.imu_factor = c->imu_factor; e.point_distance_factor = c->point_distance_factor;
  e.prior_factor = c->prior_factor; e.marginalization_factor = c->marginalization_factor;
  e.enable_deskew = c->enable_deskew; e.cutoff_deskew = c->cutoff_deskew; e.keep_features = c->keep_features;
  e.pim.acc_n = c->acc_n; e.pim.gyr_n = c->gyr_n; e.pim.acc_w = c->acc_w; e.pim.gyr_w = c->gyr_w; e.pim.g_norm = c->g_norm;
  e.max_num_iterations = c->max_num_iterations; e.max_solver_time = c->max_solver_time; e.extrinsic_stage = c->extrinsic_stage;
  e.init_window_factor = c->init_window_factor > 0 ? c->init_window_factor : 1;
  e.device_solve = c->device_solve != 0; e.inline_marg = c->inline_marg != 0;
  for (const char *name : {"LIO_DEVICE_SOLVE", "LIO_DEVICE_MARG"}) if (const char *v = std::getenv(name)) { if (std::atoi(v) != 0) e.device_solve = true; else if (std::string(name) == "LIO_DEVICE_SOLVE") e.device_solve = false; }
  e.stream_sync = c->stream_sync != 0;
  e.moments_form = (c->moments_form == 1 || c->moments_form == 2) ? c->moments_form : 0;
  e.resident_moments = (c->resident_moments >= 1 && c->resident_moments <= 3) ? c->resident_moments : 0;
  // Estimator.cc:189-194: the estimator's filter sizes and thresholds configure its PointMapping base (created on first use)
  h->map_cfg.corner_filter_size = c->corner_filter_size; h->map_cfg.surf_filter_size = c->surf_filter_size;
  h->map_cfg.min_match_sq_dis = c->min_match_sq_dis; h->map_cfg.min_plane_dis = c->min_plane_dis; h->map_cfg.num_max_iterations = 10;
  h->map_cfg.map_builder = 0; h->map_cfg.enable_4d = 1; h->map_cfg.skip_count = 2;
  int rc = guarded([&] {
    h->e.reset(new Estimator(e));
    if (e.device_solve) {   // a batch of one window: the device loop and the device marginalization serve this handle's solves
      h->solo.reset(new EstimatorBatch({h->e.get()}));
      EstimatorBatch *b = h->solo.get();
      h->e->solve_hook_ = [b](lio_solve_report *rep) { b->Solve(rep); return b->window_ok(0); };
    }
    return LIO_OK;
  });
  if (rc != LIO_OK) { delete h; return nullptr; }
  return h;
}
static void dissolve_batch(lio_est_batch *B);
void lio_est_destroy(lio_est *h) {
  if (!h) return;
  if (h->owner) {   // destroyed before its batch: the batch goes first (it holds pointers into every member)
    std::fprintf(stderr, "[lio_hip] lio_est_destroy: the handle is a member of a batch; the batch is dissolved\n");
    dissolve_batch(h->owner);
  }
  delete h;
}

int lio_est_process_imu(lio_est *h, double dt, const double acc[3], const double gyr[3], double stamp) {
  if (!h || !acc || !gyr) return LIO_ERR_ARG;
  h->e->ProcessImu(dt, v3(acc), v3(gyr), stamp);
  return LIO_OK;
}
int lio_est_process_imu_batch(lio_est *h, size_t n, const double *dt, const double *acc, const double *gyr, const double *stamp) {
  if (!h || (n && (!dt || !acc || !gyr || !stamp))) return LIO_ERR_ARG;
  for (size_t k = 0; k < n; ++k) h->e->ProcessImu(dt[k], v3(acc + 3 * k), v3(gyr + 3 * k), stamp[k]);
  return LIO_OK;
}
int lio_est_process_laser_odom(lio_est *h, const lio_transform_f *T, const float *surf, size_t ns, const float *corner, size_t nc, double stamp,
                               lio_solve_report *rep) {
  if (!h || !T || (!surf && ns) || (!corner && nc)) return LIO_ERR_ARG;
  return guarded([&] { return h->e->ProcessLaserOdom(toT(*T), surf, ns, corner, nc, stamp, rep) ? LIO_OK : LIO_ERR_STATE; });
}
// ProcessCompactData (Estimator.cc:776-856).  The post-initialisation map-database refresh (:703-708) only feeds the
// published surround map and is not reproduced.
int lio_est_process_compact(lio_est *h, const float *data, size_t n, double stamp, lio_transform_f *T_out, lio_solve_report *rep) {
  if (!h || !data) return LIO_ERR_ARG;
  lio_transform_f Tsum;
  size_t nc = 0, ns = 0, nf = 0;
  int rc = lio_compact_decode(data, n, &Tsum, &nc, &ns, &nf);
  if (rc != LIO_OK) return rc;
  return guarded([&] {
    Estimator &e = *h->e;
    if (e.inited_ && !e.cfg_.imu_factor) return LIO_ERR_STATE;  // LOAM-only operation after init is not part of this path
    if (!h->map) h->map.reset(new MappingDev(h->map_cfg));
    MappingDev &m = *h->map;
    const float *corner = data + 4 * 3, *surf = data + 4 * (3 + nc);
    if (e.inited_) {  // :780-803: predict transform_tobe_mapped_ with the IMU-propagated body motion
      const int W = e.W_;
      auto bodyPose = [&](int i) {
        // Quaterniond(R).cast<float>(): build the quaternion in double, then cast
        const Quat<double> qd = fromRot(e.Rs_[i]);
        return Rigidf(Quat<float>(float(qd.w), float(qd.x), float(qd.y), float(qd.z)), Vec3<float>(float(e.Ps_[i].x), float(e.Ps_[i].y), float(e.Ps_[i].z)));
      };
      const Rigidf d_trans = compose(rinverse(bodyPose(W - 1)), bodyPose(W));
      m.transform_tobe_mapped_ = compose(compose(compose(m.transform_tobe_mapped_, e.transform_lb_), d_trans), rinverse(e.transform_lb_));
      m.transform_sum_ = toT(Tsum);
    } else {
      m.Process(corner, nc, surf, ns, toT(Tsum));
    }
    const Rigidf T_to_init = m.transform_aft_mapped_;
    if (T_out) fromT(T_to_init, T_out);
    const bool was_inited = e.inited_;
    const bool ok = was_inited ? e.ProcessLaserOdom(T_to_init, reinterpret_cast<const float4 *>(surf), ns, false, stamp, rep)
                               : e.ProcessLaserOdom(T_to_init, m.StackDevice(1), m.StackSize(1), true, stamp, rep);
    if (!ok) return LIO_ERR_STATE;
    if (!was_inited && e.inited_) m.imu_inited_ = true;  // SetInitFlag(true) (:545)
    return LIO_OK;
  });
}
int lio_est_get_stage(const lio_est *h, int *stage, int *cir_buf_count, int *extrinsic_stage, int *last_event, double *R_WI, double *g_vec) {
  if (!h) return LIO_ERR_ARG;
  const Estimator &e = *h->e;
  if (stage) *stage = e.inited_ ? 1 : 0;
  if (cir_buf_count) *cir_buf_count = e.cir_buf_count_;
  if (extrinsic_stage) *extrinsic_stage = e.extrinsic_stage_;
  if (last_event) *last_event = e.last_event_;
  if (R_WI) for (int k = 0; k < 9; ++k) R_WI[k] = e.R_WI_.m[k];
  if (g_vec) { g_vec[0] = e.g_vec_.x; g_vec[1] = e.g_vec_.y; g_vec[2] = e.g_vec_.z; }
  return LIO_OK;
}
int lio_est_push_frame(lio_est *h, const lio_transform_f *T, const float *surf, size_t ns, const float *corner, size_t nc, double stamp) {
  if (!h || !T || (!surf && ns) || (!corner && nc)) return LIO_ERR_ARG;
  if (!h->e->inited_) return LIO_ERR_STATE;
  return guarded([&] { return h->e->PushFrame(toT(*T), surf, ns, corner, nc, stamp) ? LIO_OK : LIO_ERR_STATE; });
}
int lio_est_solve_optimization(lio_est *h, lio_solve_report *rep) {
  if (!h) return LIO_ERR_ARG;
  if (!h->e->inited_) return LIO_ERR_STATE;
  return guarded([&] { return h->e->SolveOptimization(rep) ? LIO_OK : LIO_ERR_STATE; });
}
int lio_est_slide_window(lio_est *h) {
  if (!h) return LIO_ERR_ARG;
  if (!h->e->inited_) return LIO_ERR_STATE;
  return guarded([&] { h->e->SlideWindow(); return LIO_OK; });
}
int lio_est_sync(lio_est *h) {
  if (!h) return LIO_ERR_ARG;
  return guarded([&] { h->e->JoinMarg(); return LIO_OK; });
}
int lio_est_set_window(lio_est *h, int n, const double *Ps, const double *Rs, const double *Vs, const double *Bas, const double *Bgs,
                       const double g[3]) {
  if (!h || !Ps || !Rs || !Vs || !Bas || !Bgs || !g || n != h->e->W_ + 1) return LIO_ERR_ARG;
  h->e->SetWindow(Ps, Rs, Vs, Bas, Bgs, g);
  return LIO_OK;
}
int lio_est_get_window(const lio_est *h, int n, double *Ps, double *Rs, double *Vs, double *Bas, double *Bgs, lio_transform_f *Tlb) {
  if (!h || n != h->e->W_ + 1) return LIO_ERR_ARG;
  const Estimator &e = *h->e;
  for (int i = 0; i < n; ++i) {
    for (int k = 0; k < 3; ++k) {
      if (Ps) Ps[3 * i + k] = e.Ps_[i][k];
      if (Vs) Vs[3 * i + k] = e.Vs_[i][k];
      if (Bas) Bas[3 * i + k] = e.Bas_[i][k];
      if (Bgs) Bgs[3 * i + k] = e.Bgs_[i][k];
    }
    if (Rs) for (int k = 0; k < 9; ++k) Rs[9 * i + k] = e.Rs_[i].m[k];
  }
  if (Tlb) fromT(e.transform_lb_, Tlb);
  return LIO_OK;
}
int lio_est_set_surf_stack(lio_est *h, int frame, const float *xyzi, size_t n) {
  if (!h || frame < 0 || frame > h->e->W_ || (!xyzi && n)) return LIO_ERR_ARG;
  return guarded([&] { h->e->SetSurfStack(frame, xyzi, n); return LIO_OK; });
}
size_t lio_est_get_surf_stack(const lio_est *h, int frame, float *out) {
  if (!h || frame < 0 || frame > h->e->W_) return 0;
  size_t n = 0;
  guarded([&] { n = h->e->GetSurfStack(frame, out); return LIO_OK; });
  return n;
}
int lio_est_set_preintegration(lio_est *h, int frame, const double acc0[3], const double gyr0[3], const double ba[3], const double bg[3],
                               const double *dt, const double *acc, const double *gyr, size_t ns) {
  if (!h || frame < 0 || frame > h->e->W_ || !acc0 || !gyr0 || !ba || !bg || (ns && (!dt || !acc || !gyr))) return LIO_ERR_ARG;
  auto p = std::make_shared<Preintegration>(v3(acc0), v3(gyr0), v3(ba), v3(bg), h->cfg.pim);
  for (size_t k = 0; k < ns; ++k) p->push_back(dt[k], v3(acc + 3 * k), v3(gyr + 3 * k));
  h->e->SetPreintegration(frame, p);
  return LIO_OK;
}
int lio_est_begin_frame(lio_est *h, const double acc[3], const double gyr[3]) {
  if (!h || !acc || !gyr) return LIO_ERR_ARG;
  h->e->BeginFrame(v3(acc), v3(gyr));
  return LIO_OK;
}
int lio_est_build_local_map(lio_est *h) {
  if (!h) return LIO_ERR_ARG;
  if (!h->e->inited_) return LIO_ERR_STATE;
  return guarded([&] { h->e->BuildLocalMap(nullptr); return LIO_OK; });
}
size_t lio_est_get_local_map(const lio_est *h, float *out) {
  if (!h) return 0;
  size_t n = 0;
  guarded([&] { n = h->e->GetLocalMap(out); return LIO_OK; });
  return n;
}
size_t lio_est_get_features(const lio_est *h, int frame, double *pt, double *co, double *sc) {
  if (!h) return 0;
  size_t n = 0;
  guarded([&] { n = h->e->GetFeatures(frame, pt, co, sc); return LIO_OK; });
  return n;
}
int lio_est_get_laser_odom_transform(const lio_est *h, lio_transform_f *out) {
  if (!h || !out) return LIO_ERR_ARG;
  fromT(h->e->laser_odom_transform_, out);
  return LIO_OK;
}
int lio_est_get_prior(const lio_est *h, double *JtJ, double *Jtr, double *x0, int *x0_len) {
  if (!h) return LIO_ERR_ARG;
  return guarded([&] {
    const_cast<lio_est *>(h)->e->JoinMarg();
    const auto &pr = h->e->last_marg_;
    if (!pr) return 0;
    const int n = pr->n;
    if (JtJ) std::memcpy(JtJ, pr->JtJ.a.data(), sizeof(double) * size_t(n) * n);
    if (Jtr) std::memcpy(Jtr, pr->Jtr0.data(), sizeof(double) * n);
    int len = 0;
    for (const auto &b : pr->x0) { if (x0) for (double v : b) x0[len++] = v; else len += int(b.size()); }
    if (x0_len) *x0_len = len;
    return n;
  });
}
int lio_est_get_prior_factor(const lio_est *h, double *lin_jac, double *lin_res, double *x0, int *x0_len) {
  if (!h) return LIO_ERR_ARG;
  // JoinMarg rethrows a failed worker task: nothing may unwind through the C boundary
  return guarded([&] {
    const_cast<lio_est *>(h)->e->JoinMarg();
    const auto &pr = h->e->last_marg_;
    if (!pr) return 0;
    const int n = pr->n;
    if (lin_jac) std::memcpy(lin_jac, pr->lin_jac.a.data(), sizeof(double) * size_t(n) * n);
    if (lin_res) std::memcpy(lin_res, pr->lin_res.data(), sizeof(double) * n);
    int len = 0;
    for (const auto &b : pr->x0) { if (x0) for (double v : b) x0[len++] = v; else len += int(b.size()); }
    if (x0_len) *x0_len = len;
    return n;
  });
}
int lio_est_set_prior_factor(lio_est *h, int n, const double *lin_jac, const double *lin_res, const double *x0, int x0_len) {
  if (!h || !lin_jac || !lin_res || !x0 || n <= 0) return LIO_ERR_ARG;
  return guarded([&] {
    h->e->JoinMarg();
    const auto &old = h->e->last_marg_;
    if (!old || old->n != n) return LIO_ERR_STATE;
    int len = 0;
    for (const auto &b : old->x0) len += int(b.size());
    if (len != x0_len) return LIO_ERR_STATE;
    auto pr = std::make_shared<MargPrior>(*old);   // the old object may still be referenced by a snapshot
    std::memcpy(pr->lin_jac.a.data(), lin_jac, sizeof(double) * size_t(n) * n);
    std::memcpy(pr->lin_res.data(), lin_res, sizeof(double) * n);
    len = 0;
    for (auto &b : pr->x0) for (double &v : b) v = x0[len++];
    pr->finalize();
    h->e->last_marg_ = pr;
    return LIO_OK;
  });
}
int lio_est_set_extrinsic(lio_est *h, const lio_transform_f *T) {
  if (!h || !T) return LIO_ERR_ARG;
  return guarded([&] {
    h->e->JoinMarg();   // a marginalization in flight captured its own copy of the states; join so that nothing reads half a change
    h->e->transform_lb_ = toT(*T);
    return LIO_OK;
  });
}
int lio_dense_spd_solve(const double *A, const double *b, int n, double *x) {
  if (!A || !b || !x || n < 1 || n > 128) return LIO_ERR_ARG;
  return guarded([&] {
    const int ok = ldlt_solve_device(A, b, n, x, scratch().s);
    return ok == 1 ? LIO_OK : (ok == -2 ? LIO_ERR_ARG : LIO_ERR_STATE);
  });
}
int lio_marginalize_schur(const double *A, const double *b, int m, int n, double *lin_jac, double *lin_res, double *evals) {
  if (!A || !b || !lin_jac || !lin_res || m < 1 || m > MARG_MAX_M - 1 || n < 1 || n > MARG_MAX_N) return LIO_ERR_ARG;
  return guarded([&] {
    int dev = 0;
    LIO_HIP(hipGetDevice(&dev));
    MargSchurDev md(dev);
    return md.Run(A, b, m, n, 1e-8, lin_jac, lin_res, evals, nullptr) ? LIO_OK : LIO_ERR_ARG;
  });
}
int lio_est_snapshot(lio_est *h) {
  if (!h) return LIO_ERR_ARG;
  return guarded([&] { h->e->Snapshot(); return LIO_OK; });
}
int lio_est_restore(lio_est *h) {
  if (!h) return LIO_ERR_ARG;
  return guarded([&] { return h->e->Restore() ? LIO_OK : LIO_ERR_STATE; });
}

int lio_est_copy_snapshot(lio_est *dst, lio_est *src) {
  if (!dst || !src) return LIO_ERR_ARG;
  return guarded([&] { return dst->e->CopySnapshotOf(*src->e) ? LIO_OK : LIO_ERR_STATE; });
}

int lio_est_solve_restored(lio_est *h, int steps, lio_solve_report *rep) {
  if (!h || steps < 0) return LIO_ERR_ARG;
  for (int k = 0; k < steps; ++k) {
    int rc = lio_est_restore(h);
    if (rc != LIO_OK) return rc;
    rc = lio_est_solve_optimization(h, rep);
    if (rc != LIO_OK) return rc;
  }
  return LIO_OK;
}

// ---------------------------------------------------------------- batched windows
// (re)builds the batch's EstimatorBatch objects for its partition and re-applies the options set so far
static void batch_build_parts(lio_est_batch *h) {
  h->b.reset(); h->b2.reset();   // (a part's destructor brings its device-resident priors back to the host objects)
  const int n = int(h->members.size());
  static const int env_parts = [] { const char *e = std::getenv("LIO_BW_PARTS"); const int v = e ? std::atoi(e) : 0; return (v == 1 || v == 2) ? v : 0; }();
  const int want = h->parts_opt ? h->parts_opt : (env_parts ? env_parts : (n >= kBatchSplitFrom ? 2 : 1));
  const int parts = (want == 2 && n >= 2) ? 2 : 1;
  h->n1 = parts == 2 ? (n + 1) / 2 : n;
  std::vector<Estimator *> es;
  for (int i = 0; i < h->n1; ++i) es.push_back(h->members[size_t(i)]->e.get());
  h->b.reset(new EstimatorBatch(es));
  if (parts == 2) {
    es.clear();
    for (int i = h->n1; i < n; ++i) es.push_back(h->members[size_t(i)]->e.get());
    h->b2.reset(new EstimatorBatch(es));
  }
  bool groups_set = false;
  for (const auto &o : h->options) {
    if (o.first == "loop_groups" && o.second != 0) groups_set = true;
    h->b->SetOption(o.first.c_str(), o.second);
    if (h->b2) h->b2->SetOption(o.first.c_str(), o.second);
  }
  // two parts side by side are two launch chains already: one loop group each unless the caller asked for a number
  if (h->b2 && !groups_set) { h->b->SetOption("loop_groups", 1); h->b2->SetOption("loop_groups", 1); }
}
lio_est_batch *lio_est_batch_create(lio_est *const *windows, int n) {
  if (!windows || n < 1 || n > 65535) return nullptr;
  for (int i = 0; i < n; ++i) {
    if (!windows[i] || windows[i]->adopted || windows[i]->solo) return nullptr;
    for (int j = 0; j < i; ++j) if (windows[j] == windows[i]) return nullptr;
  }
  lio_est_batch *h = new (std::nothrow) lio_est_batch;
  if (!h) return nullptr;
  for (int i = 0; i < n; ++i) h->members.push_back(windows[i]);
  int rc = guarded([&] { batch_build_parts(h); return LIO_OK; });
  if (rc != LIO_OK) { (void)guarded([&] { h->b.reset(); h->b2.reset(); return LIO_OK; }); delete h; return nullptr; }
  for (int i = 0; i < n; ++i) { windows[i]->adopted = true; windows[i]->owner = h; }
  return h;
}
static void dissolve_batch(lio_est_batch *B) {
  (void)guarded([&] { B->b.reset(); B->b2.reset(); return LIO_OK; });
  for (lio_est *m : B->members) { m->adopted = false; m->owner = nullptr; }
  B->members.clear();
}
void lio_est_batch_destroy(lio_est_batch *h) {
  if (!h) return;
  dissolve_batch(h);
  delete h;
}
int lio_est_batch_size(const lio_est_batch *h) { return (h && h->b) ? h->b->size() + (h->b2 ? h->b2->size() : 0) : 0; }
int lio_est_batch_solve(lio_est_batch *h, lio_solve_report *reps) {
  if (!h) return LIO_ERR_ARG;
  if (!h->b) return LIO_ERR_STATE;
  for (lio_est *m : h->members) if (!m->e->inited_) return LIO_ERR_STATE;
  return guarded([&] { batch_for_parts(h, [&](EstimatorBatch &b, int w0) { b.Solve(reps ? reps + w0 : nullptr); }); return LIO_OK; });
}
int lio_est_batch_set_option(lio_est_batch *h, const char *name, int value) {
  if (!h || !name) return LIO_ERR_ARG;
  if (!h->b) return LIO_ERR_STATE;
  if (std::string(name) == "parts") {   // the partition: 0 by size (two parts from kBatchSplitFrom windows), 1, 2
    if (value < 0 || value > 2) return LIO_ERR_ARG;
    if (value == h->parts_opt) return LIO_OK;
    h->parts_opt = value;
    return guarded([&] { batch_build_parts(h); return LIO_OK; });
  }
  if (!h->b->SetOption(name, value)) return LIO_ERR_ARG;
  if (h->b2) h->b2->SetOption(name, value);
  bool found = false;
  for (auto &o : h->options) if (o.first == name) { o.second = value; found = true; }
  if (!found) h->options.emplace_back(name, value);
  if (h->b2 && std::string(name) == "loop_groups" && value == 0) { h->b->SetOption("loop_groups", 1); h->b2->SetOption("loop_groups", 1); }
  return LIO_OK;
}
int lio_est_batch_solve_restored(lio_est_batch *h, int steps, lio_solve_report *reps) {
  if (!h || steps < 0) return LIO_ERR_ARG;
  if (!h->b) return LIO_ERR_STATE;
  return guarded([&] {
    std::atomic<int> bad{0};
    batch_for_parts(h, [&](EstimatorBatch &b, int w0) {   // (every part loops on its own: nothing ties the parts' steps together)
      const int w1 = w0 + b.size();
      for (int k = 0; k < steps && !bad.load(); ++k) {
        for (int w = w0; w < w1; ++w) if (!h->members[size_t(w)]->e->Restore()) { bad.store(1); return; }
        for (int w = w0; w < w1; ++w) if (!h->members[size_t(w)]->e->inited_) { bad.store(1); return; }
        b.Solve(reps ? reps + w0 : nullptr);
      }
      b.Sync();
    });
    return bad.load() ? int(LIO_ERR_STATE) : int(LIO_OK);
  });
}
int lio_est_batch_sync(lio_est_batch *h) {
  if (!h) return LIO_ERR_ARG;
  if (!h->b) return LIO_ERR_STATE;
  return guarded([&] { h->b->Sync(); if (h->b2) h->b2->Sync(); return LIO_OK; });
}
int lio_seg_sort_pairs(const unsigned *keys, const unsigned *vals, size_t n_total, const int *seg_off, const int *seg_n, int nseg, int bits, int passes, unsigned *keys_out,
                       unsigned *vals_out) {
  return guarded([&] { return seg_sort_host_test(keys, vals, n_total, seg_off, seg_n, nseg, bits, passes, keys_out, vals_out) ? LIO_OK : LIO_ERR_ARG; });
}
int lio_est_batch_stage_digest(lio_est_batch *h, int stage, unsigned long long *out) {
  if (!h || !out || stage < 0 || stage > 9) return LIO_ERR_ARG;
  if (!h->b) return LIO_ERR_STATE;
  return guarded([&] { h->b->StageDigest(stage, out); if (h->b2) h->b2->StageDigest(stage, out + h->n1); return LIO_OK; });
}
int lio_est_batch_get_clock(const lio_est_batch *h, double *out) {
  if (!h || !out) return LIO_ERR_ARG;
  if (!h->b) return LIO_ERR_STATE;
  BatchClock c;
  const int rc = guarded([&] {
    c = const_cast<lio_est_batch *>(h)->b->clock();
    if (h->b2) {
      // two parts: host phases ran side by side (the longer one counts); device times, counts and launches add up — the parts' stages
      // overlap on the GPU, so the summed device times exceed the wall time and every rate derived from them is a lower bound
      const BatchClock d = const_cast<lio_est_batch *>(h)->b2->clock();
      c.describe = std::max(c.describe, d.describe); c.map = std::max(c.map, d.map); c.grid_features = std::max(c.grid_features, d.grid_features);
      c.pack = std::max(c.pack, d.pack); c.solve = std::max(c.solve, d.solve); c.finish = std::max(c.finish, d.finish);
      c.fallback = std::max(c.fallback, d.fallback); c.total = std::max(c.total, d.total);
      c.n_device += d.n_device; c.n_host += d.n_host; c.rounds = std::max(c.rounds, d.rounds);
      for (int k = 0; k < 6; ++k) c.dev[k] += d.dev[k];
      c.dev_marg_wait += d.dev_marg_wait;
      for (int k = 0; k < 3; ++k) { c.kernel_ms[k] += d.kernel_ms[k]; c.kernel_launches[k] += d.kernel_launches[k]; }
    }
    return LIO_OK;
  });
  if (rc != LIO_OK) return rc;
  for (int k = 0; k < 6; ++k) out[10 + k] = c.dev[k];
  out[0] = c.describe; out[1] = c.map; out[2] = c.grid_features; out[3] = c.pack; out[4] = c.solve; out[5] = c.finish; out[6] = c.fallback; out[7] = c.total;
  out[8] = c.n_device; out[9] = c.rounds;
  out[16] = c.dev_marg_wait;
  for (int k = 0; k < 3; ++k) { out[17 + k] = c.kernel_ms[k]; out[20 + k] = double(c.kernel_launches[k]); }
  return LIO_OK;
}

int lio_est_set_factor_sharding(lio_est *h, int rank, int world, lio_allreduce_fn fn, void *user) {
  if (!h || world < 1 || rank < 0 || rank >= world) return LIO_ERR_ARG;
  h->e->shard_rank_ = rank; h->e->shard_world_ = world; h->e->allreduce_ = fn; h->e->allreduce_user_ = user;
  h->e->rccl_comm_ = nullptr;   // the callback form replaces an RCCL communicator set earlier (which the caller may destroy now)
  return LIO_OK;
}
int lio_est_set_factor_sharding_rccl(lio_est *h, lio_rccl *comm) {
  if (!h) return LIO_ERR_ARG;
  if (!comm) { h->e->rccl_comm_ = nullptr; h->e->shard_rank_ = 0; h->e->shard_world_ = 1; return LIO_OK; }
  h->e->rccl_comm_ = rccl_raw_comm(comm);
  h->e->shard_rank_ = lio_rccl_rank(comm); h->e->shard_world_ = lio_rccl_world(comm);
  h->e->allreduce_ = nullptr; h->e->allreduce_user_ = nullptr;
  return LIO_OK;
}
int lio_bench_voxel_grid(const float *xyzi, size_t n, float leaf, int reps, double *avg_ms, size_t *n_out) {
  if (!xyzi || n == 0 || !(leaf > 0) || reps < 1 || !avg_ms) return LIO_ERR_ARG;
  return guarded([&] {
    hipStream_t s = nullptr;
    LIO_HIP(hipStreamCreate(&s));
    DBuf<float4> in, out;
    in.reserve(n);
    LIO_HIP(hipMemcpyAsync(in.p, xyzi, n * sizeof(float4), hipMemcpyHostToDevice, s));
    VoxelGridDev vox;
    size_t m = 0;
    for (int w = 0; w < 2; ++w) m = vox.run(in.p, n, leaf, out, s);   // warm-up: buffers, sort scratch
    hipEvent_t e0, e1;
    LIO_HIP(hipEventCreate(&e0)); LIO_HIP(hipEventCreate(&e1));
    LIO_HIP(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) m = vox.run(in.p, n, leaf, out, s);
    LIO_HIP(hipEventRecord(e1, s));
    LIO_HIP(hipStreamSynchronize(s));
    float ms = 0;
    LIO_HIP(hipEventElapsedTime(&ms, e0, e1));
    *avg_ms = double(ms) / reps;
    if (n_out) *n_out = m;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(s);
    return LIO_OK;
  });
}
int lio_est_bench_batched_moments(lio_est *h, int n_windows, int reps, double *avg_ms, double *bytes) {
  if (!h || n_windows < 1 || reps < 1) return LIO_ERR_ARG;
  return guarded([&] { return h->e->BenchBatchedMoments(n_windows, reps, avg_ms, bytes) ? LIO_OK : LIO_ERR_STATE; });
}
int lio_est_enable_kernel_timing(lio_est *h, int on) {
  if (!h) return LIO_ERR_ARG;
  h->e->ResidentLaunchTiming(on == -1);   // -1: only the resident moments kernel's launches are bracketed (it stays in use)
  h->e->timers_.on = on > 0;
  h->e->timers_.sample = on > 1 ? on : 1;
  h->e->timers_.reset();
  return LIO_OK;
}
int lio_est_get_kernel_timing(lio_est *h, const char *name, double *total_ms, double *bytes) {
  if (total_ms) *total_ms = 0;
  if (bytes) *bytes = 0;
  if (!h || !name) return 0;
  static const char *names[KT_COUNT] = {"features", "odom_features", "odom_rows", "odom_update", "moments", "voxel", "knn_grid", "concat"};
  if (std::strcmp(name, "moments_resident") == 0) {
    // the resident moments kernel cannot be bracketed by HIP events per pass (one launch serves a whole solve): its passes are
    // timed on the device's wall clock, doorbell copy seen -> sums posted, slowest frame; counted since the handle was created
    int passes = 0;
    const double us = h->e->ResidentBusyUs(&passes, bytes);
    if (total_ms) *total_ms = us * 1e-3;
    return passes;
  }
  if (std::strcmp(name, "moments_resident_launch") == 0) {   // dispatch-to-exit spans recorded under lio_est_enable_kernel_timing(-1)
    int n = 0;
    guarded([&] { n = h->e->ResidentLaunchStats(total_ms); return LIO_OK; });
    return n;
  }
  for (int k = 0; k < KT_COUNT; ++k)
    if (std::strcmp(name, names[k]) == 0) {
      const KernelTimers::Acc &a = h->e->timers_.acc[k];
      if (total_ms) *total_ms = a.ms;
      if (bytes) *bytes = a.bytes;
      return a.n;
    }
  return 0;
}

}  // extern "C"
