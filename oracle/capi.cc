// oracle/capi.cc — TEST INFRASTRUCTURE ONLY (CPU oracle).  Not part of the product.
// Implements include/lio_c.h on top of the CPU restatement so tests/ can drive both back ends
// through the same symbols.  Built by oracle/Makefile into oracle/liblio_oracle.so.
#include <cstring>
#include <new>

#include "../include/lio_c.h"
#include "estimator.h"
#include "imu_init.h"
#include "mapping.h"
#include "odometry.h"
#include "pointproc.h"

using namespace orc;

struct lio_map { PointMapping m; explicit lio_map(const MappingConfig &c) : m(c) {} };
struct lio_odom { PointOdometry o; lio_odom(float sp, int io, size_t it, bool nd) : o(sp, io, it, nd) {} };

struct lio_pp { PointProcessor pp; lio_pp(float a, float b, int r) : pp(a, b, r) {} };
struct lio_pim { IntegrationBase pim; lio_pim(const V3d &a, const V3d &g, const V3d &ba, const V3d &bg, const PimConfig &c) : pim(a, g, ba, bg, c) {} };
struct lio_est {
  Estimator est;
  std::unique_ptr<Estimator> snap;
  PointMapping map;  // the PointMapping base of the reference's Estimator (Estimator.h:110)
  bool adopted = false;   // member of a lio_est_batch (a handle is in at most one)
  struct lio_est_batch *owner = nullptr;
  explicit lio_est(const EstimatorConfig &c, const MappingConfig &m) : est(c), map(m) {}
};

static V3d v3(const double *p) { return V3d(p[0], p[1], p[2]); }
static Transformf toT(const lio_transform_f &t) { return Transformf(Q<float>(t.q[3], t.q[0], t.q[1], t.q[2]), V3<float>(t.p[0], t.p[1], t.p[2])); }
static void fromT(const Transformf &T, lio_transform_f *o) { o->q[0] = T.rot.x; o->q[1] = T.rot.y; o->q[2] = T.rot.z; o->q[3] = T.rot.w; o->p[0] = T.pos.x; o->p[1] = T.pos.y; o->p[2] = T.pos.z; }
static Cloud toCloud(const float *xyzi, size_t n) { Cloud c(n); if (n) std::memcpy(c.data(), xyzi, n * sizeof(P4)); return c; }

extern "C" {

const char *lio_backend(void) { return "oracle-cpu"; }

// ---------------------------------------------------------------- PointProcessor
void lio_pp_default_config(lio_pp_config *c) {
  if (!c) return;
  c->scan_period = 0.1; c->num_scan_subregions = 8; c->num_curvature_regions = 5; c->surf_curv_th = 0.1f;
  c->max_corner_sharp = 2; c->max_corner_less_sharp = 20; c->max_surf_flat = 4; c->less_flat_filter_size = 0.2f;
  c->infer_start_ori = 0; c->rad_diff = 0.2;
}
int lio_pp_check_config(float lo, float up, int rings, const lio_pp_config *) { return (rings <= 0 || !(up > lo)) ? LIO_ERR_ARG : LIO_OK; }
lio_pp *lio_pp_create(float lo, float up, int rings, const lio_pp_config *c) {
  if (rings <= 0 || !(up > lo)) return nullptr;
  lio_pp *h = new (std::nothrow) lio_pp(lo, up, rings);
  if (h && c) {
    PPConfig &k = h->pp.config_;
    k.scan_period = c->scan_period; k.num_scan_subregions = c->num_scan_subregions; k.num_curvature_regions = c->num_curvature_regions;
    k.surf_curv_th = c->surf_curv_th; k.max_corner_sharp = c->max_corner_sharp; k.max_corner_less_sharp = c->max_corner_less_sharp;
    k.max_surf_flat = c->max_surf_flat; k.less_flat_filter_size = c->less_flat_filter_size;
    k.infer_start_ori_ = c->infer_start_ori != 0; k.rad_diff = c->rad_diff;
  }
  return h;
}
void lio_pp_destroy(lio_pp *h) { delete h; }
float lio_pp_start_ori(const lio_pp *h) { return h ? h->pp.start_ori_ : std::nanf(""); }
int lio_pp_process(lio_pp *h, const float *xyzi, size_t n) {
  if (!h || (!xyzi && n)) return LIO_ERR_ARG;
  h->pp.Process(xyzi, n);
  return LIO_OK;
}
int lio_pp_process_async(lio_pp *h, const float *xyzi, size_t n) { return lio_pp_process(h, xyzi, n); }
int lio_pp_wait(lio_pp *h) { return h ? LIO_OK : LIO_ERR_ARG; }
int lio_pp_process_batch(lio_pp *const *handles, const float *const *xyzi, const size_t *n, int n_sweeps) {
  if (n_sweeps < 0 || (n_sweeps > 0 && (!handles || !xyzi || !n))) return LIO_ERR_ARG;
  for (int k = 0; k < n_sweeps; ++k) {
    if (!handles[k] || (!xyzi[k] && n[k])) return LIO_ERR_ARG;
    for (int j = 0; j < k; ++j) if (handles[j] == handles[k]) return LIO_ERR_ARG;
  }
  int rc = LIO_OK;
  for (int k = 0; k < n_sweeps && rc == LIO_OK; ++k) rc = lio_pp_process(handles[k], xyzi[k], n[k]);
  return rc;
}
int lio_pp_process_batch_device(lio_pp *const *handles, const float *const *xyzi, const size_t *n, int n_sweeps) {
  return lio_pp_process_batch(handles, xyzi, n, n_sweeps);   // (no device here: the pointers are host memory)
}
int lio_pp_process_rings_batch(lio_pp *const *handles, const float *const *xyzi, const uint16_t *const *ring, const size_t *n, int n_sweeps) {
  if (n_sweeps < 0 || (n_sweeps > 0 && (!handles || !xyzi || !ring || !n))) return LIO_ERR_ARG;
  for (int k = 0; k < n_sweeps; ++k) {
    if (!handles[k] || ((!xyzi[k] || !ring[k]) && n[k])) return LIO_ERR_ARG;
    for (int j = 0; j < k; ++j) if (handles[j] == handles[k]) return LIO_ERR_ARG;
  }
  for (int k = 0; k < n_sweeps; ++k) handles[k]->pp.Process(xyzi[k], n[k], ring[k]);
  return LIO_OK;
}
int lio_pp_process_rings(lio_pp *h, const float *xyzi, const uint16_t *ring, size_t n) {
  if (!h || ((!xyzi || !ring) && n)) return LIO_ERR_ARG;
  h->pp.Process(xyzi, n, ring);
  return LIO_OK;
}
static const Cloud *ppCloud(const lio_pp *h, int which) {
  switch (which) {
    case LIO_PP_RINGS: return &h->pp.cloud_rings;
    case LIO_PP_SHARP: return &h->pp.sharp;
    case LIO_PP_LESS_SHARP: return &h->pp.less_sharp;
    case LIO_PP_FLAT: return &h->pp.flat;
    case LIO_PP_LESS_FLAT: return &h->pp.less_flat;
  }
  return nullptr;
}
size_t lio_pp_count(const lio_pp *h, int which) { const Cloud *c = h ? ppCloud(h, which) : nullptr; return c ? c->size() : 0; }
int lio_pp_get_cloud(const lio_pp *h, int which, float *out) {
  const Cloud *c = h ? ppCloud(h, which) : nullptr;
  if (!c || !out) return LIO_ERR_ARG;
  if (!c->empty()) std::memcpy(out, c->data(), c->size() * sizeof(P4));
  return LIO_OK;
}
int lio_pp_get_indices(const lio_pp *h, int which, int32_t *ring, int32_t *idx) {
  if (!h || which < 1 || which > 3 || !ring || !idx) return LIO_ERR_ARG;
  for (size_t k = 0; k < h->pp.pick_ring[which].size(); ++k) { ring[k] = h->pp.pick_ring[which][k]; idx[k] = h->pp.pick_idx[which][k]; }
  return LIO_OK;
}
int lio_pp_get_ring_offsets(const lio_pp *h, int32_t *out) {
  if (!h || !out) return LIO_ERR_ARG;
  for (size_t k = 0; k < h->pp.ring_offsets.size(); ++k) out[k] = h->pp.ring_offsets[k];
  return LIO_OK;
}
int lio_pp_get_curvature(const lio_pp *h, float *curv, int32_t *mask) {
  if (!h) return LIO_ERR_ARG;
  if (curv) std::memcpy(curv, h->pp.curvature.data(), h->pp.curvature.size() * sizeof(float));
  if (mask) for (size_t k = 0; k < h->pp.mask.size(); ++k) mask[k] = h->pp.mask[k];
  return LIO_OK;
}

int lio_pp_get_ring_intensity(const lio_pp *h, float *out) {
  if (!h || !out) return LIO_ERR_ARG;
  std::memcpy(out, h->pp.intensity_rings.data(), h->pp.intensity_rings.size() * sizeof(float));
  return LIO_OK;
}

// ---------------------------------------------------------------- PointOdometry
lio_odom *lio_odom_create(float scan_period, int io_ratio, int max_iter, int no_deskew) {
  if (!(scan_period > 0) || max_iter < 1) return nullptr;
  return new (std::nothrow) lio_odom(scan_period, io_ratio, size_t(max_iter), no_deskew != 0);
}
void lio_odom_destroy(lio_odom *h) { delete h; }
int lio_odom_process(lio_odom *h, const float *sharp, size_t n_sharp, const float *less_sharp, size_t n_ls, const float *flat, size_t n_flat,
                     const float *less_flat, size_t n_lf, lio_transform_f *Tsum, lio_transform_f *Tes, int *iters, int *nsel) {
  if (!h || (!sharp && n_sharp) || (!less_sharp && n_ls) || (!flat && n_flat) || (!less_flat && n_lf)) return LIO_ERR_ARG;
  h->o.Process(toCloud(sharp, n_sharp), toCloud(less_sharp, n_ls), toCloud(flat, n_flat), toCloud(less_flat, n_lf));
  if (Tsum) fromT(h->o.transform_sum_, Tsum);
  if (Tes) fromT(h->o.transform_es_, Tes);
  if (iters) *iters = h->o.iterations_done_;
  if (nsel) *nsel = h->o.last_num_sel_;
  return LIO_OK;
}
int lio_odom_get_iteration_trace(const lio_odom *h, lio_transform_f *trace, int capacity, int *kz) {
  if (!h || capacity < 0 || (!trace && capacity)) return LIO_ERR_ARG;
  const int n = int(h->o.es_trace_.size());
  for (int k = 0; k < n && k < capacity; ++k) fromT(h->o.es_trace_[size_t(k)], &trace[k]);
  if (kz) *kz = h->o.last_kz_;
  return n;
}
int lio_odom_enable(lio_odom *h, int on) {
  if (!h) return LIO_ERR_ARG;
  h->o.enable_odom_ = on != 0;
  return LIO_OK;
}
size_t lio_odom_get_last_cloud(const lio_odom *h, int which, float *out) {
  if (!h || which < 0 || which > 1) return 0;
  const Cloud &c = which == 0 ? h->o.last_corner_ : h->o.last_surf_;
  if (out && !c.empty()) std::memcpy(out, c.data(), c.size() * sizeof(P4));
  return c.size();
}

// ---------------------------------------------------------------- ImuInitializer
static bool gatherLaserTransforms(size_t n, const lio_transform_f *T, lio_pim *const *pims, std::vector<LaserTransform> &all) {
  all.resize(n);
  for (size_t i = 0; i < n; ++i) {
    all[i].transform = toT(T[i]);
    if (pims[i]) all[i].pre_integration = std::shared_ptr<IntegrationBase>(&pims[i]->pim, [](IntegrationBase *) {});
    else if (i != 0) return false;
  }
  return true;
}
int lio_imu_estimate_extrinsic_rotation(size_t n, const lio_transform_f *T, lio_pim *const *pims, lio_transform_f *lb) {
  if (n < 2 || !T || !pims || !lb) return LIO_ERR_ARG;
  std::vector<LaserTransform> all;
  if (!gatherLaserTransforms(n, T, pims, all)) return LIO_ERR_ARG;
  Transformf tlb = toT(*lb);
  bool ok = EstimateExtrinsicRotation(all, tlb);
  fromT(tlb, lb);
  return ok ? 1 : 0;
}
int lio_imu_initialization(size_t n, const lio_transform_f *T, lio_pim *const *pims, const lio_transform_f *lb, double *Vs, double *Bgs, double g[3],
                           double R_WI[9]) {
  if (n < 2 || !T || !pims || !lb || !Vs || !Bgs || !g || !R_WI) return LIO_ERR_ARG;
  std::vector<LaserTransform> all;
  if (!gatherLaserTransforms(n, T, pims, all)) return LIO_ERR_ARG;
  std::vector<V3d> vs(n), bgs(n);
  for (size_t i = 0; i < n; ++i) bgs[i] = v3(Bgs + 3 * i);
  V3d gv;
  M3d R = M3d::Identity();
  bool ok = Initialization(all, vs, bgs, gv, toT(*lb), R);
  for (size_t i = 0; i < n; ++i) {
    for (int d = 0; d < 3; ++d) { Vs[3 * i + d] = vs[i][d]; Bgs[3 * i + d] = bgs[i][d]; }
  }
  for (int d = 0; d < 3; ++d) g[d] = gv[d];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R_WI[r * 3 + c] = R(r, c);
  return ok ? 1 : 0;
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
  MappingConfig mc;
  mc.corner_filter_size = cfg.corner_filter_size; mc.surf_filter_size = cfg.surf_filter_size;
  mc.min_match_sq_dis = cfg.min_match_sq_dis; mc.min_plane_dis = cfg.min_plane_dis; mc.num_max_iterations = cfg.num_max_iterations;
  mc.map_builder = cfg.map_builder != 0; mc.enable_4d = cfg.enable_4d != 0; mc.skip_count = cfg.skip_count;
  return new (std::nothrow) lio_map(mc);
}
void lio_map_destroy(lio_map *h) { delete h; }
int lio_map_process(lio_map *h, const float *corner, size_t nc, const float *surf, size_t ns, const lio_transform_f *Tsum, lio_transform_f *Taft,
                    int *iters, int *nsel) {
  if (!h || !Tsum || (!corner && nc) || (!surf && ns)) return LIO_ERR_ARG;
  h->m.Process(toCloud(corner, nc), toCloud(surf, ns), toT(*Tsum));
  if (Taft) fromT(h->m.transform_aft_mapped, Taft);
  if (iters) *iters = h->m.last_iterations;
  if (nsel) *nsel = h->m.last_selected;
  return LIO_OK;
}
int lio_map_get_degeneracy(const lio_map *h, int *kz) {
  if (!h) return LIO_ERR_ARG;
  if (kz) *kz = h->m.last_kz;
  return h->m.last_degenerate ? 1 : 0;
}
int lio_map_set_init_flag(lio_map *h, int on) {
  if (!h) return LIO_ERR_ARG;
  h->m.imu_inited = on != 0;
  return LIO_OK;
}
int lio_map_set_transform_tobe_mapped(lio_map *h, const lio_transform_f *T) {
  if (!h || !T) return LIO_ERR_ARG;
  h->m.transform_tobe_mapped = toT(*T);
  return LIO_OK;
}
int lio_map_get_transform_tobe_mapped(const lio_map *h, lio_transform_f *T) {
  if (!h || !T) return LIO_ERR_ARG;
  fromT(h->m.transform_tobe_mapped, T);
  return LIO_OK;
}
int lio_map_update_map_database(lio_map *h, const float *corner, size_t nc, const float *surf, size_t ns, const uint32_t *valid, size_t nv,
                                const lio_transform_f *T, const int cen[3]) {
  if (!h || !T || !cen || (!corner && nc) || (!surf && ns) || (!valid && nv)) return LIO_ERR_ARG;
  std::vector<size_t> vi(nv);
  for (size_t i = 0; i < nv; ++i) {
    if (valid[i] >= uint32_t(PointMapping::L * PointMapping::Wd * PointMapping::H)) return LIO_ERR_ARG;
    vi[i] = valid[i];
  }
  h->m.UpdateMapDatabase(toCloud(corner, nc), toCloud(surf, ns), vi, toT(*T), cen);
  return LIO_OK;
}
size_t lio_map_get_cloud(const lio_map *h, int which, float *out) {
  if (!h || which < 0 || which > 3) return 0;
  const Cloud *c[4] = {&h->m.corner_stack_ds, &h->m.surf_stack_ds, &h->m.corner_from_map, &h->m.surf_from_map};
  if (out && !c[which]->empty()) std::memcpy(out, c[which]->data(), c[which]->size() * sizeof(P4));
  return c[which]->size();
}
size_t lio_map_get_cube(const lio_map *h, int cls, uint32_t idx, float *out) {
  if (!h || cls < 0 || cls > 1 || idx >= uint32_t(PointMapping::L * PointMapping::Wd * PointMapping::H)) return 0;
  const Cloud &c = cls == 0 ? h->m.corner_array[idx] : h->m.surf_array[idx];
  if (out && !c.empty()) std::memcpy(out, c.data(), c.size() * sizeof(P4));
  return c.size();
}
size_t lio_map_get_cube_state(const lio_map *h, int cen[3], uint32_t *valid) {
  if (!h) return 0;
  if (cen) for (int d = 0; d < 3; ++d) cen[d] = h->m.cen[d];
  if (valid) for (size_t i = 0; i < h->m.valid_idx.size(); ++i) valid[i] = uint32_t(h->m.valid_idx[i]);
  return h->m.valid_idx.size();
}
size_t lio_map_get_score_point_coeff(const lio_map *h, float *score, float *point, float *coeff) {
  if (!h) return 0;
  const auto &v = h->m.score_point_coeff;
  for (size_t i = 0; i < v.size(); ++i) {
    if (score) score[i] = v[i].score;
    if (point) std::memcpy(point + 4 * i, &v[i].point, sizeof(P4));
    if (coeff) std::memcpy(coeff + 4 * i, &v[i].coeff, sizeof(P4));
  }
  return v.size();
}

// ---------------------------------------------------------------- batched keyframe refinement (sequential restatement)
struct lio_kf_batch {
  MappingConfig cfg;
  struct Map { Cloud corner, surf; };
  struct Kf { int map; Cloud corner, surf; Transformf T; };
  std::vector<Map> maps;
  std::vector<Kf> kfs;
  std::vector<int32_t> last_kz;   // per keyframe, of the last refine
};
lio_kf_batch *lio_kf_batch_create(const lio_map_config *c) {
  lio_map_config cfg;
  if (c) cfg = *c; else lio_map_default_config(&cfg);
  if (cfg.num_max_iterations < 1) return nullptr;
  lio_kf_batch *h = new (std::nothrow) lio_kf_batch;
  if (!h) return nullptr;
  h->cfg.corner_filter_size = cfg.corner_filter_size; h->cfg.surf_filter_size = cfg.surf_filter_size;
  h->cfg.min_match_sq_dis = cfg.min_match_sq_dis; h->cfg.min_plane_dis = cfg.min_plane_dis; h->cfg.num_max_iterations = cfg.num_max_iterations;
  h->cfg.map_builder = cfg.map_builder != 0; h->cfg.enable_4d = cfg.enable_4d != 0; h->cfg.skip_count = cfg.skip_count;
  return h;
}
void lio_kf_batch_destroy(lio_kf_batch *h) { delete h; }
int lio_kf_batch_add_map(lio_kf_batch *h, const float *corner, size_t nc, const float *surf, size_t ns) {
  if (!h || (!corner && nc) || (!surf && ns)) return LIO_ERR_ARG;
  h->maps.push_back({toCloud(corner, nc), toCloud(surf, ns)});
  return int(h->maps.size()) - 1;
}
int lio_kf_batch_add_keyframe(lio_kf_batch *h, int map, const float *corner, size_t nc, const float *surf, size_t ns, const lio_transform_f *T) {
  if (!h || !T || (!corner && nc) || (!surf && ns) || map < 0 || size_t(map) >= h->maps.size()) return LIO_ERR_ARG;
  h->kfs.push_back({map, toCloud(corner, nc), toCloud(surf, ns), toT(*T)});
  return int(h->kfs.size()) - 1;
}
int lio_kf_batch_clear_keyframes(lio_kf_batch *h) {
  if (!h) return LIO_ERR_ARG;
  h->kfs.clear();
  return LIO_OK;
}
size_t lio_kf_batch_size(const lio_kf_batch *h) { return h ? h->kfs.size() : 0; }
int lio_kf_batch_refine(lio_kf_batch *h, lio_transform_f *T_out, int32_t *iters, int32_t *rows, double *device_ms) {
  if (!h) return LIO_ERR_ARG;
  if (device_ms) *device_ms = 0;
  const bool four_dof = h->cfg.map_builder && h->cfg.enable_4d;
  h->last_kz.assign(h->kfs.size(), 0);
  for (size_t k = 0; k < h->kfs.size(); ++k) {
    const auto &kf = h->kfs[k];
    KeyframeRefinement r = RefineKeyframe(h->cfg, h->maps[size_t(kf.map)].corner, h->maps[size_t(kf.map)].surf, kf.corner, kf.surf, kf.T, four_dof);
    if (T_out) fromT(r.T, &T_out[k]);
    if (iters) iters[k] = r.iterations;
    if (rows) rows[k] = r.selected;
    h->last_kz[k] = r.kz;
  }
  return LIO_OK;
}
int lio_kf_batch_get_degeneracy(const lio_kf_batch *h, int32_t *kz_out) {
  if (!h || !kz_out) return LIO_ERR_ARG;
  if (h->last_kz.size() != h->kfs.size()) return LIO_ERR_STATE;   // no refine since the keyframe list changed
  for (size_t k = 0; k < h->last_kz.size(); ++k) kz_out[k] = h->last_kz[k];
  return LIO_OK;
}

// ---------------------------------------------------------------- /compact_data codec
size_t lio_compact_encode(const lio_transform_f *T, const float *corner, size_t nc, const float *surf, size_t ns, const float *full, size_t nf,
                          float *out) {
  if (!T || !out || (!corner && nc) || (!surf && ns) || (!full && nf)) return 0;
  // PointOdometry.cc:737-757: one PointT is reused, so point[2] inherits intensity = qw
  float hdr[12] = {T->p[0], T->p[1], T->p[2], 0.f, T->q[0], T->q[1], T->q[2], T->q[3], float(nc), float(ns), float(nf), T->q[3]};
  std::memcpy(out, hdr, sizeof(hdr));
  float *o = out + 12;
  if (nc) std::memcpy(o, corner, nc * 16);
  o += 4 * nc;
  if (ns) std::memcpy(o, surf, ns * 16);
  o += 4 * ns;
  if (nf) std::memcpy(o, full, nf * 16);
  return 3 + nc + ns + nf;
}
int lio_compact_decode(const float *d, size_t n, lio_transform_f *T, size_t *nc, size_t *ns, size_t *nf) {
  if (!d || !nc || !ns || !nf) return LIO_ERR_ARG;
  if (n < 4) return LIO_ERR_ARG;  // PointMapping.cc:180-183
  for (int k = 8; k <= 10; ++k)
    if (!(d[k] >= 0.0f && d[k] <= float(n))) return LIO_ERR_ARG;  // wire floats: NaN / Inf / out-of-range sizes are refused
  int c = int(d[8]), s = int(d[9]), f = int(d[10]);
  if (size_t(3) + size_t(c) + size_t(s) + size_t(f) != n) return LIO_ERR_ARG;  // :191-195
  if (T) { T->p[0] = d[0]; T->p[1] = d[1]; T->p[2] = d[2]; T->q[0] = d[4]; T->q[1] = d[5]; T->q[2] = d[6]; T->q[3] = d[7]; }
  *nc = size_t(c); *ns = size_t(s); *nf = size_t(f);
  return LIO_OK;
}

// ---------------------------------------------------------------- stateless blocks
int lio_voxel_grid(const float *xyzi, size_t n, float leaf, float *out, size_t *n_out) {
  if ((!xyzi && n) || !out || !n_out || !(leaf > 0)) return LIO_ERR_ARG;
  Cloud in = toCloud(xyzi, n), o;
  VoxelGrid(in, leaf, o);
  if (!o.empty()) std::memcpy(out, o.data(), o.size() * sizeof(P4));
  *n_out = o.size();
  return LIO_OK;
}
int lio_knn(const float *map, size_t n_map, const float *query, size_t m, int k, float radius_sq, int32_t *idx, float *sqd) {
  if ((!map && n_map) || (!query && m) || k <= 0 || k > 16 || !idx || !sqd) return LIO_ERR_ARG;
  Cloud c = toCloud(map, n_map);
  KdTree t;
  t.Build(c);
  for (size_t i = 0; i < m; ++i) {
    P4 q{query[4 * i], query[4 * i + 1], query[4 * i + 2], 0};
    int id[16]; float sd[16];
    int f = t.Search(q, k, id, sd);
    for (int j = 0; j < k; ++j) {
      bool ok = j < f && !(radius_sq > 0 && !(sd[j] < radius_sq));
      idx[i * k + j] = ok ? id[j] : -1;
      sqd[i * k + j] = ok ? sd[j] : std::numeric_limits<float>::infinity();
    }
  }
  return LIO_OK;
}
int lio_calculate_features(const float *map, size_t n_map, const float *stack, size_t m, const lio_transform_f *T, float mm, float mp,
                           uint8_t *valid, float *coeff, float *score) {
  if ((!map && n_map) || (!stack && m) || !T || !valid || !coeff || !score) return LIO_ERR_ARG;
  Cloud c = toCloud(map, n_map), s = toCloud(stack, m);
  KdTree t;
  t.Build(c);
  std::vector<PlaneFeature> feats;
  std::vector<uint8_t> v;
  std::vector<std::array<float, 5>> raw;
  Estimator::CalculateFeatures(t, c, s, toT(*T), mm, mp, false, feats, &v, &raw);
  for (size_t i = 0; i < m; ++i) { valid[i] = v[i]; for (int k = 0; k < 4; ++k) coeff[4 * i + k] = raw[i][k]; score[i] = raw[i][4]; }
  return LIO_OK;
}

// ---------------------------------------------------------------- pre-integration
lio_pim *lio_pim_create(const double acc0[3], const double gyr0[3], const double ba[3], const double bg[3], double acc_n, double gyr_n,
                        double acc_w, double gyr_w, double g_norm) {
  if (!acc0 || !gyr0 || !ba || !bg) return nullptr;
  PimConfig c; c.acc_n = acc_n; c.gyr_n = gyr_n; c.acc_w = acc_w; c.gyr_w = gyr_w; c.g_norm = g_norm;
  return new (std::nothrow) lio_pim(v3(acc0), v3(gyr0), v3(ba), v3(bg), c);
}
void lio_pim_destroy(lio_pim *h) { delete h; }
int lio_pim_push_back(lio_pim *h, double dt, const double acc[3], const double gyr[3]) {
  if (!h || !acc || !gyr) return LIO_ERR_ARG;
  h->pim.push_back(dt, v3(acc), v3(gyr));
  return LIO_OK;
}
int lio_pim_repropagate(lio_pim *h, const double ba[3], const double bg[3]) {
  if (!h || !ba || !bg) return LIO_ERR_ARG;
  h->pim.Repropagate(v3(ba), v3(bg));
  return LIO_OK;
}
int lio_pim_get(const lio_pim *h, double *sum_dt, double *dp, double *dq, double *dv, double *jac, double *cov) {
  if (!h) return LIO_ERR_ARG;
  const IntegrationBase &p = h->pim;
  if (sum_dt) *sum_dt = p.sum_dt_;
  if (dp) { dp[0] = p.delta_p_.x; dp[1] = p.delta_p_.y; dp[2] = p.delta_p_.z; }
  if (dq) { dq[0] = p.delta_q_.x; dq[1] = p.delta_q_.y; dq[2] = p.delta_q_.z; dq[3] = p.delta_q_.w; }
  if (dv) { dv[0] = p.delta_v_.x; dv[1] = p.delta_v_.y; dv[2] = p.delta_v_.z; }
  if (jac) std::memcpy(jac, p.jacobian_.a.data(), 225 * sizeof(double));
  if (cov) std::memcpy(cov, p.covariance_.a.data(), 225 * sizeof(double));
  return LIO_OK;
}
int lio_pim_evaluate(const lio_pim *h, const double *pi, const double *sbi, const double *pj, const double *sbj, double *res) {
  if (!h || !pi || !sbi || !pj || !sbj || !res) return LIO_ERR_ARG;
  V3d Pi, Pj; Qd Qi, Qj;
  unpackPose(pi, Pi, Qi); unpackPose(pj, Pj, Qj);
  h->pim.Evaluate(Pi, Qi, v3(sbi), v3(sbi + 3), v3(sbi + 6), Pj, Qj, v3(sbj), v3(sbj + 3), v3(sbj + 6), res);
  return LIO_OK;
}

// ---------------------------------------------------------------- factors
int lio_factor_imu(const lio_pim *h, const double *pi, const double *sbi, const double *pj, const double *sbj, double *res, double *j0,
                   double *j1, double *j2, double *j3) {
  if (!h || !pi || !sbi || !pj || !sbj || !res) return LIO_ERR_ARG;
  const double *par[4] = {pi, sbi, pj, sbj};
  double *jac[4] = {j0, j1, j2, j3};
  bool any = j0 || j1 || j2 || j3;
  return ImuFactorEvaluate(h->pim, par, res, any ? jac : nullptr) ? LIO_OK : LIO_ERR_STATE;
}
int lio_factor_pivot_point_plane(const double point[3], const double coeff[4], const double *pp, const double *pi, const double *pex,
                                 double *res, double *j0, double *j1, double *j2) {
  if (!point || !coeff || !pp || !pi || !pex || !res) return LIO_ERR_ARG;
  const double *par[3] = {pp, pi, pex};
  double *jac[3] = {j0, j1, j2};
  bool any = j0 || j1 || j2;
  PivotPointPlaneEvaluate(v3(point), coeff, par, res, any ? jac : nullptr);
  return LIO_OK;
}
int lio_factor_prior(const double pos0[3], const double rot0[4], const double *pose, double *res, double *j) {
  if (!pos0 || !rot0 || !pose || !res) return LIO_ERR_ARG;
  PriorFactorEvaluate(v3(pos0), Qd(rot0[3], rot0[0], rot0[1], rot0[2]), pose, res, j);
  return LIO_OK;
}
int lio_pose_plus(const double *pose, const double *d, double *out) {
  if (!pose || !d || !out) return LIO_ERR_ARG;
  PosePlus(pose, d, out);
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
  if (!c || c->window_size < 1 || c->opt_window_size < 1 || c->opt_window_size > c->window_size) return nullptr;
  EstimatorConfig e;
  e.window_size = c->window_size; e.opt_window_size = c->opt_window_size;
  e.corner_filter_size = c->corner_filter_size; e.surf_filter_size = c->surf_filter_size;
  e.min_match_sq_dis = c->min_match_sq_dis; e.min_plane_dis = c->min_plane_dis;
  e.transform_lb = toT(c->transform_lb);
  e.opt_extrinsic = c->opt_extrinsic; e.imu_factor = c->imu_factor; e.point_distance_factor = c->point_distance_factor;
  e.prior_factor = c->prior_factor; e.marginalization_factor = c->marginalization_factor;
  e.enable_deskew = c->enable_deskew; e.cutoff_deskew = c->cutoff_deskew; e.keep_features = c->keep_features;
  e.pim.acc_n = c->acc_n; e.pim.gyr_n = c->gyr_n; e.pim.acc_w = c->acc_w; e.pim.gyr_w = c->gyr_w; e.pim.g_norm = c->g_norm;
  e.max_num_iterations = c->max_num_iterations; e.max_solver_time = c->max_solver_time; e.extrinsic_stage = c->extrinsic_stage;
  e.init_window_factor = c->init_window_factor > 0 ? c->init_window_factor : 1;
  MappingConfig m;  // Estimator.cc:189-194: the estimator's filter sizes and thresholds configure the PointMapping base
  m.corner_filter_size = c->corner_filter_size; m.surf_filter_size = c->surf_filter_size;
  m.min_match_sq_dis = c->min_match_sq_dis; m.min_plane_dis = c->min_plane_dis;
  return new (std::nothrow) lio_est(e, m);
}
static void dissolve_batch(struct lio_est_batch *b);
void lio_est_destroy(lio_est *h) {
  if (h && h->owner) dissolve_batch(h->owner);   // (include/lio_c.h: an adopted handle's batch is dissolved first)
  delete h;
}

int lio_est_process_imu(lio_est *h, double dt, const double acc[3], const double gyr[3], double stamp) {
  if (!h || !acc || !gyr) return LIO_ERR_ARG;
  h->est.ProcessImu(dt, v3(acc), v3(gyr), stamp);
  return LIO_OK;
}
int lio_est_process_imu_batch(lio_est *h, size_t n, const double *dt, const double *acc, const double *gyr, const double *stamp) {
  if (!h || (n && (!dt || !acc || !gyr || !stamp))) return LIO_ERR_ARG;
  for (size_t k = 0; k < n; ++k) h->est.ProcessImu(dt[k], v3(acc + 3 * k), v3(gyr + 3 * k), stamp[k]);
  return LIO_OK;
}
static void fillReport(const SolveReport &R, lio_solve_report *o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->iterations = R.iterations; o->successful_steps = R.successful; o->termination = R.termination;
  o->n_lidar_residuals = R.n_lidar; o->n_local_map = R.n_local_map; o->laser_odom_iterations = R.laser_odom_iters; o->laser_odom_kz = R.laser_odom_kz;
  o->turn_off = R.turn_off; o->convergence_flag = R.convergence_flag; o->marginalized = R.marginalized;
  o->cost_pim_before = R.cost_pim; o->cost_ppp_before = R.cost_ppp; o->cost_marg_before = R.cost_marg;
  o->initial_cost = R.initial_cost; o->final_cost = R.final_cost;
  for (size_t k = 0; k < R.trace.size() && k < 32; ++k) o->cost_trace[k] = R.trace[k];
  o->ms_build_map = R.ms_build_map; o->ms_features = R.ms_features; o->ms_prepare = R.ms_prepare; o->ms_opt = R.ms_opt;
  o->ms_marg = R.ms_marg; o->ms_total = R.ms_total;
}
int lio_est_process_laser_odom(lio_est *h, const lio_transform_f *T, const float *surf, size_t ns, const float *corner, size_t nc, double stamp,
                               lio_solve_report *rep) {
  if (!h || !T || (!surf && ns) || (!corner && nc)) return LIO_ERR_ARG;
  SolveReport R;
  if (!h->est.ProcessLaserOdom(toT(*T), toCloud(surf, ns), toCloud(corner, nc), stamp, &R)) return LIO_ERR_STATE;
  fillReport(R, rep);
  return LIO_OK;
}
// ProcessCompactData (Estimator.cc:776-856).  The post-initialisation map-database refresh (:703-708) only feeds the
// published surround map and is not reproduced.
int lio_est_process_compact(lio_est *h, const float *data, size_t n, double stamp, lio_transform_f *T_out, lio_solve_report *rep) {
  if (!h || !data) return LIO_ERR_ARG;
  lio_transform_f Tsum;
  size_t nc = 0, ns = 0, nf = 0;
  int rc = lio_compact_decode(data, n, &Tsum, &nc, &ns, &nf);
  if (rc != LIO_OK) return rc;
  Estimator &e = h->est;
  PointMapping &m = h->map;
  if (e.inited && !e.cfg.imu_factor) return LIO_ERR_STATE;  // LOAM-only operation after init is not part of this path
  Cloud corner = toCloud(data + 4 * 3, nc), surf = toCloud(data + 4 * (3 + nc), ns);
  if (e.inited) {  // :780-803: predict transform_tobe_mapped_ with the IMU-propagated body motion
    const int W = e.W;
    Transformf prev(Q<double>::FromMatrix(e.Rs[W - 1]).cast<float>(), e.Ps[W - 1].cast<float>());
    Transformf curr(Q<double>::FromMatrix(e.Rs[W]).cast<float>(), e.Ps[W].cast<float>());
    Transformf d_trans = prev.inverse() * curr;
    m.transform_tobe_mapped = m.transform_tobe_mapped * e.transform_lb * d_trans * e.transform_lb.inverse();
    m.transform_sum = toT(Tsum);
  } else {
    m.Process(corner, surf, toT(Tsum));
  }
  const Transformf T_to_init = m.transform_aft_mapped;
  if (T_out) fromT(T_to_init, T_out);
  SolveReport R;
  const bool was_inited = e.inited;
  bool ok = was_inited ? e.ProcessLaserOdom(T_to_init, surf, corner, stamp, &R)
                       : e.ProcessLaserOdom(T_to_init, m.surf_stack_ds, m.corner_stack_ds, stamp, &R);
  if (!ok) return LIO_ERR_STATE;
  if (!was_inited && e.inited) m.imu_inited = true;  // SetInitFlag(true) (:545)
  fillReport(R, rep);
  return LIO_OK;
}
int lio_est_get_stage(const lio_est *h, int *stage, int *cir_buf_count, int *extrinsic_stage, int *last_event, double *R_WI, double *g_vec) {
  if (!h) return LIO_ERR_ARG;
  const Estimator &e = h->est;
  if (stage) *stage = e.inited ? 1 : 0;
  if (cir_buf_count) *cir_buf_count = e.cir_buf_count;
  if (extrinsic_stage) *extrinsic_stage = e.extrinsic_stage;
  if (last_event) *last_event = e.last_event;
  if (R_WI) for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R_WI[r * 3 + c] = e.R_WI(r, c);
  if (g_vec) for (int d = 0; d < 3; ++d) g_vec[d] = e.g_vec[d];
  return LIO_OK;
}
int lio_est_push_frame(lio_est *h, const lio_transform_f *T, const float *surf, size_t ns, const float *corner, size_t nc, double stamp) {
  if (!h || !T || (!surf && ns) || (!corner && nc)) return LIO_ERR_ARG;
  if (!h->est.inited) return LIO_ERR_STATE;
  return h->est.PushFrame(toT(*T), toCloud(surf, ns), toCloud(corner, nc), stamp) ? LIO_OK : LIO_ERR_STATE;
}
int lio_est_solve_optimization(lio_est *h, lio_solve_report *rep) {
  if (!h) return LIO_ERR_ARG;
  SolveReport R;
  if (!h->est.inited || !h->est.SolveOptimization(&R)) return LIO_ERR_STATE;
  fillReport(R, rep);
  return LIO_OK;
}
int lio_est_slide_window(lio_est *h) {
  if (!h) return LIO_ERR_ARG;
  if (!h->est.inited) return LIO_ERR_STATE;
  h->est.SlideWindow();
  return LIO_OK;
}
int lio_est_sync(lio_est *h) { return h ? LIO_OK : LIO_ERR_ARG; }
int lio_est_set_window(lio_est *h, int n, const double *Ps, const double *Rs, const double *Vs, const double *Bas, const double *Bgs,
                       const double g[3]) {
  if (!h || !Ps || !Rs || !Vs || !Bas || !Bgs || !g) return LIO_ERR_ARG;
  Estimator &e = h->est;
  if (n != e.W + 1) return LIO_ERR_ARG;
  for (int i = 0; i < n; ++i) {
    e.Ps[i] = v3(Ps + 3 * i); e.Vs[i] = v3(Vs + 3 * i); e.Bas[i] = v3(Bas + 3 * i); e.Bgs[i] = v3(Bgs + 3 * i);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) e.Rs[i](r, c) = Rs[9 * i + 3 * r + c];
  }
  e.g_vec = v3(g);
  e.inited = true; e.first_imu = true; e.cir_buf_count = e.W;
  e.n_state = e.n_frames = e.W + 1;
  return LIO_OK;
}
int lio_est_get_window(const lio_est *h, int n, double *Ps, double *Rs, double *Vs, double *Bas, double *Bgs, lio_transform_f *Tlb) {
  if (!h) return LIO_ERR_ARG;
  const Estimator &e = h->est;
  if (n != e.W + 1) return LIO_ERR_ARG;
  for (int i = 0; i < n; ++i) {
    for (int k = 0; k < 3; ++k) {
      if (Ps) Ps[3 * i + k] = e.Ps[i][k];
      if (Vs) Vs[3 * i + k] = e.Vs[i][k];
      if (Bas) Bas[3 * i + k] = e.Bas[i][k];
      if (Bgs) Bgs[3 * i + k] = e.Bgs[i][k];
    }
    if (Rs) for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rs[9 * i + 3 * r + c] = e.Rs[i](r, c);
  }
  if (Tlb) fromT(e.transform_lb, Tlb);
  return LIO_OK;
}
int lio_est_set_surf_stack(lio_est *h, int frame, const float *xyzi, size_t n) {
  if (!h || frame < 0 || frame > h->est.W || (!xyzi && n)) return LIO_ERR_ARG;
  h->est.surf_stack[frame] = toCloud(xyzi, n);
  h->est.size_surf_stack[frame] = n;
  return LIO_OK;
}
size_t lio_est_get_surf_stack(const lio_est *h, int frame, float *out) {
  if (!h || frame < 0 || frame > h->est.W) return 0;
  const Cloud &c = h->est.surf_stack[frame];
  if (out && !c.empty()) std::memcpy(out, c.data(), c.size() * sizeof(P4));
  return c.size();
}
int lio_est_set_preintegration(lio_est *h, int frame, const double acc0[3], const double gyr0[3], const double ba[3], const double bg[3],
                               const double *dt, const double *acc, const double *gyr, size_t ns) {
  if (!h || frame < 0 || frame > h->est.W || !acc0 || !gyr0 || !ba || !bg || (ns && (!dt || !acc || !gyr))) return LIO_ERR_ARG;
  auto p = std::make_shared<IntegrationBase>(v3(acc0), v3(gyr0), v3(ba), v3(bg), h->est.cfg.pim);
  for (size_t k = 0; k < ns; ++k) p->push_back(dt[k], v3(acc + 3 * k), v3(gyr + 3 * k));
  h->est.pre_integrations[frame] = p;
  return LIO_OK;
}
int lio_est_begin_frame(lio_est *h, const double acc[3], const double gyr[3]) {
  if (!h || !acc || !gyr) return LIO_ERR_ARG;
  h->est.BeginFrame(v3(acc), v3(gyr));
  return LIO_OK;
}
int lio_est_build_local_map(lio_est *h) {
  if (!h) return LIO_ERR_ARG;
  if (!h->est.inited) return LIO_ERR_STATE;
  h->est.BuildLocalMap(nullptr);
  return LIO_OK;
}
size_t lio_est_get_local_map(const lio_est *h, float *out) {
  if (!h) return 0;
  const Cloud &c = h->est.local_map_filtered;
  if (out && !c.empty()) std::memcpy(out, c.data(), c.size() * sizeof(P4));
  return c.size();
}
size_t lio_est_get_features(const lio_est *h, int frame, double *pt, double *co, double *sc) {
  if (!h || frame < 0 || frame >= int(h->est.feature_frames.size())) return 0;
  const auto &f = h->est.feature_frames[frame];
  for (size_t k = 0; k < f.size(); ++k) {
    if (pt) { pt[3 * k] = f[k].point.x; pt[3 * k + 1] = f[k].point.y; pt[3 * k + 2] = f[k].point.z; }
    if (co) for (int j = 0; j < 4; ++j) co[4 * k + j] = f[k].coeffs[j];
    if (sc) sc[k] = f[k].score;
  }
  return f.size();
}
int lio_est_get_laser_odom_transform(const lio_est *h, lio_transform_f *out) {
  if (!h || !out) return LIO_ERR_ARG;
  fromT(h->est.laser_odom_transform, out);
  return LIO_OK;
}
int lio_est_get_prior(const lio_est *h, double *JtJ, double *Jtr, double *x0, int *x0_len) {
  if (!h) return LIO_ERR_ARG;
  const auto &pr = h->est.last_marg;
  if (!pr) return 0;
  int n = pr->n;
  if (JtJ) for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += pr->lin_jac(k, i) * pr->lin_jac(k, j); JtJ[i * n + j] = s; }
  if (Jtr) for (int i = 0; i < n; ++i) { double s = 0; for (int k = 0; k < n; ++k) s += pr->lin_jac(k, i) * pr->lin_res[k]; Jtr[i] = s; }
  int len = 0;
  for (const auto &b : pr->x0) { if (x0) for (double v : b) x0[len++] = v; else len += int(b.size()); }
  if (x0_len) *x0_len = len;
  return n;
}
int lio_est_get_prior_factor(const lio_est *h, double *lin_jac, double *lin_res, double *x0, int *x0_len) {
  if (!h) return LIO_ERR_ARG;
  const auto &pr = h->est.last_marg;
  if (!pr) return 0;
  int n = pr->n;
  if (lin_jac) for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) lin_jac[i * n + j] = pr->lin_jac(i, j);
  if (lin_res) for (int i = 0; i < n; ++i) lin_res[i] = pr->lin_res[i];
  int len = 0;
  for (const auto &b : pr->x0) { if (x0) for (double v : b) x0[len++] = v; else len += int(b.size()); }
  if (x0_len) *x0_len = len;
  return n;
}
int lio_est_set_prior_factor(lio_est *h, int n, const double *lin_jac, const double *lin_res, const double *x0, int x0_len) {
  if (!h || !lin_jac || !lin_res || !x0 || n <= 0) return LIO_ERR_ARG;
  const auto &old = h->est.last_marg;
  if (!old || old->n != n) return LIO_ERR_STATE;
  int len = 0;
  for (const auto &b : old->x0) len += int(b.size());
  if (len != x0_len) return LIO_ERR_STATE;
  auto pr = std::make_shared<MargPrior>(*old);
  for (int i = 0; i < n; ++i) { for (int j = 0; j < n; ++j) pr->lin_jac(i, j) = lin_jac[i * n + j]; pr->lin_res[i] = lin_res[i]; }
  len = 0;
  for (auto &b : pr->x0) for (double &v : b) v = x0[len++];
  h->est.last_marg = pr;
  return LIO_OK;
}
int lio_est_set_extrinsic(lio_est *h, const lio_transform_f *T) {
  if (!h || !T) return LIO_ERR_ARG;
  h->est.transform_lb = toT(*T);
  return LIO_OK;
}
int lio_dense_spd_solve(const double *A, const double *b, int n, double *x) {
  if (!A || !b || !x || n < 1 || n > 128) return LIO_ERR_ARG;
  // plain Cholesky A = L L^T (the oracle's dense solve)
  std::vector<double> Lm(A, A + size_t(n) * n), y(b, b + n);
  for (int j = 0; j < n; ++j) {
    double d = Lm[size_t(j) * n + j];
    for (int k = 0; k < j; ++k) d -= Lm[size_t(j) * n + k] * Lm[size_t(j) * n + k];
    if (!(d > 0)) return LIO_ERR_STATE;
    const double l = std::sqrt(d);
    Lm[size_t(j) * n + j] = l;
    for (int i = j + 1; i < n; ++i) {
      double sres = Lm[size_t(i) * n + j];
      for (int k = 0; k < j; ++k) sres -= Lm[size_t(i) * n + k] * Lm[size_t(j) * n + k];
      Lm[size_t(i) * n + j] = sres / l;
    }
  }
  for (int i = 0; i < n; ++i) { double sres = y[i]; for (int k = 0; k < i; ++k) sres -= Lm[size_t(i) * n + k] * y[k]; y[i] = sres / Lm[size_t(i) * n + i]; }
  for (int i = n - 1; i >= 0; --i) { double sres = y[i]; for (int k = i + 1; k < n; ++k) sres -= Lm[size_t(k) * n + i] * y[k]; y[i] = sres / Lm[size_t(i) * n + i]; }
  for (int i = 0; i < n; ++i) x[i] = y[i];
  return LIO_OK;
}
int lio_est_snapshot(lio_est *h) {
  if (!h) return LIO_ERR_ARG;
  h->snap.reset(new Estimator(h->est));
  // deep-copy the mutable pre-integration in flight (the window ones are immutable once pushed)
  if (h->est.tmp_pre_integration) h->snap->tmp_pre_integration = std::make_shared<IntegrationBase>(*h->est.tmp_pre_integration);
  return LIO_OK;
}
int lio_est_restore(lio_est *h) {
  if (!h) return LIO_ERR_ARG;
  if (!h->snap) return LIO_ERR_STATE;
  h->est = *h->snap;
  if (h->snap->tmp_pre_integration) h->est.tmp_pre_integration = std::make_shared<IntegrationBase>(*h->snap->tmp_pre_integration);
  return LIO_OK;
}

int lio_est_copy_snapshot(lio_est *dst, lio_est *src) {
  if (!dst || !src) return LIO_ERR_ARG;
  if (!src->snap) return LIO_ERR_STATE;
  dst->snap.reset(new Estimator(*src->snap));
  if (src->snap->tmp_pre_integration) dst->snap->tmp_pre_integration = std::make_shared<IntegrationBase>(*src->snap->tmp_pre_integration);
  return LIO_OK;
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

// lio_est_batch (include/lio_c.h): the oracle has one way to solve a window; a batch is a loop over its members
struct lio_est_batch { std::vector<lio_est *> members; bool dissolved = false; };
lio_est_batch *lio_est_batch_create(lio_est *const *windows, int n) {
  if (!windows || n < 1 || n > 65535) return nullptr;
  for (int i = 0; i < n; ++i) {
    if (!windows[i] || windows[i]->adopted) return nullptr;
    for (int j = 0; j < i; ++j) if (windows[j] == windows[i]) return nullptr;
  }
  lio_est_batch *b = new lio_est_batch;
  b->members.assign(windows, windows + n);
  for (lio_est *m : b->members) { m->adopted = true; m->owner = b; }
  return b;
}
static void dissolve_batch(lio_est_batch *b) {
  for (lio_est *m : b->members) { m->adopted = false; m->owner = nullptr; }
  b->members.clear();
  b->dissolved = true;
}
void lio_est_batch_destroy(lio_est_batch *b) {
  if (!b) return;
  dissolve_batch(b);
  delete b;
}
int lio_est_batch_size(const lio_est_batch *b) { return b ? int(b->members.size()) : 0; }
int lio_est_batch_solve(lio_est_batch *b, lio_solve_report *reps) {
  if (!b) return LIO_ERR_ARG;
  if (b->dissolved) return LIO_ERR_STATE;
  for (lio_est *m : b->members) if (!m->est.inited) return LIO_ERR_STATE;   // (checked up front, as the product does: nothing is solved then)
  for (size_t w = 0; w < b->members.size(); ++w) {
    const int rc = lio_est_solve_optimization(b->members[w], reps ? reps + w : nullptr);
    if (rc != LIO_OK) return rc;
  }
  return LIO_OK;
}
int lio_est_batch_solve_restored(lio_est_batch *b, int steps, lio_solve_report *reps) {
  if (!b || steps < 0) return LIO_ERR_ARG;
  if (b->dissolved) return LIO_ERR_STATE;
  for (int k = 0; k < steps; ++k) {
    for (lio_est *m : b->members) { const int rc = lio_est_restore(m); if (rc != LIO_OK) return rc; }
    const int rc = lio_est_batch_solve(b, reps);
    if (rc != LIO_OK) return rc;
  }
  return LIO_OK;
}
int lio_est_batch_sync(lio_est_batch *b) { return b ? (b->dissolved ? LIO_ERR_STATE : LIO_OK) : LIO_ERR_ARG; }
int lio_est_batch_set_option(lio_est_batch *b, const char *name, int value) {   // execution choices of the product: accepted, ignored
  if (!b || !name) return LIO_ERR_ARG;
  if (b->dissolved) return LIO_ERR_STATE;
  (void)value;
  for (const char *k : {"lanes_per_query", "occupancy", "loop_groups", "aux_threads", "aux_stream", "finish_threads", "time_kernels", "parts"})
    if (std::strcmp(name, k) == 0) return LIO_OK;
  return LIO_ERR_ARG;
}
int lio_seg_sort_pairs(const unsigned *keys, const unsigned *vals, size_t n_total, const int *seg_off, const int *seg_n, int nseg, int bits, int passes, unsigned *keys_out,
                       unsigned *vals_out) {
  // the CPU statement of what the hook sorts: a stable sort of every segment on the low passes * bits bits of the key
  if (!keys || !seg_off || !seg_n || nseg < 1 || bits < 1 || bits > 9 || passes < 1 || passes * bits > 32 || !keys_out || !vals_out) return LIO_ERR_ARG;
  for (size_t i = 0; i < n_total; ++i) { keys_out[i] = keys[i]; vals_out[i] = vals ? vals[i] : 0u; }
  const unsigned long long mask = (passes * bits >= 32) ? 0xFFFFFFFFull : ((1ull << (passes * bits)) - 1ull);
  for (int k = 0; k < nseg; ++k) {
    if (seg_off[k] < 0 || seg_n[k] < 0 || size_t(seg_off[k]) + size_t(seg_n[k]) > n_total) return LIO_ERR_ARG;
    std::vector<unsigned> idx(size_t(seg_n[k]));
    for (int i = 0; i < seg_n[k]; ++i) idx[size_t(i)] = unsigned(seg_off[k] + i);
    std::stable_sort(idx.begin(), idx.end(), [&](unsigned a, unsigned b) { return (keys[a] & mask) < (keys[b] & mask); });
    for (int i = 0; i < seg_n[k]; ++i) { keys_out[seg_off[k] + i] = keys[idx[size_t(i)]]; vals_out[seg_off[k] + i] = vals ? vals[idx[size_t(i)]] : idx[size_t(i)]; }
  }
  return LIO_OK;
}
int lio_est_batch_stage_digest(lio_est_batch *b, int stage, unsigned long long *out) {
  if (!b || !out || stage < 0 || stage > 9) return LIO_ERR_ARG;
  if (b->dissolved) return LIO_ERR_STATE;
  for (size_t w = 0; w < b->members.size(); ++w) out[w] = 0;
  return LIO_OK;
}
int lio_est_batch_get_clock(const lio_est_batch *b, double *out) {
  if (!b || !out) return LIO_ERR_ARG;
  if (b->dissolved) return LIO_ERR_STATE;
  for (int k = 0; k < 24; ++k) out[k] = 0.0;
  return LIO_OK;
}


// oracle-only probes (not part of lio_c.h): include/utils/math_utils.h:44-64, pinned by the reference's own
// assertions at test/test_point_processor/test_point_processor.cc:57-61
// ---- oracle-only hooks for tests/golden (second-sourcing the restated third-party semantics)
int orc_colpiv_qr_solve_f32(int m, int n, const float *A, const float *b, float *x) {
  if (m < 1 || n < 1 || m > 16 || n > 16 || !A || !b || !x) return -1;
  std::vector<float> Ac(A, A + size_t(m) * n), bc(b, b + m);
  colpiv_qr_solve<float>(m, n, Ac.data(), bc.data(), x);
  return 0;
}
int lio_marginalize_schur(const double *A, const double *b, int m, int n, double *lin_jac, double *lin_res, double *evals) {
  if (!A || !b || !lin_jac || !lin_res || m < 1 || n < 1) return LIO_ERR_ARG;
  const int pos = m + n;
  Mat Am(pos, pos);
  for (int i = 0; i < pos * pos; ++i) Am.a[i] = A[i];
  std::vector<double> bv(b, b + pos), lr;
  Mat lj;
  MarginalizeSchur(Am, bv, m, n, lj, lr);
  for (int i = 0; i < n * n; ++i) lin_jac[i] = lj.a[i];
  for (int i = 0; i < n; ++i) lin_res[i] = lr[i];
  if (evals)   // row k of lin_jac = sqrt(s_k) v_k^T with |v_k| = 1
    for (int k = 0; k < n; ++k) { double s2 = 0; for (int i = 0; i < n; ++i) s2 += lj.a[size_t(k) * n + i] * lj.a[size_t(k) * n + i]; evals[k] = s2; }
  return LIO_OK;
}
int orc_marginalize_schur(const double *A, const double *b, int m, int n, double *lin_jac, double *lin_res) {
  if (!A || !b || m < 1 || n < 1) return -1;
  const int pos = m + n;
  Mat Am(pos, pos);
  for (int i = 0; i < pos * pos; ++i) Am.a[i] = A[i];
  std::vector<double> bv(b, b + pos), lr;
  Mat lj;
  MarginalizeSchur(Am, bv, m, n, lj, lr);
  for (int i = 0; i < n * n; ++i) lin_jac[i] = lj.a[i];
  for (int i = 0; i < n; ++i) lin_res[i] = lr[i];
  return 0;
}
static DoglegDump g_dump;
int orc_est_solve_with_dump(lio_est *h, lio_solve_report *rep) {
  g_dump = DoglegDump();
  dogleg_dump_sink() = &g_dump;
  const int rc = lio_est_solve_optimization(h, rep);
  dogleg_dump_sink() = nullptr;
  return rc;
}
int orc_dump_sizes(int *n, int *n_lin, int *n_it) { *n = g_dump.n; *n_lin = int(g_dump.H.size()); *n_it = int(g_dump.its.size()); return 0; }
int orc_dump_lin(int k, double *H, double *g, double *cost) {
  if (k < 0 || k >= int(g_dump.H.size())) return -1;
  std::memcpy(H, g_dump.H[k].data(), sizeof(double) * g_dump.H[k].size());
  std::memcpy(g, g_dump.g[k].data(), sizeof(double) * g_dump.g[k].size());
  *cost = g_dump.cost[k];
  return 0;
}
// scalars: radius, mu, cand_cost, model_change, step_norm, x_norm, gmax; flags: lin, valid, accepted
int orc_dump_it(int k, double *scalars, int *flags, double *delta) {
  if (k < 0 || k >= int(g_dump.its.size())) return -1;
  const DoglegDump::It &it = g_dump.its[k];
  scalars[0] = it.radius; scalars[1] = it.mu; scalars[2] = it.cand_cost; scalars[3] = it.model_change; scalars[4] = it.step_norm;
  scalars[5] = it.x_norm; scalars[6] = it.gmax;
  flags[0] = it.lin; flags[1] = it.valid; flags[2] = it.accepted;
  std::memcpy(delta, it.delta.data(), sizeof(double) * it.delta.size());
  return 0;
}
double orc_normalize_rad(double r) { return NormalizeRad(r); }
double orc_normalize_deg(double d) { return NormalizeDeg(d); }

int lio_est_set_factor_sharding(lio_est *h, int rank, int world, lio_allreduce_fn fn, void *user) {
  if (!h || world < 1 || rank < 0 || rank >= world) return LIO_ERR_ARG;
  h->est.shard_rank = rank; h->est.shard_world = world; h->est.allreduce = fn; h->est.allreduce_user = user;
  return LIO_OK;
}
// RCCL lives in the product only (the oracle is a CPU library)
int lio_rccl_unique_id(unsigned char *) { return LIO_ERR_DEVICE; }
lio_rccl *lio_rccl_init(const unsigned char *, int, int) { return nullptr; }
void lio_rccl_destroy(lio_rccl *) {}
int lio_rccl_rank(const lio_rccl *) { return -1; }
int lio_rccl_world(const lio_rccl *) { return 0; }
int lio_est_set_factor_sharding_rccl(lio_est *, lio_rccl *) { return LIO_ERR_DEVICE; }
int lio_bench_voxel_grid(const float *, size_t, float, int, double *, size_t *) { return LIO_ERR_DEVICE; }
int lio_rccl_bench_all_reduce(lio_rccl *, int, int, double *) { return LIO_ERR_DEVICE; }
int lio_kf_batch_refine_gather(lio_kf_batch *, lio_rccl *, int, float *, double *) { return LIO_ERR_DEVICE; }
int lio_est_bench_batched_moments(lio_est *, int, int, double *, double *) { return LIO_ERR_STATE; }  // device-only measurement
int lio_est_enable_kernel_timing(lio_est *h, int) { return h ? LIO_OK : LIO_ERR_ARG; }
int lio_est_get_kernel_timing(lio_est *, const char *, double *t, double *b) { if (t) *t = 0; if (b) *b = 0; return 0; }

}  // extern "C"
