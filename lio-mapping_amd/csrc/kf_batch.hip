// kf_batch.hip — host side of the batched keyframe refinement; the kernels are the scan-to-map stages of cloud_kernels.hip
// indexed by keyframe.
#include "kf_batch.h"
#include "rccl_comm.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace lio {

static const int kChunk = 32768;  // keyframes per launch (grid y/z limit 65535)

KfBatchDev::KfBatchDev(const lio_map_config &cfg) : cfg_(cfg) {
  LIO_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  LIO_HIP(hipEventCreate(&ev0_));
  LIO_HIP(hipEventCreate(&ev1_));
  LIO_HIP(hipHostMalloc(reinterpret_cast<void **>(&h_nconv_), sizeof(int), hipHostMallocDefault));
  d_nconv_.reserve(1);
}

KfBatchDev::~KfBatchDev() {
  if (stream_) (void)hipStreamSynchronize(stream_);
  if (h_nconv_) (void)hipHostFree(h_nconv_);
  if (ev0_) (void)hipEventDestroy(ev0_);
  if (ev1_) (void)hipEventDestroy(ev1_);
  if (stream_) (void)hipStreamDestroy(stream_);
}

static void host_bounds(const float *xyzi, size_t n, float mn[3], float mx[3]) {
  for (int d = 0; d < 3; ++d) { mn[d] = 0.f; mx[d] = 0.f; }
  bool first = true;
  for (size_t i = 0; i < n; ++i) {
    const float *p = xyzi + 4 * i;
    if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
    for (int d = 0; d < 3; ++d) {
      if (first || p[d] < mn[d]) mn[d] = p[d];
      if (first || p[d] > mx[d]) mx[d] = p[d];
    }
    first = false;
  }
}

int KfBatchDev::AddMap(const float *corner, size_t nc, const float *surf, size_t ns) {
  std::unique_ptr<Map> m(new Map);
  m->nc = nc; m->ns = ns;
  const float cell = std::sqrt(cfg_.min_match_sq_dis) * 1.0001f;
  const float *src[2] = {corner, surf};
  const size_t cnt[2] = {nc, ns};
  DBuf<float4> *dst[2] = {&m->corner, &m->surf};
  KnnGrid *grid[2] = {&m->gc, &m->gs};
  for (int c = 0; c < 2; ++c) {
    dst[c]->reserve(std::max<size_t>(cnt[c], 1));
    float mn[3], mx[3];
    host_bounds(src[c], cnt[c], mn, mx);
    if (cnt[c]) LIO_HIP(hipMemcpyAsync(dst[c]->p, src[c], cnt[c] * sizeof(float4), hipMemcpyHostToDevice, stream_));
    grid[c]->build(dst[c]->p, cnt[c], mn, mx, cell, stream_);
  }
  LIO_HIP(hipStreamSynchronize(stream_));  // the caller's buffers are not retained
  KfMapDesc d{};
  d.corner_sorted = m->gc.sorted(); d.corner_cells = m->gc.cells(); d.corner_grid = m->gc.desc();
  d.surf_sorted = m->gs.sorted(); d.surf_cells = m->gs.cells(); d.surf_grid = m->gs.desc();
  h_md_.push_back(d);
  maps_.push_back(std::move(m));
  md_dirty_ = true;
  return int(maps_.size()) - 1;
}

int KfBatchDev::AddKeyframe(int map, const float *corner, size_t nc, const float *surf, size_t ns, const Rigid<float> &T) {
  if (map < 0 || size_t(map) >= maps_.size()) return -1;
  KfDesc d{};
  d.slot_off = int(h_stack_.size());
  d.Mc = int(nc); d.Ms = int(ns); d.map = map;
  d.nb = odom_rows_blocks(int(nc + ns));
  d.part_off = total_nb_;
  total_nb_ += d.nb;
  const Vec3<float> z = rotate(T.rot, Vec3<float>(0.f, 0.f, 10.f));  // point_on_z_axis_ (PointMapping.cc:803-806)
  d.pz[0] = z.x + T.pos.x; d.pz[1] = z.y + T.pos.y; d.pz[2] = z.z + T.pos.z;
  const size_t base = h_stack_.size();
  h_stack_.resize(base + nc + ns);
  if (nc) std::memcpy(&h_stack_[base], corner, nc * sizeof(float4));
  if (ns) std::memcpy(&h_stack_[base + nc], surf, ns * sizeof(float4));
  OdomState st;
  std::memset(&st, 0, sizeof(st));
  st.T[0] = T.rot.x; st.T[1] = T.rot.y; st.T[2] = T.rot.z; st.T[3] = T.rot.w; st.T[4] = T.pos.x; st.T[5] = T.pos.y; st.T[6] = T.pos.z;
  const Map &m = *maps_[size_t(map)];
  if (m.nc <= 10 || m.ns <= 100) { st.converged = 1; ++n_gated_; }  // the early return of :327-329: pose untouched, 0 iterations
  h_st0_.push_back(st);
  h_kd_.push_back(d);
  max_Mc_ = std::max(max_Mc_, d.Mc); max_Ms_ = std::max(max_Ms_, d.Ms); max_nb_ = std::max(max_nb_, d.nb);
  n_queries_ += (long long)(nc + ns);
  kf_dirty_ = true; order_valid_ = false;
  return int(h_kd_.size()) - 1;
}

void KfBatchDev::ClearKeyframes() {
  h_stack_.clear(); h_kd_.clear(); h_st0_.clear(); h_st_.clear();
  max_Mc_ = max_Ms_ = 0; max_nb_ = 1; total_nb_ = 0; n_gated_ = 0; n_queries_ = 0;
  kf_dirty_ = true; order_valid_ = false;
}

// The order the round kernels take the queries in: every keyframe's corner and surf queries sorted by the map cell they fall into under the
// keyframe's starting pose (they move by centimetres over the rounds).  A wave's 64 queries then walk the same few rows of cells, and the
// 16-byte gathers of its lanes fall into shared cache lines instead of 64 different ones (the walk is bound by those gathers: with every
// lane of a wave on ONE query the batched features kernel ran 2.1x faster, profiles/r6_d_query_order.txt).  Results stay in the queries'
// own slots: rows, updates and poses do not see the order.
void KfBatchDev::BuildQueryOrder(hipStream_t s) {
  const int B = int(h_kd_.size());
  const size_t n = std::max<size_t>(h_stack_.size(), 1);
  order_.reserve(n); qkeys_.reserve(n); qkeys2_.reserve(n); qvals2_.reserve(n);
  std::vector<SegDesc> seg(size_t(2) * B);
  for (int k = 0; k < B; ++k) {
    seg[size_t(2) * k] = SegDesc{h_kd_[size_t(k)].slot_off, h_kd_[size_t(k)].Mc, 0};
    seg[size_t(2) * k + 1] = SegDesc{h_kd_[size_t(k)].slot_off + h_kd_[size_t(k)].Mc, h_kd_[size_t(k)].Ms, 0};
  }
  int bits = 1;
  for (const KfMapDesc &m : h_md_)
    for (const GridDesc *g : {&m.corner_grid, &m.surf_grid}) {
      const long long nc = (long long)g->dims[0] * g->dims[1] * g->dims[2];
      while ((1ll << bits) < nc) ++bits;
    }
  const int passes = std::max(1, (bits + SS_MAX_BITS - 1) / SS_MAX_BITS);
  const SegSortPlan plan = seg_sort_plan(seg.data(), 2 * B, SS_MAX_BITS);
  qhist_.reserve(std::max<size_t>(plan.hist_entries, 1));
  d_qseg_.reserve(seg.size());
  LIO_HIP(hipMemcpyAsync(d_qseg_.p, seg.data(), seg.size() * sizeof(SegDesc), hipMemcpyHostToDevice, s));
  for (int off = 0; off < B; off += kChunk)
    launch_kf_query_keys(d_kd_.p + off, d_md_.p, d_st_.p + off, std::min(kChunk, B - off), max_Mc_, max_Ms_, d_stack_.p, qkeys_.p, s);
  // ping-pong so that the last pass lands in order_ (queries outside the grid sort last: their keys keep the bits above the passes)
  const uint32_t *ki = qkeys_.p, *vi = nullptr;
  uint32_t *kb[2] = {qkeys2_.p, qkeys_.p}, *vb[2] = {(passes & 1) ? order_.p : qvals2_.p, (passes & 1) ? qvals2_.p : order_.p};
  for (int p = 0; p < passes; ++p) {
    seg_sort_pass(d_qseg_.p, 2 * B, plan, ki, vi, kb[p & 1], vb[p & 1], qhist_.p, p * SS_MAX_BITS, SS_MAX_BITS, nullptr, s);
    ki = kb[p & 1]; vi = vb[p & 1];
  }
  LIO_HIP(hipStreamSynchronize(s));   // (seg is a local)
  order_valid_ = true;
}

void KfBatchDev::Refine() {
  hipStream_t s = stream_;
  const int B = int(h_kd_.size());
  h_st_ = h_st0_;
  rounds_ = 0; device_ms_ = 0;
  if (B == 0) return;
  if (md_dirty_) {
    d_md_.reserve(h_md_.size());
    LIO_HIP(hipMemcpyAsync(d_md_.p, h_md_.data(), h_md_.size() * sizeof(KfMapDesc), hipMemcpyHostToDevice, s));
    md_dirty_ = false;
  }
  if (kf_dirty_) {
    const size_t n = std::max<size_t>(h_stack_.size(), 1);
    d_stack_.reserve(n); coef_.reserve(n); valid_.reserve(n);
    d_kd_.reserve(size_t(B)); d_st_.reserve(size_t(B)); partials_.reserve(size_t(std::max(total_nb_, 1)) * 28);
    if (!h_stack_.empty()) LIO_HIP(hipMemcpyAsync(d_stack_.p, h_stack_.data(), h_stack_.size() * sizeof(float4), hipMemcpyHostToDevice, s));
    LIO_HIP(hipMemcpyAsync(d_kd_.p, h_kd_.data(), size_t(B) * sizeof(KfDesc), hipMemcpyHostToDevice, s));
    kf_dirty_ = false;
  }
  LIO_HIP(hipMemcpyAsync(d_st_.p, h_st0_.data(), size_t(B) * sizeof(OdomState), hipMemcpyHostToDevice, s));
  LIO_HIP(hipMemsetAsync(d_nconv_.p, 0, sizeof(int), s));
  if (!order_valid_) BuildQueryOrder(s);
  const bool four_dof = cfg_.map_builder && cfg_.enable_4d;
  const int mode = four_dof ? 2 : 1;
  const int max_it = cfg_.num_max_iterations;
  LIO_HIP(hipEventRecord(ev0_, s));
  for (int iter = 0; iter < max_it; ++iter) {
    for (int off = 0; off < B; off += kChunk) {
      const int nk = std::min(kChunk, B - off);
      launch_kf_round(d_kd_.p + off, d_md_.p, d_st_.p + off, nk, max_Mc_, max_Ms_, n_queries_, d_stack_.p, order_.p, cfg_.min_match_sq_dis, cfg_.min_plane_dis, mode, valid_.p,
                      coef_.p, s);
      launch_kf_rows(d_kd_.p + off, d_st_.p + off, nk, max_nb_, d_stack_.p, valid_.p, coef_.p, partials_.p, mode, s);
      launch_kf_update(d_kd_.p + off, d_st_.p + off, nk, partials_.p, iter, 50, four_dof ? 1 : 0, d_nconv_.p, s);
    }
    ++rounds_;
    // most keyframes converge in 5-7 rounds; converged keyframes cost nothing in later rounds, so peek sparsely
    if (iter + 1 >= 5 && (iter + 1) % 2 == 1 && iter + 1 < max_it) {
      LIO_HIP(hipMemcpyAsync(h_nconv_, d_nconv_.p, sizeof(int), hipMemcpyDeviceToHost, s));
      LIO_HIP(hipStreamSynchronize(s));
      if (*h_nconv_ + n_gated_ >= B) break;
    }
  }
  LIO_HIP(hipEventRecord(ev1_, s));
  LIO_HIP(hipMemcpyAsync(h_st_.data(), d_st_.p, size_t(B) * sizeof(OdomState), hipMemcpyDeviceToHost, s));
  LIO_HIP(hipStreamSynchronize(s));
  float ms = 0;
  LIO_HIP(hipEventElapsedTime(&ms, ev0_, ev1_));
  device_ms_ = ms;
}

__global__ void k_kf_pack(const OdomState *__restrict__ st, int B, int slots, float *__restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= slots) return;
  float *o = out + size_t(k) * 9;
  if (k < B) {
    const OdomState s = st[k];
#pragma unroll
    for (int j = 0; j < 7; ++j) o[j] = s.T[j];
    o[7] = float(s.iters); o[8] = float(s.nsel);
  } else {
#pragma unroll
    for (int j = 0; j < 9; ++j) o[j] = 0.f;
  }
}

void KfBatchDev::RefineGather(void *nccl_comm, int world, int slots_per_rank, float *packed_all) {
  if (slots_per_rank < int(h_kd_.size())) throw std::runtime_error("RefineGather: slots_per_rank smaller than this rank's keyframe count");
  // A rank whose own refinement fails still takes part in the gather (with zeroed records) and reports the failure afterwards:
  // returning before the collective would leave every peer blocked in ncclAllGather.
  std::exception_ptr local_failure;
  int B = int(h_kd_.size());
  try { Refine(); } catch (...) { local_failure = std::current_exception(); B = 0; }
  hipStream_t s = stream_;
  d_pack_.reserve(size_t(slots_per_rank) * 9);
  d_gather_.reserve(size_t(world) * slots_per_rank * 9);
  d_st_.reserve(std::max(1, B));
  hipLaunchKernelGGL(k_kf_pack, dim3(cdiv(slots_per_rank, 256)), dim3(256), 0, s, d_st_.p, B, slots_per_rank, d_pack_.p);
  LIO_HIP(hipGetLastError());
  rccl_all_gather_f32(nccl_comm, d_pack_.p, d_gather_.p, size_t(slots_per_rank) * 9, s);
  LIO_HIP(hipMemcpyAsync(packed_all, d_gather_.p, sizeof(float) * size_t(world) * slots_per_rank * 9, hipMemcpyDeviceToHost, s));
  LIO_HIP(hipStreamSynchronize(s));
  if (local_failure) std::rethrow_exception(local_failure);
}

}  // namespace lio
