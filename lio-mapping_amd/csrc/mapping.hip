// mapping.hip — PointMapping on the GPU (see mapping.h for the layout).  No CPU path: every stage below is a HIP
// kernel or a rocPRIM primitive; the host only does the 6-DoF bookkeeping (transform algebra, cube-window shift,
// FOV test of <= 125 cubes).
#include <cstring>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstddef>

#include "mapping.h"

namespace lio {

namespace {

inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__host__ __device__ inline uint32_t pack_cube(int ai, int aj, int ak) {
  return (uint32_t(ai + 512) << 20) | (uint32_t(aj + 512) << 10) | uint32_t(ak + 512);
}
__host__ __device__ inline void unpack_cube(uint32_t k, int &ai, int &aj, int &ak) {
  ai = int((k >> 20) & 1023u) - 512; aj = int((k >> 10) & 1023u) - 512; ak = int(k & 1023u) - 512;
}

// int((v + 25.0) / 50.0) + cen, minus one for negative arguments (PointMapping.cc:810-817,1126-1132)
__host__ __device__ inline int cube_coord(float v, int cen) {
  int r = int((double(v) + 25.0) / 50.0) + cen;
  if (double(v) + 25.0 < 0) --r;
  return r;
}

__device__ inline uint32_t rank_of_key(uint32_t key, const MapValidSet &vs) {
  int ai, aj, ak;
  unpack_cube(key, ai, aj, ak);
  if (ai < vs.lo[0] || ai >= vs.hi[0] || aj < vs.lo[1] || aj >= vs.hi[1] || ak < vs.lo[2] || ak >= vs.hi[2]) return LIO_MAP_RANK_DROP;
  int lo = 0, hi = vs.n;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (vs.key[mid] < key) lo = mid + 1; else hi = mid;
  }
  return (lo < vs.n && vs.key[lo] == key) ? uint32_t(lo) : LIO_MAP_RANK_REST;
}

__device__ inline void count_classes(uint32_t r, bool active, int *n_valid, int *n_rest) {
  const unsigned long long bv = __ballot(active && r < LIO_MAP_RANK_REST);
  const unsigned long long br = __ballot(active && r == LIO_MAP_RANK_REST);
  if ((threadIdx.x & 63) == 0) {
    if (bv) atomicAdd(n_valid, __popcll(bv));
    if (br) atomicAdd(n_rest, __popcll(br));
  }
}

__global__ void k_map_rank(const uint32_t *__restrict__ pkey, int n, MapValidSet vs, uint32_t *__restrict__ rk, uint32_t *__restrict__ vals,
                           MapCounters *cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t r = LIO_MAP_RANK_DROP;
  if (i < n) {
    r = rank_of_key(pkey[i], vs);
    rk[i] = r; vals[i] = uint32_t(i);
  }
  count_classes(r, i < n, &cnt->n_valid, &cnt->n_rest);
}

// sorted order = [valid cubes by rank | rest | dropped]; written out as [rest | valid]
__global__ void k_map_gather(const float4 *__restrict__ pool, const uint32_t *__restrict__ pkey, const uint32_t *__restrict__ rk_sorted,
                             const uint32_t *__restrict__ src_sorted, int n, const MapCounters *__restrict__ cnt, float4 *__restrict__ pool_out,
                             uint32_t *__restrict__ pkey_out, uint32_t *__restrict__ vrank) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const uint32_t r = rk_sorted[s];
  if (r == LIO_MAP_RANK_DROP) return;
  const int nv = cnt->n_valid, nr = cnt->n_rest;
  const uint32_t src = src_sorted[s];
  int dst;
  if (r < LIO_MAP_RANK_REST) { dst = nr + s; vrank[s] = r; }
  else dst = s - nv;
  pool_out[dst] = pool[src];
  pkey_out[dst] = pkey[src];
}

// PointAssociateToMap then the cube of the mapped point (PointMapping.cc:1122-1139)
__global__ void k_map_new(const float4 *__restrict__ sensor_pts, int n, Quat<float> q, Vec3<float> t, int c0, int c1, int c2, MapValidSet vs,
                          float4 *__restrict__ new_pts, uint32_t *__restrict__ new_key, uint32_t *__restrict__ rk, uint32_t *__restrict__ vals,
                          MapCounters *cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t r = LIO_MAP_RANK_DROP;
  if (i < n) {
    float4 p = sensor_pts[i];
    Vec3<float> v = rotate(q, Vec3<float>(p.x, p.y, p.z));
    p.x = v.x + t.x; p.y = v.y + t.y; p.z = v.z + t.z;
    const int ci = cube_coord(p.x, c0), cj = cube_coord(p.y, c1), ck = cube_coord(p.z, c2);
    uint32_t key = 0;
    if (ci >= 0 && ci < MappingDev::L && cj >= 0 && cj < MappingDev::Wd && ck >= 0 && ck < MappingDev::H) {
      key = pack_cube(ci - c0, cj - c1, ck - c2);
      r = rank_of_key(key, vs);
    }
    new_pts[i] = p; new_key[i] = key; rk[i] = r; vals[i] = uint32_t(i);
  }
  count_classes(r, i < n, &cnt->n_new_valid, &cnt->n_new_rest);
}

// new points, sorted [valid by rank | rest | dropped]: valid ones are appended to the voxel work list U behind the
// nV map points of those cubes, rest ones behind the untouched part of the pool
__global__ void k_new_scatter(const float4 *__restrict__ new_pts, const uint32_t *__restrict__ new_key, const uint32_t *__restrict__ rk_sorted,
                              const uint32_t *__restrict__ src_sorted, int n_new, const MapCounters *__restrict__ cnt, int nV, int n_rest,
                              float4 *__restrict__ u_pts, uint32_t *__restrict__ u_rank, float4 *__restrict__ pool, uint32_t *__restrict__ pkey) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_new) return;
  const int nnv = cnt->n_new_valid, nnr = cnt->n_new_rest;
  const uint32_t src = src_sorted[s];
  if (s < nnv) {
    u_pts[nV + s] = new_pts[src];
    u_rank[nV + s] = rk_sorted[s];
  } else {
    u_rank[nV + s] = 0xFFFFFFFFu;  // padding of the work list
    if (s < nnv + nnr) {
      pool[n_rest + (s - nnv)] = new_pts[src];
      pkey[n_rest + (s - nnv)] = new_key[src];
    }
  }
}

__device__ inline int f2ord(float f) { int b = __float_as_int(f); return b >= 0 ? b : b ^ 0x7FFFFFFF; }
__host__ __device__ inline float ord2f(int o) {
  int b = o >= 0 ? o : o ^ 0x7FFFFFFF;
  float f;
#if defined(__HIP_DEVICE_COMPILE__)
  f = __int_as_float(b);
#else
  std::memcpy(&f, &b, sizeof(f));
#endif
  return f;
}

__global__ void k_cube_bounds_init(int *__restrict__ cb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < LIO_MAP_MAX_VALID * 6) cb[i] = (i % 6) < 3 ? INT_MAX : INT_MIN;
}

// per-cube bounding box of the work list (the min_p/max_p pcl::VoxelGrid computes per cube cloud); LDS atomics per
// block, one global atomic per touched (cube, component)
__global__ void __launch_bounds__(256) k_cube_bounds(const float4 *__restrict__ u_pts, const uint32_t *__restrict__ u_rank, int n, int *__restrict__ cb) {
  __shared__ int sm[LIO_MAP_MAX_VALID * 6];
  for (int k = threadIdx.x; k < LIO_MAP_MAX_VALID * 6; k += blockDim.x) sm[k] = (k % 6) < 3 ? INT_MAX : INT_MIN;
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const uint32_t r = u_rank[i];
    if (r < LIO_MAP_MAX_VALID) {
      const float4 p = u_pts[i];
      int *b = sm + r * 6;
      atomicMin(b + 0, f2ord(p.x)); atomicMin(b + 1, f2ord(p.y)); atomicMin(b + 2, f2ord(p.z));
      atomicMax(b + 3, f2ord(p.x)); atomicMax(b + 4, f2ord(p.y)); atomicMax(b + 5, f2ord(p.z));
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < LIO_MAP_MAX_VALID * 6; k += blockDim.x) {
    const int v = sm[k];
    if ((k % 6) < 3) { if (v != INT_MAX) atomicMin(cb + k, v); }
    else if (v != INT_MIN) atomicMax(cb + k, v);
  }
}

__global__ void k_cube_vox_keys(const float4 *__restrict__ u_pts, const uint32_t *__restrict__ u_rank, int n, const int *__restrict__ cb,
                                float inv_leaf, unsigned long long *__restrict__ keys, uint32_t *__restrict__ vals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t r = u_rank[i];
  unsigned long long key = ~0ull;
  if (r < LIO_MAP_MAX_VALID) {
    const int *b = cb + r * 6;
    const int minb0 = int(floorf(ord2f(b[0]) * inv_leaf)), minb1 = int(floorf(ord2f(b[1]) * inv_leaf)), minb2 = int(floorf(ord2f(b[2]) * inv_leaf));
    const int div0 = int(floorf(ord2f(b[3]) * inv_leaf)) - minb0 + 1, div1 = int(floorf(ord2f(b[4]) * inv_leaf)) - minb1 + 1;
    const float4 p = u_pts[i];
    const int i0 = int(floorf(p.x * inv_leaf) - float(minb0));
    const int i1 = int(floorf(p.y * inv_leaf) - float(minb1));
    const int i2 = int(floorf(p.z * inv_leaf) - float(minb2));
    const unsigned int vk = static_cast<unsigned int>(i0 + i1 * div0 + i2 * div0 * div1);
    key = (static_cast<unsigned long long>(r) << 32) | vk;
  }
  keys[i] = key;
  vals[i] = uint32_t(i);
}

__global__ void k_heads64(const unsigned long long *__restrict__ keys, int n, int *__restrict__ flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long k = keys[i];
  flags[i] = (k != ~0ull && (i == 0 || keys[i - 1] != k)) ? 1 : 0;
}

// voxel centroids (x, y, z, intensity averaged; ascending (cube, voxel)) written behind the untouched pool part
__global__ void k_cube_centroids(const float4 *__restrict__ u_pts, const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ vals,
                                 const int *__restrict__ flags, const int *__restrict__ pos, int n, MapValidSet vs, int n_rest, MapCounters *cnt,
                                 float4 *__restrict__ pool, uint32_t *__restrict__ pkey, uint32_t *__restrict__ vrank) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (i == n - 1) cnt->n_out = pos[i] + flags[i];
  if (!flags[i]) return;
  const unsigned long long k = keys[i];
  float ax = 0, ay = 0, az = 0, ai = 0;
  int e = i;
  while (e < n && keys[e] == k) {  // stable sort: map points first, then the new ones, each in their own order
    const float4 p = u_pts[vals[e]];
    ax += p.x; ay += p.y; az += p.z; ai += p.w;
    ++e;
  }
  const float c = float(e - i);
  const uint32_t r = uint32_t(k >> 32);
  const int dst = n_rest + cnt->n_new_rest + pos[i];
  pool[dst] = make_float4(ax / c, ay / c, az / c, ai / c);
  pkey[dst] = vs.key[r];
  vrank[pos[i]] = r;
}

// stack round trip of PointMapping::Process (:789-801 then :991-1003): to the map with the predicted transform and
// back again, in float, exactly as the reference does before it voxel-filters the stack
__global__ void k_stack_roundtrip(const float4 *__restrict__ in, int n, Quat<float> q, Vec3<float> t, float4 *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 p = in[i];
  Vec3<float> m = rotate(q, Vec3<float>(p.x, p.y, p.z));
  m = Vec3<float>(m.x + t.x, m.y + t.y, m.z + t.z);
  Vec3<float> v(m.x - t.x, m.y - t.y, m.z - t.z);
  Vec3<float> r = rotate(conj(q), v);
  out[i] = make_float4(r.x, r.y, r.z, p.w);
}

// math_utils.h:186-203 (degrees)
inline Vec3<double> r2ypr_deg(const Mat3<double> &R) {
  const double y = std::atan2(R(1, 0), R(0, 0));
  const double p = std::atan2(-R(2, 0), R(0, 0) * std::cos(y) + R(1, 0) * std::sin(y));
  const double r = std::atan2(R(0, 2) * std::sin(y) - R(1, 2) * std::cos(y), -R(0, 1) * std::sin(y) + R(1, 1) * std::cos(y));
  return Vec3<double>(y / M_PI * 180.0, p / M_PI * 180.0, r / M_PI * 180.0);
}
inline Mat3<double> to_double(const Mat3<float> &m) { Mat3<double> r; for (int k = 0; k < 9; ++k) r.m[k] = double(m.m[k]); return r; }

template <typename T> T *pinned_alloc() {
  T *p = nullptr;
  LIO_HIP(hipHostMalloc(reinterpret_cast<void **>(&p), sizeof(T), hipHostMallocDefault));
  std::memset(p, 0, sizeof(T));
  return p;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
MappingDev::MappingDev(const lio_map_config &cfg) : cfg_(cfg) {
  int nd = 0;
  LIO_HIP(hipGetDeviceCount(&nd));
  if (nd <= 0) throw DeviceError("no HIP device: the product has no CPU path");
  LIO_HIP(hipStreamCreate(&stream_));
  LIO_HIP(hipStreamCreateWithFlags(&stream2_, hipStreamNonBlocking));
  LIO_HIP(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
  LIO_HIP(hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming));
  for (ClassMap &m : cls_) {
    m.h_counters = pinned_alloc<MapCounters>();
    m.h_bounds = pinned_alloc<VoxParams>();
    m.counters.reserve(1);
    m.bounds.reserve(1);
    m.cube_bounds.reserve(LIO_MAP_MAX_VALID * 6);
  }
  // coherent: the update kernel posts the state and a completion word here (dev.h: HostSignal)
  LIO_HIP(hipHostMalloc(reinterpret_cast<void **>(&h_state_), 128, hipHostMallocCoherent));
  static_assert(sizeof(OdomState) <= 64, "mailbox layout");
  std::memset(h_state_, 0, 128);
  h_flag_ = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(h_state_) + 64);
  d_state_.reserve(1);
}

MappingDev::~MappingDev() {
  for (ClassMap &m : cls_) {
    if (m.h_counters) (void)hipHostFree(m.h_counters);
    if (m.h_bounds) (void)hipHostFree(m.h_bounds);
  }
  if (h_state_) (void)hipHostFree(h_state_);
  if (ev_fork_) (void)hipEventDestroy(ev_fork_);
  if (ev_join_) (void)hipEventDestroy(ev_join_);
  if (stream2_) (void)hipStreamDestroy(stream2_);
  if (stream_) (void)hipStreamDestroy(stream_);
}

MapValidSet MappingDev::MakeValidSet(const uint32_t *valid_idx, size_t n, const int cen_of_idx[3]) const {
  MapValidSet vs;
  std::memset(&vs, 0, sizeof(vs));
  const int dims[3] = {L, Wd, H};
  for (int d = 0; d < 3; ++d) { vs.lo[d] = -cen_[d]; vs.hi[d] = dims[d] - cen_[d]; }
  std::vector<uint32_t> keys;
  for (size_t k = 0; k < n; ++k) {
    // FromIndex (PointMapping.h:153-160), re-based on the current centre (:1165-1179)
    const int residual = int(valid_idx[k] % uint32_t(L * Wd));
    const int ck = int(valid_idx[k] / uint32_t(L * Wd)), cj = residual / L, ci = residual % L;
    const int a[3] = {ci - cen_of_idx[0], cj - cen_of_idx[1], ck - cen_of_idx[2]};
    bool inside = true;
    for (int d = 0; d < 3; ++d) inside = inside && a[d] >= vs.lo[d] && a[d] < vs.hi[d];
    if (inside) keys.push_back(pack_cube(a[0], a[1], a[2]));
  }
  std::sort(keys.begin(), keys.end());
  keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
  if (keys.size() > LIO_MAP_MAX_VALID) throw std::runtime_error("PointMapping: more than 125 valid cubes");
  vs.n = int(keys.size());
  for (int k = 0; k < vs.n; ++k) vs.key[k] = keys[k];
  return vs;
}

bool MappingDev::LayoutMatches(const ClassMap &m, const MapValidSet &vs) const {
  if (!m.layout_ok || int(m.layout_keys.size()) != vs.n) return false;
  for (int d = 0; d < 3; ++d) if (m.layout_lo[d] != vs.lo[d] || m.layout_hi[d] != vs.hi[d]) return false;
  return std::equal(m.layout_keys.begin(), m.layout_keys.end(), vs.key);
}

void MappingDev::LayoutLaunch(ClassMap &m, const MapValidSet &vs) {
  hipStream_t s = stream_;
  std::memset(m.h_counters, 0, sizeof(MapCounters));
  if (m.n == 0) return;
  const size_t n = m.n;
  const int ni = int(n);
  m.rk.reserve(n); m.rk2.reserve(n); m.vals.reserve(n); m.vals2.reserve(n);
  m.pool2.reserve(n); m.pkey2.reserve(n); m.vrank.reserve(n);
  LIO_HIP(hipMemsetAsync(m.counters.p, 0, sizeof(MapCounters), s));
  hipLaunchKernelGGL(k_map_rank, dim3(cdiv(ni, 256)), dim3(256), 0, s, m.pkey.p, ni, vs, m.rk.p, m.vals.p, m.counters.p);
  size_t tb = 0;
  LIO_HIP(rocprim::radix_sort_pairs(nullptr, tb, m.rk.p, m.rk2.p, m.vals.p, m.vals2.p, n, 0, 8, s));
  m.tmp.reserve(tb + 256);
  LIO_HIP(rocprim::radix_sort_pairs(m.tmp.p, tb, m.rk.p, m.rk2.p, m.vals.p, m.vals2.p, n, 0, 8, s));
  hipLaunchKernelGGL(k_map_gather, dim3(cdiv(ni, 256)), dim3(256), 0, s, m.pool.p, m.pkey.p, m.rk2.p, m.vals2.p, ni, m.counters.p, m.pool2.p,
                     m.pkey2.p, m.vrank.p);
  LIO_HIP(hipGetLastError());
  LIO_HIP(hipMemcpyAsync(m.h_counters, m.counters.p, sizeof(MapCounters), hipMemcpyDeviceToHost, s));
}

void MappingDev::LayoutFinish(ClassMap &m, const MapValidSet &vs) {
  if (m.n != 0) {
    std::swap(m.pool, m.pool2);
    std::swap(m.pkey, m.pkey2);
  }
  m.n_valid = size_t(m.h_counters->n_valid);
  m.n_rest = size_t(m.h_counters->n_rest);
  m.n = m.n_valid + m.n_rest;
  m.layout_ok = true;
  m.layout_keys.assign(vs.key, vs.key + vs.n);
  for (int d = 0; d < 3; ++d) { m.layout_lo[d] = vs.lo[d]; m.layout_hi[d] = vs.hi[d]; }
}

// pool must already be laid out for vs.  new_sensor_pts: device, sensor frame.
void MappingDev::UpdateLaunch(ClassMap &m, const float4 *new_sensor_pts, size_t n_new, const MapValidSet &vs, const Rigid<float> &T, float leaf) {
  hipStream_t s = stream_;
  std::memset(m.h_counters, 0, sizeof(MapCounters));
  const size_t nV = m.n_valid, nU = nV + n_new;
  m.h_counters->n_out = int(nV);
  if (nU == 0) return;
  m.pool.reserve(m.n_rest + n_new + nU, s, true, m.n);
  m.pkey.reserve(m.n_rest + n_new + nU, s, true, m.n);
  m.u_pts.reserve(nU); m.u_rank.reserve(nU); m.k64.reserve(nU); m.k64b.reserve(nU); m.flags.reserve(nU); m.pos.reserve(nU);
  m.vals.reserve(nU); m.vals2.reserve(nU); m.vrank.reserve(nU, s, true, nV);
  LIO_HIP(hipMemsetAsync(m.counters.p, 0, sizeof(MapCounters), s));
  if (nV) {
    LIO_HIP(hipMemcpyAsync(m.u_pts.p, m.pool.p + m.n_rest, nV * sizeof(float4), hipMemcpyDeviceToDevice, s));
    LIO_HIP(hipMemcpyAsync(m.u_rank.p, m.vrank.p, nV * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
  }
  if (n_new) {
    const int nn = int(n_new);
    m.new_pts.reserve(n_new); m.new_key.reserve(n_new); m.rk.reserve(n_new); m.rk2.reserve(n_new);
    hipLaunchKernelGGL(k_map_new, dim3(cdiv(nn, 256)), dim3(256), 0, s, new_sensor_pts, nn, T.rot, T.pos, cen_[0], cen_[1], cen_[2], vs, m.new_pts.p,
                       m.new_key.p, m.rk.p, m.vals.p, m.counters.p);
    size_t tb = 0;
    LIO_HIP(rocprim::radix_sort_pairs(nullptr, tb, m.rk.p, m.rk2.p, m.vals.p, m.vals2.p, n_new, 0, 8, s));
    m.tmp.reserve(tb + 256);
    LIO_HIP(rocprim::radix_sort_pairs(m.tmp.p, tb, m.rk.p, m.rk2.p, m.vals.p, m.vals2.p, n_new, 0, 8, s));
    hipLaunchKernelGGL(k_new_scatter, dim3(cdiv(nn, 256)), dim3(256), 0, s, m.new_pts.p, m.new_key.p, m.rk2.p, m.vals2.p, nn, m.counters.p, int(nV),
                       int(m.n_rest), m.u_pts.p, m.u_rank.p, m.pool.p, m.pkey.p);
  }
  const int nu = int(nU);
  const float inv_leaf = 1.0f / leaf;
  hipLaunchKernelGGL(k_cube_bounds_init, dim3(cdiv(LIO_MAP_MAX_VALID * 6, 256)), dim3(256), 0, s, m.cube_bounds.p);
  hipLaunchKernelGGL(k_cube_bounds, dim3(cdiv(nu, 256)), dim3(256), 0, s, m.u_pts.p, m.u_rank.p, nu, m.cube_bounds.p);
  hipLaunchKernelGGL(k_cube_vox_keys, dim3(cdiv(nu, 256)), dim3(256), 0, s, m.u_pts.p, m.u_rank.p, nu, m.cube_bounds.p, inv_leaf,
                     reinterpret_cast<unsigned long long *>(m.k64.p), m.vals.p);
  size_t tb2 = 0, tb3 = 0;
  LIO_HIP(rocprim::radix_sort_pairs(nullptr, tb2, m.k64.p, m.k64b.p, m.vals.p, m.vals2.p, nU, 0, 40, s));
  LIO_HIP(rocprim::exclusive_scan(nullptr, tb3, m.flags.p, m.pos.p, 0, nU, rocprim::plus<int>(), s));
  m.tmp.reserve(std::max(tb2, tb3) + 256);
  LIO_HIP(rocprim::radix_sort_pairs(m.tmp.p, tb2, m.k64.p, m.k64b.p, m.vals.p, m.vals2.p, nU, 0, 40, s));
  hipLaunchKernelGGL(k_heads64, dim3(cdiv(nu, 256)), dim3(256), 0, s, reinterpret_cast<unsigned long long *>(m.k64b.p), nu, m.flags.p);
  LIO_HIP(rocprim::exclusive_scan(m.tmp.p, tb3, m.flags.p, m.pos.p, 0, nU, rocprim::plus<int>(), s));
  hipLaunchKernelGGL(k_cube_centroids, dim3(cdiv(nu, 256)), dim3(256), 0, s, m.u_pts.p, reinterpret_cast<unsigned long long *>(m.k64b.p), m.vals2.p,
                     m.flags.p, m.pos.p, nu, vs, int(m.n_rest), m.counters.p, m.pool.p, m.pkey.p, m.vrank.p);
  LIO_HIP(hipGetLastError());
  LIO_HIP(hipMemcpyAsync(m.h_counters, m.counters.p, sizeof(MapCounters), hipMemcpyDeviceToHost, s));
}

void MappingDev::UpdateFinish(ClassMap &m) {
  m.n_rest += size_t(m.h_counters->n_new_rest);
  m.n_valid = size_t(m.h_counters->n_out);
  m.n = m.n_rest + m.n_valid;
}

void MappingDev::UpdateMapDatabase(const float *corner_ds, size_t n_corner, const float *surf_ds, size_t n_surf, const uint32_t *valid_idx,
                                   size_t n_valid, const Rigid<float> &T, const int cube_center[3]) {
  hipStream_t s = stream_;
  const MapValidSet vs = MakeValidSet(valid_idx, n_valid, cube_center);
  bool relayout[2];
  for (int c = 0; c < 2; ++c) {
    relayout[c] = !LayoutMatches(cls_[c], vs);
    if (relayout[c]) LayoutLaunch(cls_[c], vs);
  }
  if (relayout[0] || relayout[1]) LIO_HIP(hipStreamSynchronize(s));
  for (int c = 0; c < 2; ++c) if (relayout[c]) LayoutFinish(cls_[c], vs);
  const float *src[2] = {corner_ds, surf_ds};
  const size_t cnt[2] = {n_corner, n_surf};
  const float leaf[2] = {cfg_.corner_filter_size, cfg_.surf_filter_size};
  for (int c = 0; c < 2; ++c) {
    ClassMap &m = cls_[c];
    if (cnt[c]) {
      m.in.reserve(cnt[c]);
      LIO_HIP(hipMemcpyAsync(m.in.p, src[c], cnt[c] * sizeof(float4), hipMemcpyHostToDevice, s));
    }
    UpdateLaunch(m, m.in.p, cnt[c], vs, T, leaf[c]);
  }
  LIO_HIP(hipStreamSynchronize(s));
  for (int c = 0; c < 2; ++c) UpdateFinish(cls_[c]);
}

// ------------------------------------------------------------------------------------------------
void MappingDev::Process(const float *corner_last, size_t n_corner, const float *surf_last, size_t n_surf, const Rigid<float> &sum) {
  hipStream_t s = stream_;
  const bool dbg = std::getenv("LIO_DEBUG_TIMING") != nullptr;
  const double tt0 = now_ms();
  transform_sum_ = sum;
  score_ready_ = false;
  iterations_ = 0; num_selected_ = 0; degenerate_ = false; kz_ = 0;
  const bool builder = cfg_.map_builder != 0;
  if (builder && !system_init_) {  // MapBuilder::ProcessMap (MapBuilder.cc:227-232)
    system_init_ = true;
    transform_bef_mapped_ = transform_tobe_mapped_ = transform_aft_mapped_ = transform_sum_;
  }
  if (builder || !imu_inited_) {  // TransformAssociateToMap (:755-758) / Transform4DAssociateToMap (MapBuilder.cc:55-75)
    const Rigid<float> sumT = fromAffine(linearOf(transform_sum_), transform_sum_.pos);
    const Rigid<float> incre = compose(rinverse(transform_bef_mapped_), sumT);
    const Rigid<float> full = compose(transform_tobe_mapped_, incre);
    if (builder && cfg_.enable_4d) {
      // keep the odometry's roll/pitch, take only the yaw of the increment
      const Vec3<double> r0 = r2ypr_deg(to_double(toRot(normalized(full.rot))));
      const Vec3<double> r00 = r2ypr_deg(to_double(toRot(normalized(transform_sum_.rot))));
      const float y = float(double(float(r0.x - r00.x)) / 180.0 * M_PI);
      Mat3<float> Rz = Mat3<float>::identity();
      Rz(0, 0) = std::cos(y); Rz(0, 1) = -std::sin(y); Rz(1, 0) = std::sin(y); Rz(1, 1) = std::cos(y);
      transform_tobe_mapped_.pos = full.pos;
      transform_tobe_mapped_.rot = fromRot(Rz * toRot(normalized(transform_sum_.rot)));
    } else {
      transform_tobe_mapped_ = full;
    }
  }
  const Rigid<float> T0 = transform_tobe_mapped_;
  // stack clouds go up while the host does the cube bookkeeping
  const float *src[2] = {corner_last, surf_last};
  const size_t cnt[2] = {n_corner, n_surf};
  for (int c = 0; c < 2; ++c) {
    ClassMap &m = cls_[c];
    m.n_stack = 0;
    if (!cnt[c]) continue;
    m.in.reserve(cnt[c]); m.stack_raw.reserve(cnt[c]);
    LIO_HIP(hipMemcpyAsync(m.in.p, src[c], cnt[c] * sizeof(float4), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_stack_roundtrip, dim3(cdiv(int(cnt[c]), 256)), dim3(256), 0, s, m.in.p, int(cnt[c]), T0.rot, T0.pos, m.stack_raw.p);
  }
  {
    Vec3<float> z = rotate(T0.rot, Vec3<float>(0.f, 0.f, 10.f));
    pz_[0] = z.x + T0.pos.x; pz_[1] = z.y + T0.pos.y; pz_[2] = z.z + T0.pos.z;
  }
  // sensor cube and window shift (:808-925); absolute keys make the shift a change of cen_ only
  const float posv[3] = {T0.pos.x, T0.pos.y, T0.pos.z};
  const int dims[3] = {L, Wd, H};
  int cc[3];
  for (int d = 0; d < 3; ++d) {
    cc[d] = cube_coord(posv[d], cen_[d]);
    while (cc[d] < 3) { ++cc[d]; ++cen_[d]; }
    while (cc[d] >= dims[d] - 3) { --cc[d]; --cen_[d]; }
  }
  // cubes of the 5x5x5 neighbourhood with a corner inside the +-60 deg cone (:938-989)
  valid_idx_.clear();
  auto sqdiff = [](const float a[3], const float b[3]) {
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return dx * dx + dy * dy + dz * dz;
  };
  for (int i = cc[0] - 2; i <= cc[0] + 2; ++i)
    for (int j = cc[1] - 2; j <= cc[1] + 2; ++j)
      for (int k = cc[2] - 2; k <= cc[2] + 2; ++k) {
        if (!(i >= 0 && i < L && j >= 0 && j < Wd && k >= 0 && k < H)) continue;
        const float cx = 50.0f * (i - cen_[0]), cy = 50.0f * (j - cen_[1]), cz = 50.0f * (k - cen_[2]);
        bool fov = false;
        for (int ii = -1; ii <= 1; ii += 2)
          for (int jj = -1; jj <= 1; jj += 2)
            for (int kk = -1; kk <= 1; kk += 2) {
              const float corner[3] = {cx + 25.0f * ii, cy + 25.0f * jj, cz + 25.0f * kk};
              const float s1 = sqdiff(posv, corner), s2 = sqdiff(pz_, corner);
              const float check1 = 100.0f + s1 - s2 - 10.0f * std::sqrt(3.0f) * std::sqrt(s1);
              const float check2 = 100.0f + s1 - s2 + 10.0f * std::sqrt(3.0f) * std::sqrt(s1);
              if (check1 < 0 && check2 > 0) fov = true;
            }
        if (fov) valid_idx_.push_back(uint32_t(i + L * j + L * Wd * k));
      }
  const MapValidSet vs = MakeValidSet(valid_idx_.data(), valid_idx_.size(), cen_);

  // laser_cloud_*_from_map_ = tail of the pool after the layout pass
  bool relayout[2];
  for (int c = 0; c < 2; ++c) {
    relayout[c] = !LayoutMatches(cls_[c], vs);
    if (relayout[c]) LayoutLaunch(cls_[c], vs);
  }
  if (relayout[0] || relayout[1]) LIO_HIP(hipStreamSynchronize(s));
  for (int c = 0; c < 2; ++c) if (relayout[c]) LayoutFinish(cls_[c], vs);

  const double tt1 = now_ms();
  // VoxelGrid of the stacks (:1005-1015)
  const float leaf[2] = {cfg_.corner_filter_size, cfg_.surf_filter_size};
  // the two filters are independent: the surf one goes to a second stream, forked after the round-trip kernels above
  LIO_HIP(hipEventRecord(ev_fork_, s));
  LIO_HIP(hipStreamWaitEvent(stream2_, ev_fork_, 0));
  hipStream_t vs_stream[2] = {s, stream2_};
  for (int c = 0; c < 2; ++c) cls_[c].vox.launch(cls_[c].stack_raw.p, cnt[c], leaf[c], cls_[c].stack_ds, vs_stream[c]);
  // finish() only waits for the output COUNT (posted to the host's mailbox); the consumers on `s` wait for the surf filter's
  // stream through an event
  LIO_HIP(hipEventRecord(ev_join_, stream2_));
  LIO_HIP(hipStreamWaitEvent(s, ev_join_, 0));
  for (int c = 0; c < 2; ++c) cls_[c].n_stack = cls_[c].vox.finish();

  const double tt2 = now_ms();
  if (builder) {  // MapBuilder.cc:527-558: optimise every skip_count-th call, always update the map
    if (odom_count_ % cfg_.skip_count == 0) Optimize(cfg_.enable_4d != 0);
    else {
      n_from_map_[0] = cls_[0].n_valid; n_from_map_[1] = cls_[1].n_valid; from_map_in_u_ = false;
      transform_bef_mapped_ = transform_sum_; transform_aft_mapped_ = transform_tobe_mapped_;
    }
    ++odom_count_;
  } else {
    Optimize(false);
  }

  const double tt3 = now_ms();
  if (builder || !imu_inited_) {
    for (int c = 0; c < 2; ++c) UpdateLaunch(cls_[c], cls_[c].stack_ds.p, cls_[c].n_stack, vs, transform_tobe_mapped_, leaf[c]);
    LIO_HIP(hipStreamSynchronize(s));
    for (int c = 0; c < 2; ++c) UpdateFinish(cls_[c]);
    from_map_in_u_ = true;
  }
  if (dbg)
    std::fprintf(stderr, "[lio_hip map timing] upload+layout %.3f  stack voxel %.3f  optimise %.3f (%d rounds)  map update %.3f  total %.3f ms\n", tt1 - tt0,
                 tt2 - tt1, tt3 - tt2, iterations_, now_ms() - tt3, now_ms() - tt0);
}

void MappingDev::Optimize(bool four_dof) {
  hipStream_t s = stream_;
  ClassMap &mc = cls_[0], &ms = cls_[1];
  n_from_map_[0] = mc.n_valid; n_from_map_[1] = ms.n_valid;
  from_map_in_u_ = false;
  if (mc.n_valid <= 10 || ms.n_valid <= 100) return;  // :327-329 (no TransformUpdate either)
  // search grids over the two from-map clouds
  ClassMap *cm[2] = {&mc, &ms};
  for (ClassMap *m : cm) {
    launch_cloud_bounds(m->pool.p + m->n_rest, int(m->n_valid), m->partial, m->bounds.p, s);
    LIO_HIP(hipMemcpyAsync(m->h_bounds, m->bounds.p, sizeof(VoxParams), hipMemcpyDeviceToHost, s));
  }
  LIO_HIP(hipStreamSynchronize(s));
  const float cell = std::sqrt(cfg_.min_match_sq_dis) * 1.0001f;
  for (ClassMap *m : cm) m->grid.build(m->pool.p + m->n_rest, m->n_valid, m->h_bounds->mn, m->h_bounds->mx, cell, s);

  const int Mc = int(mc.n_stack), Ms = int(ms.n_stack), M = Mc + Ms;
  if (M == 0) {  // every round has < 50 rows: the loop runs dry, then TransformUpdate
    iterations_ = cfg_.num_max_iterations;
    transform_bef_mapped_ = transform_sum_; transform_aft_mapped_ = transform_tobe_mapped_;
    return;
  }
  stack_all_.reserve(size_t(M)); f_valid_.reserve(size_t(M)); f_coef_.reserve(size_t(M)); f_abs_.reserve(size_t(M));
  if (Mc) LIO_HIP(hipMemcpyAsync(stack_all_.p, mc.stack_ds.p, size_t(Mc) * sizeof(float4), hipMemcpyDeviceToDevice, s));
  if (Ms) LIO_HIP(hipMemcpyAsync(stack_all_.p + Mc, ms.stack_ds.p, size_t(Ms) * sizeof(float4), hipMemcpyDeviceToDevice, s));
  OdomState st;
  std::memset(&st, 0, sizeof(st));
  const Rigid<float> &T = transform_tobe_mapped_;
  st.T[0] = T.rot.x; st.T[1] = T.rot.y; st.T[2] = T.rot.z; st.T[3] = T.rot.w; st.T[4] = T.pos.x; st.T[5] = T.pos.y; st.T[6] = T.pos.z;
  *h_state_ = st;
  LIO_HIP(hipMemcpyAsync(d_state_.p, h_state_, sizeof(OdomState), hipMemcpyHostToDevice, s));
  const float *d_T = reinterpret_cast<const float *>(d_state_.p);
  const int *d_conv = reinterpret_cast<const int *>(reinterpret_cast<const char *>(d_state_.p) + offsetof(OdomState, converged));
  const int nb = odom_rows_blocks(M);
  d_partials_.reserve(size_t(nb) * 28);
  FeatArgs fa{};
  fa.nframes = 1; fa.max_M = Ms;
  fa.fr[0].stack = stack_all_.p + Mc; fa.fr[0].M = Ms; fa.fr[0].slot_off = Mc; fa.fr[0].tf_index = 0;
  fa.min_match_sq_dis = cfg_.min_match_sq_dis; fa.min_plane_dis = cfg_.min_plane_dis;
  fa.mapping_mode = four_dof ? 2 : 1;
  for (int d = 0; d < 3; ++d) fa.fixed_pz[d] = pz_[d];
  const int max_it = cfg_.num_max_iterations;
  int iter = 0;
  // Convergence is read back at these rounds only: a peek costs a sync + D2H + relaunch bubble (~35 us), a round that runs
  // after convergence costs four no-op launches (~18 us), and the loop typically needs 5-7 rounds.
  static const int kPeek[] = {6, 8, 10, 10};
  int peek_i = 0;
  bool done = false;
  while (iter < max_it && !done) {
    int until = max_it;
    while (peek_i < 4 && kPeek[peek_i] <= iter) ++peek_i;
    if (peek_i < 4) until = std::min(max_it, kPeek[peek_i]);
    HostSignal sig{};
    for (; iter < until; ++iter) {
      // the corner (line) and surf (plane) searches of a round are independent: one launch, blockIdx.y picks the branch
      launch_map_round(fa, stack_all_.p, Mc, d_T, mc.grid.sorted(), mc.grid.cells(), mc.grid.desc(), ms.grid.sorted(), ms.grid.cells(), ms.grid.desc(),
                       f_valid_.p, f_coef_.p, f_abs_.p, d_conv, s);
      launch_odom_rows(stack_all_.p, M, M, f_valid_.p, f_coef_.p, d_state_.p, d_partials_.p, nb, s, four_dof ? 2 : 1);
      // the last round before a look at the convergence flag posts the state to the host's mailbox (dev.h: HostSignal)
      if (iter == until - 1 && host_signal_enabled()) { sig.flag = h_flag_; sig.seq = ++seq_; }
      launch_odom_update(d_partials_.p, nb, d_state_.p, iter, s, 50, four_dof ? 1 : 0, h_state_, sig);
    }
    if (sig.flag) {
      wait_host_signal(sig, s);
    } else {
      LIO_HIP(hipMemcpyAsync(h_state_, d_state_.p, sizeof(OdomState), hipMemcpyDeviceToHost, s));
      LIO_HIP(hipStreamSynchronize(s));
    }
    if (h_state_->converged) done = true;
  }
  st = *h_state_;
  transform_tobe_mapped_.rot = Quat<float>(st.T[3], st.T[0], st.T[1], st.T[2]);
  transform_tobe_mapped_.pos = Vec3<float>(st.T[4], st.T[5], st.T[6]);
  iterations_ = st.iters;
  num_selected_ = st.nsel;
  degenerate_ = st.degenerate != 0;
  kz_ = degenerate_ ? st.kz : 0;
  transform_bef_mapped_ = transform_sum_;            // TransformUpdate (:760-763)
  transform_aft_mapped_ = transform_tobe_mapped_;
  n_score_slots_ = size_t(M);
  score_ready_ = !four_dof;  // OptimizeMap keeps no score list
}

// ------------------------------------------------------------------------------------------------
size_t MappingDev::GetCloud(int which, float *out) {
  hipStream_t s = stream_;
  const ClassMap &m = cls_[which & 1];
  const float4 *src = nullptr;
  size_t n = 0;
  if (which < 2) { src = m.stack_ds.p; n = m.n_stack; }
  else {
    n = n_from_map_[which & 1];
    src = from_map_in_u_ ? m.u_pts.p : m.pool.p + m.n_rest;
  }
  if (out && n) {
    LIO_HIP(hipMemcpyAsync(out, src, n * sizeof(float4), hipMemcpyDeviceToHost, s));
    LIO_HIP(hipStreamSynchronize(s));
  }
  return n;
}

size_t MappingDev::GetCube(int cls, uint32_t cube_idx, float *out) {
  hipStream_t s = stream_;
  const ClassMap &m = cls_[cls];
  if (m.n == 0) return 0;
  const int residual = int(cube_idx % uint32_t(L * Wd));
  const int ck = int(cube_idx / uint32_t(L * Wd)), cj = residual / L, ci = residual % L;
  const uint32_t key = pack_cube(ci - cen_[0], cj - cen_[1], ck - cen_[2]);
  std::vector<uint32_t> keys(m.n);
  std::vector<float4> pts(m.n);
  LIO_HIP(hipMemcpyAsync(keys.data(), m.pkey.p, m.n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  LIO_HIP(hipMemcpyAsync(pts.data(), m.pool.p, m.n * sizeof(float4), hipMemcpyDeviceToHost, s));
  LIO_HIP(hipStreamSynchronize(s));
  size_t cnt = 0;
  for (size_t i = 0; i < m.n; ++i)
    if (keys[i] == key) {
      if (out) std::memcpy(out + 4 * cnt, &pts[i], sizeof(float4));
      ++cnt;
    }
  return cnt;
}

// score_point_coeff_ (:725-750): surf selections of the last executed round, descending score
size_t MappingDev::GetScorePointCoeff(float *score, float *point, float *coeff) {
  if (!score_ready_) return 0;
  hipStream_t s = stream_;
  const size_t Mc = cls_[0].n_stack, Ms = cls_[1].n_stack;
  if (Ms == 0) return 0;
  std::vector<uint8_t> valid(Ms);
  std::vector<float4> coef(Ms), absc(Ms), pts(Ms);
  LIO_HIP(hipMemcpyAsync(valid.data(), f_valid_.p + Mc, Ms, hipMemcpyDeviceToHost, s));
  LIO_HIP(hipMemcpyAsync(coef.data(), f_coef_.p + Mc, Ms * sizeof(float4), hipMemcpyDeviceToHost, s));
  LIO_HIP(hipMemcpyAsync(absc.data(), f_abs_.p + Mc, Ms * sizeof(float4), hipMemcpyDeviceToHost, s));
  LIO_HIP(hipMemcpyAsync(pts.data(), stack_all_.p + Mc, Ms * sizeof(float4), hipMemcpyDeviceToHost, s));
  LIO_HIP(hipStreamSynchronize(s));
  struct E { float sc; uint32_t i; };
  std::vector<E> e;
  for (size_t i = 0; i < Ms; ++i)
    if (valid[i]) e.push_back({std::sqrt(coef[i].x * coef[i].x + coef[i].y * coef[i].y + coef[i].z * coef[i].z), uint32_t(i)});
  if (e.size() < 50) return 0;
  std::stable_sort(e.begin(), e.end(), [](const E &a, const E &b) { return a.sc > b.sc; });
  for (size_t k = 0; k < e.size(); ++k) {
    if (score) score[k] = e[k].sc;
    if (point) std::memcpy(point + 4 * k, &pts[e[k].i], sizeof(float4));
    if (coeff) std::memcpy(coeff + 4 * k, &absc[e[k].i], sizeof(float4));
  }
  return e.size();
}

}  // namespace lio
