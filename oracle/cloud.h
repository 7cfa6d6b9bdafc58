// oracle/cloud.h — TEST INFRASTRUCTURE ONLY (CPU oracle).  Not part of the product.
//
// Restates the PCL 1.8 semantics the hot path uses.  PCL/FLANN are a
// third-party dependency absent from /root/reference (ROS melodic default PCL 1.8.1, README.md:22-24)
// so parity with PCL itself is UNPINNED; the algorithms below are the documented ones:
//   pcl::VoxelGrid<PointXYZI>::applyFilter  — call sites PointProcessor.cc:738-749, Estimator.cc:678-687,1518-1519
//   pcl::KdTreeFLANN::nearestKSearch        — call sites Estimator.cc:1019,1544-1545, PointOdometry.cc:345,444
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

namespace orc {

struct P4 {
  float x, y, z, i;
};
typedef std::vector<P4> Cloud;

// pcl::VoxelGrid, downsample_all_data=true, min_points_per_voxel=0 (B.1).  Within-voxel accumulation
// order: PCL uses an unstable std::sort on the voxel index, so the order is unspecified there; the
// oracle (and the HIP path) fix it to ascending input index (stable sort).
inline void VoxelGrid(const Cloud &in, float leaf, Cloud &out) {
  out.clear();
  if (in.empty()) return;
  float inv = 1.0f / leaf;  // inverse_leaf_size_ = Array4f::Ones() / leaf_size_
  float mn[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
  float mx[3] = {-std::numeric_limits<float>::max(), -std::numeric_limits<float>::max(), -std::numeric_limits<float>::max()};
  for (const P4 &p : in) {
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
    mn[0] = std::min(mn[0], p.x); mn[1] = std::min(mn[1], p.y); mn[2] = std::min(mn[2], p.z);
    mx[0] = std::max(mx[0], p.x); mx[1] = std::max(mx[1], p.y); mx[2] = std::max(mx[2], p.z);
  }
  int64_t dx = int64_t((mx[0] - mn[0]) * inv) + 1, dy = int64_t((mx[1] - mn[1]) * inv) + 1, dz = int64_t((mx[2] - mn[2]) * inv) + 1;
  if (dx * dy * dz > int64_t(std::numeric_limits<int32_t>::max())) { out = in; return; }  // PCL warns and copies
  int minb[3], maxb[3];
  for (int d = 0; d < 3; ++d) {
    minb[d] = int(std::floor(mn[d] * inv));
    maxb[d] = int(std::floor(mx[d] * inv));
  }
  int divb[3] = {maxb[0] - minb[0] + 1, maxb[1] - minb[1] + 1, maxb[2] - minb[2] + 1};
  int mul[3] = {1, divb[0], divb[0] * divb[1]};
  struct KI { int key; uint32_t idx; };
  std::vector<KI> ki;
  ki.reserve(in.size());
  for (uint32_t n = 0; n < in.size(); ++n) {
    const P4 &p = in[n];
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
    int i0 = int(std::floor(p.x * inv) - float(minb[0]));
    int i1 = int(std::floor(p.y * inv) - float(minb[1]));
    int i2 = int(std::floor(p.z * inv) - float(minb[2]));
    ki.push_back({i0 * mul[0] + i1 * mul[1] + i2 * mul[2], n});
  }
  std::stable_sort(ki.begin(), ki.end(), [](const KI &a, const KI &b) { return a.key < b.key; });
  size_t s = 0;
  while (s < ki.size()) {
    size_t e = s;
    float ax = 0, ay = 0, az = 0, ai = 0;  // AccumulatorXYZ / AccumulatorIntensity are float
    while (e < ki.size() && ki[e].key == ki[s].key) {
      const P4 &p = in[ki[e].idx];
      ax += p.x; ay += p.y; az += p.z; ai += p.i;
      ++e;
    }
    float n = float(e - s);
    out.push_back({ax / n, ay / n, az / n, ai / n});
    s = e;
  }
}

// Exact k-NN over a static cloud; results ascending by (squared distance, index) (B.2).
// kd-tree with leaf size 15 like FLANN's KDTreeSingleIndex; L2_Simple accumulation order
// ((dx*dx + dy*dy) + dz*dz) in float.
class KdTree {
 public:
  void Build(const Cloud &c) {
    pts_ = &c;
    idx_.resize(c.size());
    for (size_t i = 0; i < c.size(); ++i) idx_[i] = int(i);
    nodes_.clear();
    if (!c.empty()) build(0, int(c.size()));
  }
  // returns number found (<= k)
  int Search(const P4 &q, int k, int *out_idx, float *out_sqd) const {
    struct Cand { float d; int i; };
    Cand best[16];
    int nb = 0;
    if (nodes_.empty()) return 0;
    search(0, q, k, best, nb);
    for (int j = 0; j < nb; ++j) { out_idx[j] = best[j].i; out_sqd[j] = best[j].d; }
    return nb;
  }
  static inline float sqd(const P4 &a, const P4 &b) {
    float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    float r = dx * dx;
    r += dy * dy;
    r += dz * dz;
    return r;
  }

 private:
  struct Node { int lo, hi, left, right, dim; float split; };
  const Cloud *pts_ = nullptr;
  std::vector<int> idx_;
  std::vector<Node> nodes_;
  float coord(int i, int d) const { const P4 &p = (*pts_)[i]; return d == 0 ? p.x : (d == 1 ? p.y : p.z); }
  int build(int lo, int hi) {
    int id = int(nodes_.size());
    nodes_.push_back({lo, hi, -1, -1, 0, 0.f});
    if (hi - lo <= 15) return id;
    float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
    for (int i = lo; i < hi; ++i)
      for (int d = 0; d < 3; ++d) { float v = coord(idx_[i], d); mn[d] = std::min(mn[d], v); mx[d] = std::max(mx[d], v); }
    int dim = 0;
    if (mx[1] - mn[1] > mx[dim] - mn[dim]) dim = 1;
    if (mx[2] - mn[2] > mx[dim] - mn[dim]) dim = 2;
    int mid = (lo + hi) / 2;
    std::nth_element(idx_.begin() + lo, idx_.begin() + mid, idx_.begin() + hi,
                     [&](int a, int b) { return coord(a, dim) < coord(b, dim); });
    float split = coord(idx_[mid], dim);
    int l = build(lo, mid);
    int r = build(mid, hi);
    nodes_[id].left = l; nodes_[id].right = r; nodes_[id].dim = dim; nodes_[id].split = split;
    return id;
  }
  template <typename Cand>
  void search(int id, const P4 &q, int k, Cand *best, int &nb) const {
    const Node &n = nodes_[id];
    if (n.left < 0) {
      for (int i = n.lo; i < n.hi; ++i) {
        int pi = idx_[i];
        float d = sqd((*pts_)[pi], q);
        if (nb < k || d < best[nb - 1].d || (d == best[nb - 1].d && pi < best[nb - 1].i)) {
          int pos = (nb < k) ? nb++ : k - 1;
          while (pos > 0 && (best[pos - 1].d > d || (best[pos - 1].d == d && best[pos - 1].i > pi))) { best[pos] = best[pos - 1]; --pos; }
          best[pos] = {d, pi};
        }
      }
      return;
    }
    float qv = n.dim == 0 ? q.x : (n.dim == 1 ? q.y : q.z);
    float diff = qv - n.split;
    int first = diff < 0 ? n.left : n.right, second = diff < 0 ? n.right : n.left;
    search(first, q, k, best, nb);
    // conservative bound (<=) keeps equal-distance candidates reachable for the index tiebreak
    if (nb < k || diff * diff <= best[nb - 1].d) search(second, q, k, best, nb);
  }
};

}  // namespace orc
