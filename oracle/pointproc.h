// oracle/pointproc.h — TEST INFRASTRUCTURE ONLY (CPU oracle).  Not part of the product.
//
// CPU restatement of lio::PointProcessor (evenly spaced rings, uneven=false):
//   src/point_processor/PointProcessor.cc:185-205 (PointToRing wrapper), :207-426 (binning + rel time),
//   :542-585 (PrepareRing), :587-622 (PrepareSubregion), :624-645 (MaskPickedInRing),
//   :647-783 (ExtractFeaturePoints); include/point_processor/PointProcessor.h:104-120,153-156.
// Float/double mixing follows SURVEY.md Appendix A.1/A.2 literally.  Parity object: the ordered
// (ring, in-ring index) pick lists (bit-exact) and the clouds.
// Pinned (round 3) against the reference's own PointProcessor.cc compiled where it lies (oracle/ref_pointproc.cc, oracle/ref_shim,
// `make ref`): tests/golden/ref_pointproc_digests.json, tests/test_ref_pointproc_digests.py — bit for bit, in order.
#pragma once
#include "cloud.h"
#include "liomath.h"

namespace orc {

struct PPConfig {
  double scan_period = 0.1;
  int num_scan_subregions = 8;
  int num_curvature_regions = 5;
  float surf_curv_th = 0.1f;
  int max_corner_sharp = 2;
  int max_corner_less_sharp = 20;
  int max_surf_flat = 4;
  float less_flat_filter_size = 0.2f;
  bool infer_start_ori_ = false;   // PointProcessor.h:119
  double rad_diff = 0.2;           // PointProcessor.h:117
};

// PointProcessor.cc:70-72; NormalizeRad (liomath.h) is instantiated with T = float at PointProcessor.cc:354-364
inline double AbsRadDistance(double a, double b) { return std::fabs(NormalizeRad(a - b)); }

// utils/CircularBuffer.h:118-172 — push overwrites the oldest element once full; [i] counts from the oldest
struct Ring10 {
  float buf[10];
  size_t size_ = 0, start_ = 0;
  void push(float v) { if (size_ < 10) buf[size_++] = v; else { buf[start_] = v; start_ = (start_ + 1) % 10; } }
  float operator[](size_t i) const { return buf[(start_ + i) % 10]; }
  float first() const { return buf[start_]; }
  float last() const { return buf[size_ == 0 ? 0 : (start_ + size_ - 1) % 10]; }
  size_t size() const { return size_; }
};

struct PointProcessor {
  float lower_bound_, upper_bound_, factor_;
  int num_rings_;
  PPConfig config_;
  float start_ori_ = std::nanf("");
  Ring10 start_ori_buf1_, start_ori_buf2_;   // PointProcessor.h: CircularBuffer<float>{10} each

  std::vector<Cloud> laser_scans;   // intensity = ring + rel_time
  std::vector<std::vector<float>> intensity_scans;   // the intensity channel of the reference's intensity_scans: int(input intensity) + rel_time (:413)
  std::vector<float> intensity_rings;                // ... concatenated in ring order (= cloud_in_rings_'s intensities, :195)
  std::vector<int> ring_offsets;    // rings+1
  Cloud cloud_rings;                // laser_scans concatenated
  std::vector<float> curvature;     // per cloud_rings point
  std::vector<int> mask;            // final scan_ring_mask_ per cloud_rings point
  Cloud sharp, less_sharp, flat, less_flat;
  std::vector<int> pick_ring[4], pick_idx[4];  // index by LIO_PP_* (1..3)

  PointProcessor(float lo, float up, int rings) : lower_bound_(lo), upper_bound_(up), num_rings_(rings) {
    factor_ = (rings - 1) / (up - lo);  // PointProcessor.cc:79 (float)
  }

  // PointProcessor.h:153-156 — float RadToDeg, float subtraction/multiply, +0.5 in double, trunc
  int ElevationToRing(float rad) const { return int((RadToDeg(rad) - lower_bound_) * factor_ + 0.5); }

  // ring != nullptr selects the PointIR overload of PointToRing (uneven_ == true, PointProcessor.cc:185-190)
  void Process(const float *xyzi, size_t n, const uint16_t *ring = nullptr) {
    laser_scans.assign(num_rings_, Cloud());
    intensity_scans.assign(num_rings_, std::vector<float>());
    sharp.clear(); less_sharp.clear(); flat.clear(); less_flat.clear();
    for (int k = 0; k < 4; ++k) { pick_ring[k].clear(); pick_idx[k].clear(); }
    if (ring) PointToRingIR(xyzi, ring, n); else PointToRing(xyzi, n);
    ExtractFeaturePoints();
  }

  // PointProcessor.cc:428-536: ring from the point's own field; azimuths behind the first one are unwrapped by 2 pi
  // (half_passed is never set: its condition needs i > 3 * cloud_size / 2, :487); end_ori_ = largest unwrapped azimuth
  // (from 0, :439,495-497); rel_time = scan_period * (azimuth - start_ori_) / (end_ori_ - start_ori_) (:507-524).
  void PointToRingIR(const float *xyzi, const uint16_t *ring_field, size_t n) {
    bool start_flag = false;
    start_ori_ = 0.f;
    float end_ori = 0.f;
    for (size_t i = 0; i < n; ++i) {
      P4 p{xyzi[4 * i], xyzi[4 * i + 1], xyzi[4 * i + 2], xyzi[4 * i + 3]};
      if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;  // :456-460
      float azi_rad = float(2 * M_PI - std::atan2(p.y, p.x));                            // :464
      if (azi_rad >= 2 * M_PI) azi_rad = float(azi_rad - 2 * M_PI);                      // :466-468
      int scan_id = ring_field[i];                                                        // :470
      if (scan_id >= num_rings_ || scan_id < 0) continue;                                 // :472-476
      if (!start_flag) { start_ori_ = azi_rad; start_flag = true; }                       // :478-481
      float azi_rad_rel = azi_rad - start_ori_;
      if (azi_rad_rel < 0) azi_rad = float(azi_rad + 2 * M_PI);                           // :486-489 (half_passed stays false)
      if (end_ori < azi_rad) end_ori = azi_rad;                                           // :495-497
      intensity_scans[scan_id].push_back(p.i);
      p.i = azi_rad;
      laser_scans[scan_id].push_back(p);
    }
    const float range_ori = end_ori - start_ori_;                                         // :507
    for (int ring = 0; ring < num_rings_; ++ring)
      for (size_t k = 0; k < laser_scans[ring].size(); ++k) {
        P4 &p = laser_scans[ring][k];
        float azi_rad_rel = p.i - start_ori_;
        float rel_time = float(config_.scan_period * azi_rad_rel / range_ori);            // :521 (double * float / float)
        p.i = ring + rel_time;
        intensity_scans[ring][k] = int(intensity_scans[ring][k]) + rel_time;              // :524
      }
    FinishRings();
  }

  void PointToRing(const float *xyzi, size_t n) {
    bool start_flag = false;
    for (size_t i = 0; i < n; ++i) {
      P4 p{xyzi[4 * i], xyzi[4 * i + 1], xyzi[4 * i + 2], xyzi[4 * i + 3]};
      if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;  // :240-244
      float dis = std::sqrt(p.x * p.x + p.y * p.y);   // :246
      float ele_rad = std::atan2(p.z, dis);           // :247 float overload
      float azi_rad = float(2 * M_PI - std::atan2(p.y, p.x));  // :248 double - float -> float
      if (azi_rad >= 2 * M_PI) azi_rad = float(azi_rad - 2 * M_PI);  // :251-253
      int scan_id = ElevationToRing(ele_rad);
      if (scan_id >= num_rings_ || scan_id < 0) continue;  // :257-259
      if (!start_flag) { start_ori_ = azi_rad; start_flag = true; }  // :261-264
      intensity_scans[scan_id].push_back(p.i);
      p.i = azi_rad;  // :268
      laser_scans[scan_id].push_back(p);
    }
    if (config_.infer_start_ori_) {  // :348-387
      start_ori_buf2_.push(start_ori_);
      if (start_ori_buf1_.size() >= 10) {
        float start_ori_diff1 = start_ori_buf1_.last() - start_ori_buf1_.first();
        float start_ori_step1 = NormalizeRad(start_ori_diff1) / 9;
        float start_ori_diff2 = start_ori_buf2_.last() - start_ori_buf2_.first();
        float start_ori_step2 = NormalizeRad(start_ori_diff2) / 9;
        if (std::fabs(NormalizeRad(start_ori_ - start_ori_buf1_.last())) > config_.rad_diff) {
          start_ori_ = start_ori_buf1_.last() + start_ori_step1;
          start_ori_ = NormalizeRad(start_ori_);
          if (start_ori_ < 0) start_ori_ += 2 * M_PI;
        }
        bool steady = AbsRadDistance(start_ori_step1, start_ori_step2) < 0.05;
        for (int k = 9; k >= 1; --k) steady = steady && AbsRadDistance(start_ori_buf2_[k] - start_ori_buf2_[k - 1], start_ori_step1) < 0.05;
        // :383 reads ring_out[0]->front() unguarded; an empty ring 0 leaves the value alone here
        if (steady && !laser_scans[0].empty()) start_ori_ = laser_scans[0].front().i;
      }
      start_ori_buf1_.push(start_ori_);
    }
    // :393-423 second pass: intensity = ring + rel_time
    for (int ring = 0; ring < num_rings_; ++ring)
      for (size_t k = 0; k < laser_scans[ring].size(); ++k) {
        P4 &p = laser_scans[ring][k];
        float azi_rad_rel = p.i - start_ori_;
        if (azi_rad_rel < 0) azi_rad_rel = float(azi_rad_rel + 2 * M_PI);
        float rel_time = float(config_.scan_period * azi_rad_rel / (2 * M_PI));
        p.i = ring + rel_time;
        intensity_scans[ring][k] = int(intensity_scans[ring][k]) + rel_time;   // :413
      }
    FinishRings();
  }

  void FinishRings() {  // :191-201
    ring_offsets.assign(num_rings_ + 1, 0);
    cloud_rings.clear();
    intensity_rings.clear();
    for (int r = 0; r < num_rings_; ++r) {
      cloud_rings.insert(cloud_rings.end(), laser_scans[r].begin(), laser_scans[r].end());
      intensity_rings.insert(intensity_rings.end(), intensity_scans[r].begin(), intensity_scans[r].end());
      ring_offsets[r + 1] = int(cloud_rings.size());
    }
    curvature.assign(cloud_rings.size(), 0.f);
    mask.assign(cloud_rings.size(), 0);
  }

  // math_utils.h:85-103
  static float SqDiff(const P4 &a, const P4 &b) {
    float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return dx * dx + dy * dy + dz * dz;
  }
  static float SqDiffW(const P4 &a, const P4 &b, float wb) {
    float dx = a.x - b.x * wb, dy = a.y - b.y * wb, dz = a.z - b.z * wb;
    return dx * dx + dy * dy + dz * dz;
  }
  static float Dist(const P4 &p) { return std::sqrt(p.x * p.x + p.y * p.y + p.z * p.z); }
  static float SqDist(const P4 &p) { return p.x * p.x + p.y * p.y + p.z * p.z; }

  void PrepareRing(const Cloud &scan, int *m) {
    const int nc = config_.num_curvature_regions;
    const size_t n = scan.size();
    for (size_t i = nc; i < n - nc; ++i) {
      const P4 &pp = scan[i - 1], &pc = scan[i], &pn = scan[i + 1];
      float diff_next2 = SqDiff(pc, pn);
      if (diff_next2 > 0.1) {  // float vs double literal
        float depth = Dist(pc), depth_next = Dist(pn);
        if (depth > depth_next) {
          float wd = std::sqrt(SqDiffW(pn, pc, depth_next / depth)) / depth_next;
          if (wd < 0.1) {
            for (int k = 0; k <= nc; ++k) m[i - nc + k] = 1;  // fill_n(&mask[i-5], 6, 1) :564
            continue;
          }
        } else {
          float wd = std::sqrt(SqDiffW(pc, pn, depth / depth_next)) / depth;
          if (wd < 0.1) {
            // fill_n(&mask[i+1], 6, 1) :570 — the reference writes one past the end when i = n-6;
            // that heap byte is never read, so the restatement clamps.
            for (int k = 0; k <= nc; ++k) if (i + 1 + k < n) m[i + 1 + k] = 1;
            continue;
          }
        }
      }
      float diff_prev2 = SqDiff(pc, pp);
      float dis2 = SqDist(pc);
      if (diff_next2 > 0.0002 * dis2 && diff_prev2 > 0.0002 * dis2) m[i] = 1;  // double * float
    }
  }

  void MaskPicked(const Cloud &scan, int *m, size_t idx) {
    const int nc = config_.num_curvature_regions;
    m[idx] = 1;
    for (int i = 1; i <= nc; ++i) {
      if (SqDiff(scan[idx + i], scan[idx + i - 1]) > 0.05) break;
      m[idx + i] = 1;
    }
    for (int i = 1; i <= nc; ++i) {
      if (SqDiff(scan[idx - i], scan[idx - i + 1]) > 0.05) break;
      m[idx - i] = 1;
    }
  }

  void ExtractFeaturePoints() {
    const int nc = config_.num_curvature_regions, ns = config_.num_scan_subregions;
    for (int r = 0; r < num_rings_; ++r) {
      // :657-662 with scan_ranges = (start, end inclusive)
      size_t start_idx = size_t(ring_offsets[r]);
      size_t cloud_size_after = size_t(ring_offsets[r + 1]);
      size_t end_idx = cloud_size_after > 0 ? cloud_size_after - 1 : 0;
      if (end_idx <= start_idx + 2 * nc) continue;
      const Cloud &scan = laser_scans[r];
      size_t scan_size = scan.size();
      int *m = mask.data() + ring_offsets[r];
      float *curv = curvature.data() + ring_offsets[r];
      PrepareRing(scan, m);
      Cloud ring_less_flat;
      for (int j = 0; j < ns; ++j) {
        size_t sp = (size_t(nc) * (ns - j) + (scan_size - nc) * j) / ns;
        size_t ep = (size_t(nc) * (ns - 1 - j) + (scan_size - nc) * (j + 1)) / ns - 1;
        if (ep <= sp) continue;
        size_t region_size = ep - sp + 1;
        // PrepareSubregion :587-622
        std::vector<std::pair<float, size_t>> pairs(region_size);
        std::vector<int> labels(region_size, 0);
        int npn = 2 * nc;
        for (size_t i = sp, k = 0; i <= ep; ++i, ++k) {
          float dx = -npn * scan[i].x, dy = -npn * scan[i].y, dz = -npn * scan[i].z;
          for (int q = 1; q <= nc; ++q) {
            dx += scan[i + q].x + scan[i - q].x;
            dy += scan[i + q].y + scan[i - q].y;
            dz += scan[i + q].z + scan[i - q].z;
          }
          float c = dx * dx + dy * dy + dz * dz;
          pairs[k] = {c, i};
          curv[i] = c;
        }
        std::sort(pairs.begin(), pairs.end());
        int num_largest = 0;
        for (size_t k = region_size; k > 0 && num_largest < config_.max_corner_less_sharp;) {
          const auto &ci = pairs[--k];
          float c = ci.first; size_t idx = ci.second;
          if (m[idx] == 0 && c > config_.surf_curv_th) {
            ++num_largest;
            if (num_largest <= config_.max_corner_sharp) {
              labels[idx - sp] = 2;
              sharp.push_back(scan[idx]); pick_ring[1].push_back(r); pick_idx[1].push_back(int(idx));
            } else {
              labels[idx - sp] = 1;
            }
            less_sharp.push_back(scan[idx]); pick_ring[2].push_back(r); pick_idx[2].push_back(int(idx));
            MaskPicked(scan, m, idx);
          }
        }
        int num_smallest = 0;
        for (size_t k = 0; k < region_size && num_smallest < config_.max_surf_flat; ++k) {
          const auto &ci = pairs[k];
          float c = ci.first; size_t idx = ci.second;
          if (m[idx] == 0 && c < config_.surf_curv_th) {
            ++num_smallest;
            labels[idx - sp] = -1;
            flat.push_back(scan[idx]); pick_ring[3].push_back(r); pick_idx[3].push_back(int(idx));
            MaskPicked(scan, m, idx);
          }
        }
        for (size_t k = 0; k < region_size; ++k)
          if (labels[k] <= 0) ring_less_flat.push_back(scan[sp + k]);
      }
      if (ring_less_flat.empty()) continue;
      Cloud ds;
      VoxelGrid(ring_less_flat, config_.less_flat_filter_size, ds);
      less_flat.insert(less_flat.end(), ds.begin(), ds.end());
    }
    // :755-778 rel-time recompute on the averaged points
    for (P4 &p : less_flat) {
      float azi_rad = float(2 * M_PI - std::atan2(p.y, p.x));
      if (azi_rad >= 2 * M_PI) azi_rad = float(azi_rad - 2 * M_PI);
      float azi_rad_rel = azi_rad - start_ori_;
      if (azi_rad_rel < 0) azi_rad_rel = float(azi_rad_rel + 2 * M_PI);
      float rel_time = float(config_.scan_period * azi_rad_rel / (2 * M_PI));
      p.i = int(p.i) + rel_time;
    }
  }
};

}  // namespace orc
