// oracle/odometry.h — TEST INFRASTRUCTURE ONLY (CPU oracle).  Not part of the product.
//
// CPU restatement of lio::PointOdometry (LOAM scan-to-scan step, BASELINE.json configs[1]):
//   src/point_processor/PointOdometry.cc:237-260 TransformToStart, :262-292 TransformToEnd,
//   :294-683 Process (correspondences every 5th iteration :344-380/:443-488, edge coefficients :391-435,
//   plane coefficients :497-531, 6x6 Gauss-Newton :539-580, degeneracy mask :584-615 (threshold 10, A.6),
//   update :617-640, abort test :642-650, accumulation :654-663, cloud swap :667-676).
// Weights / damping follow SURVEY.md A.7-A.8.  K=1 nearest neighbour = exact (B.2), ties -> lower index.
// Pinned (round 3) against the reference's own PointOdometry.cc compiled where it lies (oracle/ref_odometry.cc, oracle/ref_shim,
// `make ref`): tests/golden/ref_odometry_digests.json, tests/test_ref_odometry_digests.py — transforms, TransformToEnd clouds and
// /compact_data messages bit for bit (kd-tree = exact search; QR / eigen-solver forwarded to liomath.h, so not pinned by it).
#pragma once
#include "cloud.h"
#include "liomath.h"

namespace orc {

struct PointOdometry {
  float scan_period_, time_factor_;
  int io_ratio_;
  size_t num_max_iterations_;
  bool no_deskew_ = false, enable_odom_ = true, system_inited_ = false;
  double delta_r_abort_ = 0.1, delta_t_abort_ = 0.1;
  Twist<float> transform_es_, transform_sum_;
  Cloud last_corner_, last_surf_;
  KdTree kd_corner_, kd_surf_;
  bool trees_valid_ = false;
  size_t frame_count_ = 0;
  int iterations_done_ = 0, last_num_sel_ = 0;
  int last_kz_ = 0;                       // leading update components masked by iteration 0's degeneracy test (:584-615)
  std::vector<Twist<float>> es_trace_;    // transform_es_ at the end of every iteration of the last Process

  PointOdometry(float scan_period, int io_ratio, size_t max_iter, bool no_deskew)
      : scan_period_(scan_period), time_factor_(1 / scan_period), io_ratio_(io_ratio), num_max_iterations_(max_iter), no_deskew_(no_deskew) {}

  bool TransformToStart(const P4 &pi, P4 &po) const {
    float s = time_factor_ * (pi.i - int(pi.i));
    if (no_deskew_) s = 0;
    if (s < 0 || s > 1.001) { po = pi; return false; }
    po.x = pi.x - s * transform_es_.pos.x;
    po.y = pi.y - s * transform_es_.pos.y;
    po.z = pi.z - s * transform_es_.pos.z;
    po.i = pi.i;
    Q<float> q_id;
    Q<float> q_s = q_id.slerp(s, transform_es_.rot);
    V3<float> v = q_s.conjugate() * V3<float>(po.x, po.y, po.z);
    po.x = v.x; po.y = v.y; po.z = v.z;
    return true;
  }
  void TransformToEnd(Cloud &cloud) const {
    for (P4 &p : cloud) {
      float s = time_factor_ * (p.i - int(p.i));
      if (no_deskew_) s = 0;
      p.x -= s * transform_es_.pos.x; p.y -= s * transform_es_.pos.y; p.z -= s * transform_es_.pos.z;
      p.i = float(int(p.i));
      Q<float> q_id;
      Q<float> q_s = q_id.slerp(s, transform_es_.rot);
      V3<float> v = q_s.conjugate() * V3<float>(p.x, p.y, p.z);
      v = transform_es_.rot * v;
      p.x = v.x + transform_es_.pos.x; p.y = v.y + transform_es_.pos.y; p.z = v.z + transform_es_.pos.z;
    }
  }
  static float SqDiff(const P4 &a, const P4 &b) {
    float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return dx * dx + dy * dy + dz * dz;
  }

  void Process(const Cloud &sharp, Cloud less_sharp, const Cloud &flat, Cloud less_flat) {
    iterations_done_ = 0; last_num_sel_ = 0; last_kz_ = 0; es_trace_.clear();
    if (!system_inited_) {
      last_corner_.swap(less_sharp); last_surf_.swap(less_flat);
      kd_corner_.Build(last_corner_); kd_surf_.Build(last_surf_);
      trees_valid_ = true;
      system_inited_ = true;
      return;
    }
    bool is_degenerate = false;
    ++frame_count_;
    const size_t last_corner_size = last_corner_.size(), last_surf_size = last_surf_.size();
    if (enable_odom_) {
      if (last_corner_size > 10 && last_surf_size > 100) {
        const size_t nc = sharp.size(), ns = flat.size();
        std::vector<int> ic1(nc, -1), ic2(nc, -1), is1(ns, -1), is2(ns, -1), is3(ns, -1);
        int kz = 0;
        for (size_t iter = 0; iter < num_max_iterations_; ++iter) {
          ++iterations_done_;
          std::vector<P4> ori, coef;
          P4 sel;
          for (size_t i = 0; i < nc; ++i) {
            TransformToStart(sharp[i], sel);
            if (iter % 5 == 0) {
              int idx; float sq;
              int closest = -1, second = -1;
              if (kd_corner_.Search(sel, 1, &idx, &sq) == 1 && sq < 25) {
                closest = idx;
                int cs = int(last_corner_[closest].i);
                float d2, best = 25;
                for (int j = closest + 1; j < int(last_corner_size); ++j) {
                  if (int(last_corner_[j].i) > cs + 2.5) break;
                  d2 = SqDiff(last_corner_[j], sel);
                  if (int(last_corner_[j].i) > cs && d2 < best) { best = d2; second = j; }
                }
                for (int j = closest - 1; j >= 0; --j) {
                  if (int(last_corner_[j].i) < cs - 2.5) break;
                  d2 = SqDiff(last_corner_[j], sel);
                  if (int(last_corner_[j].i) < cs && d2 < best) { best = d2; second = j; }
                }
              }
              ic1[i] = closest; ic2[i] = second;
            }
            if (ic2[i] >= 0) {
              const P4 &t1 = last_corner_[ic1[i]], &t2 = last_corner_[ic2[i]];
              float x0 = sel.x, y0 = sel.y, z0 = sel.z, x1 = t1.x, y1 = t1.y, z1 = t1.z, x2 = t2.x, y2 = t2.y, z2 = t2.z;
              float a012 = std::sqrt(((x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1)) * ((x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1)) +
                                     ((x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1)) * ((x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1)) +
                                     ((y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1)) * ((y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1)));
              float l12 = std::sqrt((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2));
              float la = ((y1 - y2) * ((x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1)) + (z1 - z2) * ((x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1))) / a012 / l12;
              float lb = -((x1 - x2) * ((x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1)) - (z1 - z2) * ((y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1))) / a012 / l12;
              float lc = -((x1 - x2) * ((x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1)) + (y1 - y2) * ((y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1))) / a012 / l12;
              float ld2 = a012 / l12;
              float s = 1;
              if (iter >= 5) s = 1 - 1.8f * std::fabs(ld2);
              if (s > 0.1 && ld2 != 0) { ori.push_back(sharp[i]); coef.push_back({s * la, s * lb, s * lc, s * ld2}); }
            }
          }
          for (size_t i = 0; i < ns; ++i) {
            TransformToStart(flat[i], sel);
            if (iter % 5 == 0) {
              int idx; float sq;
              int closest = -1, second = -1, third = -1;
              if (kd_surf_.Search(sel, 1, &idx, &sq) == 1 && sq < 25) {
                closest = idx;
                int cs = int(last_surf_[closest].i);
                float d2, b2 = 25, b3 = 25;
                for (int j = closest + 1; j < int(last_surf_size); ++j) {
                  if (int(last_surf_[j].i) > cs + 2.5) break;
                  d2 = SqDiff(last_surf_[j], sel);
                  if (int(last_surf_[j].i) <= cs) { if (d2 < b2) { b2 = d2; second = j; } }
                  else { if (d2 < b3) { b3 = d2; third = j; } }
                }
                for (int j = closest - 1; j >= 0; --j) {
                  if (int(last_surf_[j].i) < cs - 2.5) break;
                  d2 = SqDiff(last_surf_[j], sel);
                  if (int(last_surf_[j].i) >= cs) { if (d2 < b2) { b2 = d2; second = j; } }
                  else { if (d2 < b3) { b3 = d2; third = j; } }
                }
              }
              is1[i] = closest; is2[i] = second; is3[i] = third;
            }
            if (is2[i] >= 0 && is3[i] >= 0) {
              const P4 &t1 = last_surf_[is1[i]], &t2 = last_surf_[is2[i]], &t3 = last_surf_[is3[i]];
              float pa = (t2.y - t1.y) * (t3.z - t1.z) - (t3.y - t1.y) * (t2.z - t1.z);
              float pb = (t2.z - t1.z) * (t3.x - t1.x) - (t3.z - t1.z) * (t2.x - t1.x);
              float pc = (t2.x - t1.x) * (t3.y - t1.y) - (t3.x - t1.x) * (t2.y - t1.y);
              float pd = -(pa * t1.x + pb * t1.y + pc * t1.z);
              float ps = std::sqrt(pa * pa + pb * pb + pc * pc);
              pa /= ps; pb /= ps; pc /= ps; pd /= ps;
              float pd2 = pa * sel.x + pb * sel.y + pc * sel.z + pd;
              float s = 1;
              if (iter >= 5) s = 1 - 1.8f * std::fabs(pd2) / std::sqrt(std::sqrt(sel.x * sel.x + sel.y * sel.y + sel.z * sel.z));
              if (s > 0.1 && pd2 != 0) { ori.push_back(flat[i]); coef.push_back({s * pa, s * pb, s * pc, s * pd2}); }
            }
          }
          const int nsel = int(ori.size());
          last_num_sel_ = nsel;
          if (nsel < 10) { es_trace_.push_back(transform_es_); continue; }
          float AtA[36] = {0}, AtB[6] = {0};
          Q<float> R0 = transform_es_.rot.normalized();  // SO3 ctor normalises
          M3<float> Rt = transform_es_.rot.toRotationMatrix().transpose();
          for (int i = 0; i < nsel; ++i) {
            V3<float> p(ori[i].x, ori[i].y, ori[i].z), w(coef[i].x, coef[i].y, coef[i].z);
            V3<float> pmt = p - transform_es_.pos;
            V3<float> c = transform_es_.rot.conjugate() * pmt;
            M3<float> S = Skew(c);
            float a[6];
            a[0] = w.x * S(0, 0) + w.y * S(1, 0) + w.z * S(2, 0);
            a[1] = w.x * S(0, 1) + w.y * S(1, 1) + w.z * S(2, 1);
            a[2] = w.x * S(0, 2) + w.y * S(1, 2) + w.z * S(2, 2);
            a[3] = -(w.x * Rt(0, 0) + w.y * Rt(1, 0) + w.z * Rt(2, 0));
            a[4] = -(w.x * Rt(0, 1) + w.y * Rt(1, 1) + w.z * Rt(2, 1));
            a[5] = -(w.x * Rt(0, 2) + w.y * Rt(1, 2) + w.z * Rt(2, 2));
            float bb = float(-0.1 * coef[i].i);
            for (int r = 0; r < 6; ++r) { for (int cc = 0; cc < 6; ++cc) AtA[r * 6 + cc] += a[r] * a[cc]; AtB[r] += a[r] * bb; }
          }
          float Ac[36], Bc[6], X[6];
          std::memcpy(Ac, AtA, sizeof(Ac)); std::memcpy(Bc, AtB, sizeof(Bc));
          colpiv_qr_solve<float>(6, 6, Ac, Bc, X);
          if (iter == 0) {
            float E[6], V[36];
            sym_eigen<float>(6, AtA, E, V);
            is_degenerate = false; kz = 0;
            for (int i = 0; i < 6; ++i) { if (E[i] < 10.f) { ++kz; is_degenerate = true; } else break; }
            last_kz_ = kz;
          }
          if (is_degenerate) for (int i = 0; i < kz; ++i) X[i] = 0.f;  // matP = diag(0..0,1..1) (A.6)
          transform_es_.pos.x += X[3]; transform_es_.pos.y += X[4]; transform_es_.pos.z += X[5];
          transform_es_.rot = transform_es_.rot * DeltaQ(V3<float>(X[0], X[1], X[2]));
          if (!std::isfinite(transform_es_.pos.x)) transform_es_.pos.x = 0;
          if (!std::isfinite(transform_es_.pos.y)) transform_es_.pos.y = 0;
          if (!std::isfinite(transform_es_.pos.z)) transform_es_.pos.z = 0;
          float delta_r = RadToDeg(R0.angularDistance(transform_es_.rot));
          float delta_t = float(std::sqrt(std::pow(X[3] * 100, 2) + std::pow(X[4] * 100, 2) + std::pow(X[5] * 100, 2)));
          es_trace_.push_back(transform_es_);
          if (delta_r < delta_r_abort_ && delta_t < delta_t_abort_) break;
        }
      }
      Twist<float> se = transform_es_.inverse();
      transform_sum_ = transform_sum_ * se;
      TransformToEnd(less_sharp);
      TransformToEnd(less_flat);
      transform_es_.rot.normalize();
    }
    last_corner_.swap(less_sharp); last_surf_.swap(less_flat);
    if (last_corner_.size() > 10 && last_surf_.size() > 100) { kd_corner_.Build(last_corner_); kd_surf_.Build(last_surf_); trees_valid_ = true; }
  }
};

}  // namespace orc
