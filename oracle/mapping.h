// oracle/mapping.h — TEST INFRASTRUCTURE ONLY (CPU oracle).  Not part of the product.
//
// Restates the scan-to-map stage (LOAM mapping) of the reference:
//   src/point_processor/PointMapping.cc
//     :303-323   PointAssociateToMap / PointAssociateTobeMapped
//     :325-753   OptimizeTransformTobeMapped (corner: 5-NN + 3x3 eigen line fit; surf: 5-NN + plane fit)
//     :755-763   TransformAssociateToMap / TransformUpdate
//     :765-1110  Process (cube-map shift, FOV cube selection, stack round trip + VoxelGrid)
//     :1112-1208 UpdateMapDatabase
//   include/point_processor/PointMapping.h:148-160 (ToIndex / FromIndex), :243-249 (score map, thresholds)
// Parity UNPINNED vs PCL/Eigen (absent here, see cloud.h / liomath.h headers).
// Pinned (round 3) against the reference's own PointMapping.cc compiled where it lies (oracle/ref_mapping.cc, oracle/ref_shim,
// `make ref`): tests/golden/ref_mapping_digests.json, tests/test_ref_mapping_digests.py — transforms, stacks, from-map clouds, window
// state and cube contents bit for bit (VoxelGrid / QR / eigen-solver forwarded to this oracle's own restatements, so not pinned by it).
#pragma once
#include <array>
#include <cstdio>
#include <cstdlib>

#include "cloud.h"
#include "liomath.h"

namespace orc {

typedef Twist<float> Transformf;

struct MappingConfig {
  float corner_filter_size = 0.2f, surf_filter_size = 0.4f;  // PointMapping.cc:122-123; Estimator.cc:189-191
  float min_match_sq_dis = 1.0f, min_plane_dis = 0.2f;       // PointMapping.h:245-246
  int num_max_iterations = 10;                               // PointMapping.h:171
  // MapBuilder (src/map_builder/MapBuilder.cc): the same stage driven as ProcessMap / OptimizeMap.  Pinned (round 3) against the
  // reference's own MapBuilder.cc compiled where it lies (oracle/ref_mapbuilder.cc, tests/test_ref_mapbuilder_digests.py)
  bool map_builder = false;
  bool enable_4d = true;   // MapBuilder.h:66
  int skip_count = 2;      // MapBuilder.h:67
};

struct ScorePointCoeff { float score; P4 point; P4 coeff; };

struct PointMapping {
  MappingConfig cfg;
  static constexpr int L = 21, Wd = 21, H = 11;  // laser_cloud_length_/width_/height_ (PointMapping.cc:79-81)
  int cen[3] = {10, 10, 5};                       // laser_cloud_cen_length_/width_/height_ (:76-78)
  std::vector<Cloud> corner_array, surf_array;
  Transformf transform_sum, transform_tobe_mapped, transform_bef_mapped, transform_aft_mapped;
  bool imu_inited = false;
  Cloud corner_stack_ds, surf_stack_ds, corner_from_map, surf_from_map;
  std::vector<size_t> valid_idx, surround_idx;
  std::vector<ScorePointCoeff> score_point_coeff;  // descending score, insertion order among equals
  P4 point_on_z_axis{0, 0, 0, 0};
  float matP[36];
  int last_iterations = 0, last_selected = 0;
  bool last_degenerate = false;
  int last_kz = 0;   // leading update components masked (PointMapping.cc:650-680, MapBuilder.cc:930-960)
  bool system_init = false;  // MapBuilder.h:65
  int odom_count = 0;        // MapBuilder.h:69

  explicit PointMapping(const MappingConfig &c = MappingConfig()) : cfg(c) {
    corner_array.resize(size_t(L) * Wd * H);
    surf_array.resize(size_t(L) * Wd * H);
    for (int i = 0; i < 36; ++i) matP[i] = (i % 7 == 0) ? 1.f : 0.f;
  }

  static size_t ToIndex(int i, int j, int k) { return size_t(i) + size_t(L) * j + size_t(L) * Wd * k; }
  static void FromIndex(size_t index, int &i, int &j, int &k) {
    int residual = int(index % (size_t(L) * Wd));
    k = int(index / (size_t(L) * Wd));
    j = residual / L;
    i = residual % L;
  }

  static P4 ToMap(const P4 &pi, const Transformf &T) {  // :303-314
    V3<float> r = T.rot * V3<float>(pi.x, pi.y, pi.z);
    return P4{r.x + T.pos.x, r.y + T.pos.y, r.z + T.pos.z, pi.i};
  }
  static P4 TobeMapped(const P4 &pi, const Transformf &T) {  // :316-323
    V3<float> v(pi.x - T.pos.x, pi.y - T.pos.y, pi.z - T.pos.z);
    V3<float> r = T.rot.conjugate() * v;
    return P4{r.x, r.y, r.z, pi.i};
  }
  static float SqDiff(const P4 &a, const P4 &b) {
    float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return dx * dx + dy * dy + dz * dz;
  }
  bool InFov(const P4 &pos, const P4 &p) const {  // law-of-cosines test, +-60 deg about the sensor z axis
    float s1 = SqDiff(pos, p), s2 = SqDiff(point_on_z_axis, p);
    float check1 = 100.0f + s1 - s2 - 10.0f * std::sqrt(3.0f) * std::sqrt(s1);
    float check2 = 100.0f + s1 - s2 + 10.0f * std::sqrt(3.0f) * std::sqrt(s1);
    return check1 < 0 && check2 > 0;
  }

  void TransformAssociateToMap() {  // :755-758; Twist * Affine converts through Twist(Affine)
    Transformf sumT = Transformf::FromAffine(transform_sum.linear(), transform_sum.pos);
    Transformf incre = transform_bef_mapped.inverse() * sumT;
    transform_tobe_mapped = transform_tobe_mapped * incre;
  }
  void TransformUpdate() { transform_bef_mapped = transform_sum; transform_aft_mapped = transform_tobe_mapped; }
  // MapBuilder::Transform4DAssociateToMap (MapBuilder.cc:55-75): position from the 6-DoF increment, rotation = the
  // odometry's with only the yaw of the increment applied
  void Transform4DAssociateToMap() {
    Transformf sumT = Transformf::FromAffine(transform_sum.linear(), transform_sum.pos);
    Transformf incre = transform_bef_mapped.inverse() * sumT;
    Transformf full = transform_tobe_mapped * incre;
    V3<double> r0 = R2ypr(full.rot.normalized().toRotationMatrix().cast<double>());
    V3<double> r00 = R2ypr(transform_sum.rot.normalized().toRotationMatrix().cast<double>());
    float y = float(double(float(r0.x - r00.x)) / 180.0 * M_PI);  // ypr2R<Vector3f>: Rz only, pitch = roll = 0
    M3<float> Rz = M3<float>::Identity();
    Rz(0, 0) = std::cos(y); Rz(0, 1) = -std::sin(y); Rz(1, 0) = std::sin(y); Rz(1, 1) = std::cos(y);
    transform_tobe_mapped.pos = full.pos;
    transform_tobe_mapped.rot = Q<float>::FromMatrix(Rz * transform_sum.rot.normalized().toRotationMatrix());
  }

  static int CubeCoord(float v, int c) {
    int r = int((double(v) + 25.0) / 50.0) + c;
    if (double(v) + 25.0 < 0) --r;
    return r;
  }

  // shift the 21x21x11 cube window so the sensor cube stays >= 3 cubes from the border (:820-925)
  void ShiftAxis(int axis, int dir) {
    const int dims[3] = {L, Wd, H};
    auto at = [&](int a, int b, int c) {  // index with `axis` coordinate a, the other two b, c
      int ijk[3];
      ijk[axis] = a; ijk[(axis + 1) % 3] = b; ijk[(axis + 2) % 3] = c;
      return ToIndex(ijk[0], ijk[1], ijk[2]);
    };
    int n = dims[axis], nb = dims[(axis + 1) % 3], nc = dims[(axis + 2) % 3];
    for (int b = 0; b < nb; ++b)
      for (int c = 0; c < nc; ++c) {
        if (dir > 0) {  // contents move towards larger index; slot 0 cleared
          for (int a = n - 1; a >= 1; --a) { std::swap(corner_array[at(a, b, c)], corner_array[at(a - 1, b, c)]); std::swap(surf_array[at(a, b, c)], surf_array[at(a - 1, b, c)]); }
          corner_array[at(0, b, c)].clear(); surf_array[at(0, b, c)].clear();
        } else {
          for (int a = 0; a < n - 1; ++a) { std::swap(corner_array[at(a, b, c)], corner_array[at(a + 1, b, c)]); std::swap(surf_array[at(a, b, c)], surf_array[at(a + 1, b, c)]); }
          corner_array[at(n - 1, b, c)].clear(); surf_array[at(n - 1, b, c)].clear();
        }
      }
  }

  void Process(const Cloud &corner_last, const Cloud &surf_last, const Transformf &sum) {
    transform_sum = sum;
    if (cfg.map_builder) {  // MapBuilder::ProcessMap (MapBuilder.cc:220-247)
      if (!system_init) { system_init = true; transform_bef_mapped = transform_tobe_mapped = transform_aft_mapped = transform_sum; }
      if (cfg.enable_4d) Transform4DAssociateToMap(); else TransformAssociateToMap();
    } else if (!imu_inited) {
      TransformAssociateToMap();
    }
    Cloud corner_stack, surf_stack;
    for (const P4 &p : corner_last) corner_stack.push_back(ToMap(p, transform_tobe_mapped));
    for (const P4 &p : surf_last) surf_stack.push_back(ToMap(p, transform_tobe_mapped));
    point_on_z_axis = ToMap(P4{0.f, 0.f, 10.f, 0.f}, transform_tobe_mapped);

    int cc[3];
    const float posv[3] = {transform_tobe_mapped.pos.x, transform_tobe_mapped.pos.y, transform_tobe_mapped.pos.z};
    for (int d = 0; d < 3; ++d) cc[d] = CubeCoord(posv[d], cen[d]);
    const int dims[3] = {L, Wd, H};
    for (int d = 0; d < 3; ++d) {
      while (cc[d] < 3) { ShiftAxis(d, +1); ++cc[d]; ++cen[d]; }
      while (cc[d] >= dims[d] - 3) { ShiftAxis(d, -1); --cc[d]; --cen[d]; }
    }

    valid_idx.clear(); surround_idx.clear();
    P4 tpos{posv[0], posv[1], posv[2], 0};
    for (int i = cc[0] - 2; i <= cc[0] + 2; ++i)
      for (int j = cc[1] - 2; j <= cc[1] + 2; ++j)
        for (int k = cc[2] - 2; k <= cc[2] + 2; ++k) {
          if (!(i >= 0 && i < L && j >= 0 && j < Wd && k >= 0 && k < H)) continue;
          float cx = 50.0f * (i - cen[0]), cy = 50.0f * (j - cen[1]), cz = 50.0f * (k - cen[2]);
          bool fov = false;
          for (int ii = -1; ii <= 1; ii += 2)
            for (int jj = -1; jj <= 1; jj += 2)
              for (int kk = -1; kk <= 1; kk += 2) {
                P4 corner{cx + 25.0f * ii, cy + 25.0f * jj, cz + 25.0f * kk, 0};
                if (InFov(tpos, corner)) fov = true;
              }
          size_t idx = ToIndex(i, j, k);
          if (fov) valid_idx.push_back(idx);
          surround_idx.push_back(idx);
        }

    corner_from_map.clear(); surf_from_map.clear();
    for (size_t idx : valid_idx) {
      corner_from_map.insert(corner_from_map.end(), corner_array[idx].begin(), corner_array[idx].end());
      surf_from_map.insert(surf_from_map.end(), surf_array[idx].begin(), surf_array[idx].end());
    }
    for (P4 &p : corner_stack) p = TobeMapped(p, transform_tobe_mapped);
    for (P4 &p : surf_stack) p = TobeMapped(p, transform_tobe_mapped);
    VoxelGrid(corner_stack, cfg.corner_filter_size, corner_stack_ds);
    VoxelGrid(surf_stack, cfg.surf_filter_size, surf_stack_ds);

    if (cfg.map_builder) {  // MapBuilder.cc:527-558
      if (odom_count % cfg.skip_count == 0) OptimizeTransformTobeMapped(cfg.enable_4d);
      else { last_iterations = 0; last_selected = 0; TransformUpdate(); }
      ++odom_count;
      UpdateMapDatabase(corner_stack_ds, surf_stack_ds, valid_idx, transform_tobe_mapped, cen);
      return;
    }
    OptimizeTransformTobeMapped(false);

    if (!imu_inited) UpdateMapDatabase(corner_stack_ds, surf_stack_ds, valid_idx, transform_tobe_mapped, cen);
  }

  // four_dof = MapBuilder::OptimizeMap (MapBuilder.cc:624-1014): no sign flip of the plane coefficients, rotation
  // Jacobian in the map frame weighted diag(5e-3, 5e-3, 1), left-multiplied rotation update, no score list
  void OptimizeTransformTobeMapped(bool four_dof) {
    last_iterations = 0; last_selected = 0; last_degenerate = false; last_kz = 0;
    if (corner_from_map.size() <= 10 || surf_from_map.size() <= 100) return;
    KdTree tree_corner, tree_surf;
    tree_corner.Build(corner_from_map);
    tree_surf.Build(surf_from_map);
    bool is_degenerate = false;
    for (int i = 0; i < 36; ++i) matP[i] = (i % 7 == 0) ? 1.f : 0.f;
    Transformf &T = transform_tobe_mapped;
    struct Sel { P4 ori, coeff; };
    std::vector<Sel> sel;
    std::vector<ScorePointCoeff> spc;  // (unused score, p_ori, abs_coeff) of the surf branch
    std::vector<P4> spc_coeff;
    score_point_coeff.clear();
    for (int iter = 0; iter < cfg.num_max_iterations; ++iter) {
      ++last_iterations;
      sel.clear(); spc.clear(); spc_coeff.clear();
      P4 tpos{T.pos.x, T.pos.y, T.pos.z, 0};
      for (const P4 &po : corner_stack_ds) {
        P4 ps = ToMap(po, T);
        int idx[5]; float sq[5];
        if (tree_corner.Search(ps, 5, idx, sq) < 5) continue;
        if (!(sq[4] < cfg.min_match_sq_dis)) continue;
        V3<float> vc(0, 0, 0);
        for (int j = 0; j < 5; ++j) { const P4 &m = corner_from_map[idx[j]]; vc.x += m.x; vc.y += m.y; vc.z += m.z; }
        vc.x /= 5.0f; vc.y /= 5.0f; vc.z /= 5.0f;
        float a00 = 0, a10 = 0, a20 = 0, a11 = 0, a21 = 0, a22 = 0;
        for (int j = 0; j < 5; ++j) {
          const P4 &m = corner_from_map[idx[j]];
          float ax = m.x - vc.x, ay = m.y - vc.y, az = m.z - vc.z;
          a00 += ax * ax; a10 += ax * ay; a20 += ax * az; a11 += ay * ay; a21 += ay * az; a22 += az * az;
        }
        a00 /= 5.0f; a10 /= 5.0f; a20 /= 5.0f; a11 /= 5.0f; a21 /= 5.0f; a22 /= 5.0f;
        float A1[9] = {a00, a10, a20, a10, a11, a21, a20, a21, a22};  // solver reads the lower triangle
        float D1[3], V1[9];
        sym_eigen<float>(3, A1, D1, V1);
        if (!(D1[2] > 3 * D1[1])) continue;
        float x0 = ps.x, y0 = ps.y, z0 = ps.z;
        float x1 = float(double(vc.x) + 0.1 * double(V1[0 * 3 + 2])), y1 = float(double(vc.y) + 0.1 * double(V1[1 * 3 + 2])),
              z1 = float(double(vc.z) + 0.1 * double(V1[2 * 3 + 2]));
        float x2 = float(double(vc.x) - 0.1 * double(V1[0 * 3 + 2])), y2 = float(double(vc.y) - 0.1 * double(V1[1 * 3 + 2])),
              z2 = float(double(vc.z) - 0.1 * double(V1[2 * 3 + 2]));
        V3<float> X0(x0, y0, z0), X1(x1, y1, z1), X2(x2, y2, z2);
        V3<float> a012_vec = (X0 - X1).cross(X0 - X2);
        V3<float> ntp = (X1 - X2).cross(a012_vec).normalized();
        float a012 = a012_vec.norm(), l12 = (X1 - X2).norm();
        float la = ntp.x, lb = ntp.y, lc = ntp.z, ld2 = a012 / l12;
        float s = 1 - 0.9f * std::fabs(ld2);
        P4 coeff{s * la, s * lb, s * lc, s * ld2};
        if (s > 0.1 && InFov(tpos, ps)) sel.push_back({po, coeff});
      }
      for (const P4 &po : surf_stack_ds) {
        P4 ps = ToMap(po, T);
        int idx[5]; float sq[5];
        if (tree_surf.Search(ps, 5, idx, sq) < 5) continue;
        if (!(sq[4] < cfg.min_match_sq_dis)) continue;
        float A[15], B[5] = {-1, -1, -1, -1, -1}, X[3];
        for (int j = 0; j < 5; ++j) { A[j * 3 + 0] = surf_from_map[idx[j]].x; A[j * 3 + 1] = surf_from_map[idx[j]].y; A[j * 3 + 2] = surf_from_map[idx[j]].z; }
        colpiv_qr_solve<float>(5, 3, A, B, X);
        float pa = X[0], pb = X[1], pc = X[2], pd = 1;
        float pn = std::sqrt(pa * pa + pb * pb + pc * pc);
        pa /= pn; pb /= pn; pc /= pn; pd /= pn;
        bool plane_valid = true;
        for (int j = 0; j < 5; ++j) {
          const P4 &m = surf_from_map[idx[j]];
          if (std::fabs(pa * m.x + pb * m.y + pc * m.z + pd) > cfg.min_plane_dis) { plane_valid = false; break; }
        }
        if (!plane_valid) continue;
        float pd2 = pa * ps.x + pb * ps.y + pc * ps.z + pd;
        float s = 1 - 0.9f * std::fabs(pd2) / std::sqrt(std::sqrt(ps.x * ps.x + ps.y * ps.y + ps.z * ps.z));
        P4 coeff, abs_coeff;
        if (pd2 > 0 || four_dof) { coeff = P4{s * pa, s * pb, s * pc, s * pd2}; abs_coeff = P4{pa, pb, pc, pd}; }
        else { coeff = P4{-s * pa, -s * pb, -s * pc, -s * pd2}; abs_coeff = P4{-pa, -pb, -pc, -pd}; }
        if (s > 0.1 && InFov(tpos, ps)) {
          sel.push_back({po, coeff});
          spc.push_back({0.f, po, abs_coeff});
          spc_coeff.push_back(coeff);
        }
      }
      last_selected = int(sel.size());
      if (sel.size() < 50) continue;
      float AtA[36] = {0}, AtB[6] = {0};
      Q<float> R0 = T.rot.normalized();
      M3<float> Rm = T.rot.toRotationMatrix();
      M3<float> Rinv = T.rot.inverse().toRotationMatrix();
      for (const Sel &f : sel) {
        V3<float> p(f.ori.x, f.ori.y, f.ori.z), w(f.coeff.x, f.coeff.y, f.coeff.z);
        M3<float> RS = Rm * Skew(p);
        float a[6];
        a[0] = -(w.x * RS(0, 0) + w.y * RS(1, 0) + w.z * RS(2, 0));
        a[1] = -(w.x * RS(0, 1) + w.y * RS(1, 1) + w.z * RS(2, 1));
        a[2] = -(w.x * RS(0, 2) + w.y * RS(1, 2) + w.z * RS(2, 2));
        if (four_dof) {  // (-w^T R skew(p)) R^-1 diag(5e-3, 5e-3, 1)
          const float t0 = a[0], t1 = a[1], t2 = a[2];
          a[0] = (t0 * Rinv(0, 0) + t1 * Rinv(1, 0) + t2 * Rinv(2, 0)) * 5e-3f;
          a[1] = (t0 * Rinv(0, 1) + t1 * Rinv(1, 1) + t2 * Rinv(2, 1)) * 5e-3f;
          a[2] = (t0 * Rinv(0, 2) + t1 * Rinv(1, 2) + t2 * Rinv(2, 2)) * 1.f;
        }
        a[3] = w.x; a[4] = w.y; a[5] = w.z;
        float bb = -f.coeff.i;
        for (int r = 0; r < 6; ++r) { for (int c = 0; c < 6; ++c) AtA[r * 6 + c] += a[r] * a[c]; AtB[r] += a[r] * bb; }
      }
      float Ac[36], Bc[6], X[6];
      std::memcpy(Ac, AtA, sizeof(AtA)); std::memcpy(Bc, AtB, sizeof(AtB));
      colpiv_qr_solve<float>(6, 6, Ac, Bc, X);
      if (iter == 0) {
        float E[6], V[36];
        sym_eigen<float>(6, AtA, E, V);
        is_degenerate = false;
        for (int k = 0; k < 36; ++k) matP[k] = 0;
        int kz = 0;
        for (int i = 0; i < 6; ++i) { if (E[i] < 100.f) { ++kz; is_degenerate = true; } else break; }  // A.6
        for (int i = kz; i < 6; ++i) matP[i * 6 + i] = 1.f;
        last_kz = kz;
      }
      if (is_degenerate) {
        float X2[6];
        for (int i = 0; i < 6; ++i) { float s = 0; for (int j = 0; j < 6; ++j) s += matP[i * 6 + j] * X[j]; X2[i] = s; }
        std::memcpy(X, X2, sizeof(X));
      }
      last_degenerate = is_degenerate;
      T.pos.x += X[3]; T.pos.y += X[4]; T.pos.z += X[5];
      T.rot = four_dof ? DeltaQ(V3<float>(X[0], X[1], X[2])) * T.rot : T.rot * DeltaQ(V3<float>(X[0], X[1], X[2]));
      if (!std::isfinite(T.pos.x)) T.pos.x = 0;
      if (!std::isfinite(T.pos.y)) T.pos.y = 0;
      if (!std::isfinite(T.pos.z)) T.pos.z = 0;
      float delta_r = RadToDeg(R0.angularDistance(T.rot));
      float delta_t = std::sqrt(std::pow(X[3] * 100, 2) + std::pow(X[4] * 100, 2) + std::pow(X[5] * 100, 2));
      if (std::getenv("LIO_ORACLE_DEBUG")) std::fprintf(stderr, "[map] it %d nsel %zu X %g %g %g | %g %g %g pos %g %g %g\n", iter, sel.size(), X[0], X[1], X[2], X[3], X[4], X[5], T.pos.x, T.pos.y, T.pos.z);
      if (delta_r < 0.05 && delta_t < 0.05) break;
    }
    TransformUpdate();
    if (!four_dof && spc.size() >= 50) {
      for (size_t i = 0; i < spc.size(); ++i) {
        const P4 &c = spc_coeff[i];
        spc[i].score = std::sqrt(c.x * c.x + c.y * c.y + c.z * c.z);
      }
      score_point_coeff = spc;
      std::stable_sort(score_point_coeff.begin(), score_point_coeff.end(),
                       [](const ScorePointCoeff &a, const ScorePointCoeff &b) { return a.score > b.score; });
    }
  }

  void UpdateMapDatabase(const Cloud &corner_ds, const Cloud &surf_ds, const std::vector<size_t> &margin_valid_idx,
                         const Transformf &T, const int margin_cen[3]) {
    auto insert = [&](const Cloud &in, std::vector<Cloud> &arr) {
      for (const P4 &p : in) {
        P4 ps = ToMap(p, T);
        int ci = CubeCoord(ps.x, cen[0]), cj = CubeCoord(ps.y, cen[1]), ck = CubeCoord(ps.z, cen[2]);
        if (ci >= 0 && ci < L && cj >= 0 && cj < Wd && ck >= 0 && ck < H) arr[ToIndex(ci, cj, ck)].push_back(ps);
      }
    };
    insert(corner_ds, corner_array);
    insert(surf_ds, surf_array);
    for (size_t index : margin_valid_idx) {
      int li, lj, lk;
      FromIndex(index, li, lj, lk);
      float cx = 50.0f * (li - margin_cen[0]), cy = 50.0f * (lj - margin_cen[1]), cz = 50.0f * (lk - margin_cen[2]);
      int ci = CubeCoord(cx, cen[0]), cj = CubeCoord(cy, cen[1]), ck = CubeCoord(cz, cen[2]);
      if (!(ci >= 0 && ci < L && cj >= 0 && cj < Wd && ck >= 0 && ck < H)) continue;
      size_t idx = ToIndex(ci, cj, ck);
      Cloud tmp;
      VoxelGrid(corner_array[idx], cfg.corner_filter_size, tmp); corner_array[idx].swap(tmp);
      VoxelGrid(surf_array[idx], cfg.surf_filter_size, tmp); surf_array[idx].swap(tmp);
    }
  }
};

// One keyframe of the batched refinement (BASELINE.json configs[4]): the scan-to-map loop above run on caller-supplied
// from-map clouds and down-sampled stacks, from the initial pose T_in.  The keyframes of a batch are independent.
struct KeyframeRefinement { Transformf T; int iterations = 0, selected = 0, kz = 0; bool degenerate = false; };
inline KeyframeRefinement RefineKeyframe(const MappingConfig &cfg, const Cloud &corner_map, const Cloud &surf_map, const Cloud &corner_stack,
                                         const Cloud &surf_stack, const Transformf &T_in, bool four_dof) {
  PointMapping pm(cfg);
  pm.transform_tobe_mapped = T_in;
  pm.point_on_z_axis = pm.ToMap(P4{0.f, 0.f, 10.f, 0.f}, T_in);  // fixed before the iterations (PointMapping.cc:803-806)
  pm.corner_from_map = corner_map; pm.surf_from_map = surf_map;
  pm.corner_stack_ds = corner_stack; pm.surf_stack_ds = surf_stack;
  pm.OptimizeTransformTobeMapped(four_dof);
  KeyframeRefinement r;
  r.T = pm.transform_tobe_mapped; r.iterations = pm.last_iterations; r.selected = pm.last_selected; r.degenerate = pm.last_degenerate; r.kz = pm.last_kz;
  return r;
}

}  // namespace orc
