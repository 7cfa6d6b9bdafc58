// oracle/estimator.h — TEST INFRASTRUCTURE ONLY (CPU oracle).  Not part of the product.
//
// CPU restatement of lio::Estimator's post-initialisation hot path:
//   src/imu_processor/Estimator.cc:62-103   TransformToEnd (deskew)
//   :338-427   ProcessImu           :430-488,620-774  ProcessLaserOdom (INITED branch)
//   :970-1097  CalculateFeatures    :1242-1359        CalculateLaserOdom
//   :1361-1646 BuildLocalMap        :1648-2438        SolveOptimization
//   :2440-2568 VectorToDouble / DoubleToVector        :2570-2666 SlideWindow
//   include/utils/CircularBuffer.h:164-172 (push-on-full semantics, A.12)
// Pinned (round 3) against the reference's own Estimator.cc, compiled where it lies against oracle/ref_shim (oracle/ref_estimator.cc ->
// _ref/libref_estimator.so; Ceres' Problem / Solve stood in): thirteen replays from t = 0 (BASELINE.json's HDL-64E / window-15 configuration among them), tests/golden/ref_estimator_run.npz,
// tests/test_ref_estimator_run.py — events, factor counts, iteration counts equal; states within 3e-9 m step by step (1.2e-7 once).
// Compile-time switches of the reference kept at their shipped values: USE_CORNER off, FIX_MAP off
// (Estimator.h:55-56).  The wall-clock solver cap (A.14) is a config knob (max_solver_time).
#pragma once
#include <array>

#include "cloud.h"
#include "imu_init.h"
#include "solver.h"

namespace orc {

typedef Twist<float> Transformf;

struct EstimatorConfig {
  int window_size = 15, opt_window_size = 5;
  float corner_filter_size = 0.2f, surf_filter_size = 0.4f;
  float min_match_sq_dis = 1.0f, min_plane_dis = 0.2f;
  Transformf transform_lb{Q<float>(1, 0, 0, 0), V3<float>(0, 0, -0.1f)};
  bool opt_extrinsic = false, imu_factor = true, point_distance_factor = false, prior_factor = false,
       marginalization_factor = true, enable_deskew = true, cutoff_deskew = false, keep_features = false;
  PimConfig pim;
  int max_num_iterations = 10;
  double max_solver_time = 0.10;
  int extrinsic_stage = 2;        // estimate_extrinsic (Estimator.h:81)
  int init_window_factor = 3;     // Estimator.h:80
};

struct SolveReport {
  int iterations = 0, successful = 0, termination = 0, n_lidar = 0, n_local_map = 0, laser_odom_iters = 0, laser_odom_kz = 0;
  bool turn_off = true, convergence_flag = false, marginalized = false;
  double cost_pim = 0, cost_ppp = 0, cost_marg = 0, initial_cost = 0, final_cost = 0;
  std::vector<double> trace;
  double ms_build_map = 0, ms_features = 0, ms_prepare = 0, ms_opt = 0, ms_marg = 0, ms_total = 0;
};

struct StampedTransform { double time; Transformf transform; };

static inline double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Estimator.cc:62-103 (time_factor = 10 at the call sites :668-670)
static inline void TransformToEnd(Cloud &cloud, const Transformf &tes, float time_factor) {
  for (P4 &p : cloud) {
    float s = time_factor * (p.i - int(p.i));
    p.x -= s * tes.pos.x; p.y -= s * tes.pos.y; p.z -= s * tes.pos.z;
    p.i -= int(p.i);
    Q<float> q_id, q_e = tes.rot;
    Q<float> q_s = q_id.slerp(s, q_e);
    V3<float> v = q_s.conjugate().normalized() * V3<float>(p.x, p.y, p.z);
    v = q_e * v;
    p.x = v.x + tes.pos.x; p.y = v.y + tes.pos.y; p.z = v.z + tes.pos.z;
  }
}

// PointMapping.cc:303-314 PointAssociateToMap
static inline V3<float> AssociateToMap(const V3<float> &p, const Transformf &T) {
  V3<float> r = T.rot * p;
  return V3<float>(r.x + T.pos.x, r.y + T.pos.y, r.z + T.pos.z);
}

struct Estimator {
  EstimatorConfig cfg;
  int W, Wo;
  // circular buffers held as dense vectors of W+1 logical slots
  std::vector<V3d> Ps, Vs, Bas, Bgs;
  std::vector<M3d> Rs;
  std::vector<std::shared_ptr<IntegrationBase>> pre_integrations;
  std::vector<Cloud> surf_stack, corner_stack;
  std::vector<size_t> size_surf_stack, size_corner_stack;
  std::vector<StampedTransform> imu_stamped;  // capacity 100 (Estimator.h:279)
  std::shared_ptr<IntegrationBase> tmp_pre_integration;
  V3d acc_last, gyr_last, g_vec;
  Transformf transform_lb, transform_es;
  bool first_imu = false, inited = false, init_local_map = false, convergence_flag = false;
  int cir_buf_count = 0;
  std::shared_ptr<MargPrior> last_marg;
  // outputs of the last BuildLocalMap
  Cloud local_map_filtered;
  std::vector<std::vector<PlaneFeature>> feature_frames;
  Transformf laser_odom_transform;
  int laser_odom_iters = 0;
  int laser_odom_kz = 0;  // leading update components masked by the degeneracy test of round 0 (Estimator.cc:1308-1339)
  double ms_features_acc = 0;
  int shard_rank = 0, shard_world = 1;
  int (*allreduce)(double *, int, void *) = nullptr;
  void *allreduce_user = nullptr;
  // initialisation stage (Estimator.cc:430-618, 858-958)
  std::vector<LaserTransform> all_laser_transforms;
  int n_state = 0;   // CircularBuffer size of Ps_/Rs_/Vs_/Bas_/Bgs_
  int n_frames = 0;  // CircularBuffer size of pre_integrations_/all_laser_transforms_/the stacks
  int laser_odom_recv_count = 0, extrinsic_stage = 2;
  double initial_time = -1;
  M3d R_WI = M3d::Identity();
  enum Event { EV_SKIPPED = 0, EV_FILLING = 1, EV_INIT_FAILED = 2, EV_INITIALISED = 3, EV_SOLVED = 4 };
  int last_event = EV_SKIPPED;

  explicit Estimator(const EstimatorConfig &c) : cfg(c), W(c.window_size), Wo(c.opt_window_size) {
    transform_lb = c.transform_lb;
    Ps.assign(W + 1, V3d()); Vs = Bas = Bgs = Ps;
    Rs.assign(W + 1, M3d::Identity());
    pre_integrations.assign(W + 1, nullptr);
    surf_stack.assign(W + 1, Cloud()); corner_stack.assign(W + 1, Cloud());
    size_surf_stack.assign(W + 1, 0); size_corner_stack.assign(W + 1, 0);
    all_laser_transforms.assign(W + 1, LaserTransform());
    extrinsic_stage = c.extrinsic_stage;
    g_vec = V3d(0, 0, -c.pim.g_norm);
    // ClearState (Estimator.cc:234-288): tmp_pre_integration_ exists from the start (acc/gyr zero until the first IMU)
    tmp_pre_integration = std::make_shared<IntegrationBase>(acc_last, gyr_last, Bas[0], Bgs[0], cfg.pim);
  }

  template <typename T> static void pushFull(std::vector<T> &buf, const T &v) {  // CircularBuffer.h:164-172 on a full buffer
    for (size_t i = 0; i + 1 < buf.size(); ++i) buf[i] = buf[i + 1];
    buf.back() = v;
  }
  // CircularBuffer::push with `size` elements held: append while there is room, else drop the oldest
  template <typename T> static void pushAt(std::vector<T> &buf, int size, const T &v) {
    if (size < int(buf.size())) buf[size] = v; else pushFull(buf, v);
  }
  void pushState(int from) {  // Ps_.push(Ps_[from]) ... (Estimator.cc:2646-2651)
    const V3d p = Ps[from], v = Vs[from], ba = Bas[from], bg = Bgs[from];
    const M3d r = Rs[from];
    pushAt(Ps, n_state, p); pushAt(Vs, n_state, v); pushAt(Rs, n_state, r); pushAt(Bas, n_state, ba); pushAt(Bgs, n_state, bg);
    if (n_state < W + 1) ++n_state;
  }

  // ---- Estimator.cc:338-427 (steady state: buffers full, cir_buf_count == W)
  void ProcessImu(double dt, const V3d &acc, const V3d &gyr, double stamp) {
    if (!first_imu) {
      first_imu = true; acc_last = acc; gyr_last = gyr;
      if (n_state == 0) { Ps[0] = V3d(); Vs[0] = V3d(); Bas[0] = V3d(); Bgs[0] = V3d(); Rs[0] = M3d::Identity(); n_state = 1; }  // :342-354
    }
    if (cir_buf_count != 0) {
      if (tmp_pre_integration) tmp_pre_integration->push_back(dt, acc, gyr);
      int j = cir_buf_count;
      V3d un_acc_0 = Rs[j] * (acc_last - Bas[j]) + g_vec;
      V3d un_gyr = 0.5 * (gyr_last + gyr) - Bgs[j];
      Rs[j] = Rs[j] * DeltaQ(un_gyr * dt).toRotationMatrix();
      V3d un_acc_1 = Rs[j] * (acc - Bas[j]) + g_vec;
      V3d un_acc = 0.5 * (un_acc_0 + un_acc_1);
      Ps[j] += dt * Vs[j] + 0.5 * dt * dt * un_acc;
      Vs[j] += dt * un_acc;
      StampedTransform tt;
      tt.time = stamp;
      tt.transform.pos = Ps[j].cast<float>();
      tt.transform.rot = Q<float>::FromMatrix(Rs[j].cast<float>());
      if (imu_stamped.size() >= 100) imu_stamped.erase(imu_stamped.begin());
      imu_stamped.push_back(tt);
    }
    acc_last = acc; gyr_last = gyr;
  }

  void BeginFrame(const V3d &acc, const V3d &gyr) {
    acc_last = acc; gyr_last = gyr; first_imu = true;
    tmp_pre_integration = std::make_shared<IntegrationBase>(acc_last, gyr_last, Bas[cir_buf_count], Bgs[cir_buf_count], cfg.pim);
  }

  // ---- Estimator.cc:430-774: frame push, initialisation stage, INITED branch
  bool ProcessLaserOdom(const Transformf &transform_in, Cloud surf_last, Cloud corner_last, double stamp, SolveReport *rep) {
    ++laser_odom_recv_count;
    if (!inited && laser_odom_recv_count % cfg.init_window_factor != 0) { last_event = EV_SKIPPED; return true; }  // :436-439
    if (!PushFrame(transform_in, std::move(surf_last), std::move(corner_last), stamp)) return false;
    if (!inited) {
      if (cir_buf_count == W) {
        bool init_result = false;
        if (!cfg.imu_factor) {
          init_result = true;
          SetStatesFromLaser();
        } else {
          if (extrinsic_stage == 2) {
            if (EstimateExtrinsicRotation(all_laser_transforms, transform_lb)) extrinsic_stage = 1;
          }
          if (extrinsic_stage != 2 && (stamp - initial_time) > 0.1) {
            init_result = RunInitialization();
            initial_time = stamp;
          }
        }
        if (init_result) {
          inited = true;
          SolveOptimization(rep);
          SlideWindow();
          last_event = EV_INITIALISED;
        } else {
          SlideWindow();
          last_event = EV_INIT_FAILED;
        }
      } else {
        SlideWindow();
        ++cir_buf_count;
        last_event = EV_FILLING;
      }
      return true;
    }
    SolveOptimization(rep);
    SlideWindow();
    last_event = EV_SOLVED;
    return true;
  }
  void SetStatesFromLaser() {  // :507-513, :892-906
    for (int i = 0; i <= W; ++i) {
      Transformf trans_bi = all_laser_transforms[i].transform * transform_lb;
      Ps[i] = trans_bi.pos.cast<double>();
      Rs[i] = trans_bi.rot.normalized().toRotationMatrix().cast<double>();
    }
  }
  // ---- Estimator.cc:858-958
  bool RunInitialization() {
    {
      V3d sum_g;  // the reference leaves it uninitialised (Estimator.cc:863); zero is the only meaningful reading
      for (int i = 0; i < W; ++i) {
        const IntegrationBase &pim = *all_laser_transforms[i + 1].pre_integration;
        sum_g += pim.delta_v_ / pim.sum_dt_;
      }
      V3d aver_g = sum_g * (1.0 / W);
      double var = 0;
      for (int i = 0; i < W; ++i) {
        const IntegrationBase &pim = *all_laser_transforms[i + 1].pre_integration;
        V3d d = pim.delta_v_ / pim.sum_dt_ - aver_g;
        var += d.dot(d);
      }
      var = std::sqrt(var / W);
      if (var < 0.25) return false;  // "IMU excitation not enough!"
    }
    V3d g_vec_in_laser;
    bool init_result = Initialization(all_laser_transforms, Vs, Bgs, g_vec_in_laser, transform_lb, R_WI);
    SetStatesFromLaser();
    M3d R0 = R_WI.transpose();
    double yaw = R2ypr(R0 * Rs[0]).x;
    R0 = ypr2R(V3d(-yaw, 0, 0)) * R0;
    R_WI = R0.transpose();
    g_vec = R0 * g_vec_in_laser;
    for (int i = 0; i <= cir_buf_count; ++i) pre_integrations[i]->Repropagate(Bas[i], Bgs[i]);
    for (int i = 0; i <= cir_buf_count; ++i) { Ps[i] = R0 * Ps[i]; Rs[i] = R0 * Rs[i]; Vs[i] = R0 * Vs[i]; }
    return init_result;
  }
  bool PushFrame(const Transformf &transform_in, Cloud surf_last, Cloud corner_last, double stamp) {
    LaserTransform lt;
    lt.time = stamp; lt.transform = transform_in; lt.pre_integration = tmp_pre_integration;
    pushAt(pre_integrations, n_frames, tmp_pre_integration);
    pushAt(all_laser_transforms, n_frames, lt);
    tmp_pre_integration = std::make_shared<IntegrationBase>(acc_last, gyr_last, Bas[cir_buf_count], Bgs[cir_buf_count], cfg.pim);
    const int n_before = n_frames;
    if (n_frames < W + 1) ++n_frames;
    if (!inited) {  // :474-481: the stacks are PointMapping's down-sampled clouds (sensor frame)
      pushAt(size_surf_stack, n_before, surf_last.size()); pushAt(surf_stack, n_before, surf_last);
      pushAt(size_corner_stack, n_before, corner_last.size()); pushAt(corner_stack, n_before, corner_last);
      return true;
    }
    if (cfg.enable_deskew || cfg.cutoff_deskew) {
      if (!cfg.cutoff_deskew) {
        if (imu_stamped.empty()) return false;
        double time_e = imu_stamped.back().time;
        Transformf transform_e = imu_stamped.back().transform;
        double time_s = time_e;
        Transformf transform_s = transform_e;
        for (int i = int(imu_stamped.size()) - 1; i >= 0; --i) {
          time_s = imu_stamped[i].time;
          transform_s = imu_stamped[i].transform;
          if (time_e - imu_stamped[i].time >= 0.1) break;
        }
        Transformf body_es = transform_e.inverse() * transform_s;
        {
          float s = float(0.1 / (time_e - time_s));
          Q<float> q_id;
          body_es.rot = q_id.slerp(s, body_es.rot);
          body_es.pos = s * body_es.pos;
        }
        transform_es = transform_lb * body_es * transform_lb.inverse();
        TransformToEnd(surf_last, transform_es, 10);
        TransformToEnd(corner_last, transform_es, 10);
      }
      Cloud surf_ds, corner_ds;
      VoxelGrid(surf_last, cfg.surf_filter_size, surf_ds);
      VoxelGrid(corner_last, cfg.corner_filter_size, corner_ds);
      pushFull(size_surf_stack, surf_ds.size()); pushFull(surf_stack, surf_ds);
      pushFull(size_corner_stack, corner_ds.size()); pushFull(corner_stack, corner_ds);
    } else {
      // reference pushes laser_cloud_*_stack_downsampled_ produced by PointMapping (out of scope):
      // the caller's clouds are taken as already down-sampled.
      pushFull(size_surf_stack, surf_last.size()); pushFull(surf_stack, surf_last);
      pushFull(size_corner_stack, corner_last.size()); pushFull(corner_stack, corner_last);
    }
    return true;
  }

  Twist<double> LidarPose(int i, const Twist<double> &lb) const {
    Qd rot = Qd::FromMatrix(Rs[i] * lb.rot.inverse().toRotationMatrix());
    V3d pos = Ps[i] - rot * lb.pos;
    return Twist<double>(rot, pos);
  }

  // ---- Estimator.cc:970-1097, surf branch.  Shared with the stateless C entry point.
  static void CalculateFeatures(const KdTree &tree, const Cloud &map, const Cloud &stack, const Transformf &T, float min_match_sq_dis,
                                float min_plane_dis, bool keep, std::vector<PlaneFeature> &features, std::vector<uint8_t> *valid_out = nullptr,
                                std::vector<std::array<float, 5>> *raw_out = nullptr) {
    if (!keep) features.clear();
    if (valid_out) valid_out->assign(stack.size(), 0);
    if (raw_out) raw_out->assign(stack.size(), std::array<float, 5>{0, 0, 0, 0, 0});
    for (size_t i = 0; i < stack.size(); ++i) {
      const P4 &po = stack[i];
      V3<float> sel = AssociateToMap(V3<float>(po.x, po.y, po.z), T);
      P4 q{sel.x, sel.y, sel.z, po.i};
      int idx[5]; float sq[5];
      int found = tree.Search(q, 5, idx, sq);
      if (found < 5) continue;
      if (sq[4] < min_match_sq_dis) {
        float A[15], B[5] = {-1, -1, -1, -1, -1}, X[3];
        for (int j = 0; j < 5; ++j) { A[j * 3 + 0] = map[idx[j]].x; A[j * 3 + 1] = map[idx[j]].y; A[j * 3 + 2] = map[idx[j]].z; }
        colpiv_qr_solve<float>(5, 3, A, B, X);
        float pa = X[0], pb = X[1], pc = X[2], pd = 1;
        float ps = std::sqrt(pa * pa + pb * pb + pc * pc);
        pa /= ps; pb /= ps; pc /= ps; pd /= ps;
        bool planeValid = true;
        for (int j = 0; j < 5; ++j) {
          if (std::fabs(pa * map[idx[j]].x + pb * map[idx[j]].y + pc * map[idx[j]].z + pd) > min_plane_dis) { planeValid = false; break; }
        }
        if (!planeValid) continue;
        float pd2 = pa * sel.x + pb * sel.y + pc * sel.z + pd;
        float s = 1 - 0.9f * std::fabs(pd2) / std::sqrt(std::sqrt(sel.x * sel.x + sel.y * sel.y + sel.z * sel.z));
        float c0 = s * pa, c1 = s * pb, c2 = s * pc, c3 = s * pd;
        bool in_fov = false;
        V3<float> pz = AssociateToMap(V3<float>(0.f, 0.f, 10.f), T);
        float dx1 = T.pos.x - sel.x, dy1 = T.pos.y - sel.y, dz1 = T.pos.z - sel.z;
        float side1 = dx1 * dx1 + dy1 * dy1 + dz1 * dz1;
        float dx2 = pz.x - sel.x, dy2 = pz.y - sel.y, dz2 = pz.z - sel.z;
        float side2 = dx2 * dx2 + dy2 * dy2 + dz2 * dz2;
        float check1 = 100.0f + side1 - side2 - 10.0f * std::sqrt(3.0f) * std::sqrt(side1);
        float check2 = 100.0f + side1 - side2 + 10.0f * std::sqrt(3.0f) * std::sqrt(side1);
        if (check1 < 0 && check2 > 0) in_fov = true;
        if (s > 0.1 && in_fov) {
          PlaneFeature f;
          f.score = s;
          f.point = V3d(po.x, po.y, po.z);
          f.coeffs[0] = c0; f.coeffs[1] = c1; f.coeffs[2] = c2; f.coeffs[3] = c3;
          features.push_back(f);
          if (valid_out) (*valid_out)[i] = 1;
          if (raw_out) (*raw_out)[i] = {c0, c1, c2, c3, s};
        }
      }
    }
  }

  // ---- Estimator.cc:1242-1359
  void CalculateLaserOdom(const KdTree &tree, const Cloud &map, const Cloud &stack, Transformf &T, std::vector<PlaneFeature> &features) {
    bool is_degenerate = false;
    float matP[36];
    laser_odom_iters = 0; laser_odom_kz = 0;
    for (size_t iter = 0; iter < 10; ++iter) {  // num_max_iterations_ = 10 (PointMapping.h:171)
      ++laser_odom_iters;
      CalculateFeatures(tree, map, stack, T, cfg.min_match_sq_dis, cfg.min_plane_dis, cfg.keep_features, features);
      size_t n = features.size();
      float AtA[36] = {0}, AtB[6] = {0};
      Q<float> R0 = T.rot.normalized();  // SO3 ctor normalises (so3.hpp)
      M3<float> Rm = T.rot.toRotationMatrix();  // Quaternion * Matrix goes through toRotationMatrix()
      // Eigen evaluates mat_At * mat_A / mat_At * mat_B as GEMM/GEMV in float; the summation order inside
      // Eigen's kernels is unspecified — the oracle sums rows sequentially in float.
      for (size_t i = 0; i < n; ++i) {
        const PlaneFeature &f = features[i];
        V3<float> p{float(f.point.x), float(f.point.y), float(f.point.z)};
        V3<float> w{float(f.coeffs[0]), float(f.coeffs[1]), float(f.coeffs[2])};
        float ci = float(f.coeffs[3]);
        M3<float> RS = Rm * Skew(p);
        // J_r = -w^T (R skew(p))
        float a[6];
        a[0] = -(w.x * RS(0, 0) + w.y * RS(1, 0) + w.z * RS(2, 0));
        a[1] = -(w.x * RS(0, 1) + w.y * RS(1, 1) + w.z * RS(2, 1));
        a[2] = -(w.x * RS(0, 2) + w.y * RS(1, 2) + w.z * RS(2, 2));
        a[3] = w.x; a[4] = w.y; a[5] = w.z;
        V3<float> rp = T.rot * p;
        float d2 = w.x * (rp.x + T.pos.x) + w.y * (rp.y + T.pos.y) + w.z * (rp.z + T.pos.z) + ci;
        float bb = -d2;
        for (int r = 0; r < 6; ++r) { for (int c = 0; c < 6; ++c) AtA[r * 6 + c] += a[r] * a[c]; AtB[r] += a[r] * bb; }
      }
      float Acopy[36], Bcopy[6], X[6];
      std::memcpy(Acopy, AtA, sizeof(AtA)); std::memcpy(Bcopy, AtB, sizeof(AtB));
      colpiv_qr_solve<float>(6, 6, Acopy, Bcopy, X);
      if (iter == 0) {
        float E[6], V[36];
        sym_eigen<float>(6, AtA, E, V);
        is_degenerate = false;
        // A.6: matP = V2 * V^-1 with leading ROWS of V zeroed == diag(0..0,1..1)
        for (int k = 0; k < 36; ++k) matP[k] = 0;
        int kz = 0;
        for (int i = 0; i < 6; ++i) { if (E[i] < 100.f) { ++kz; is_degenerate = true; } else break; }
        for (int i = kz; i < 6; ++i) matP[i * 6 + i] = 1.f;
        laser_odom_kz = kz;
      }
      if (is_degenerate) {
        float X2[6];
        for (int i = 0; i < 6; ++i) { float s = 0; for (int j = 0; j < 6; ++j) s += matP[i * 6 + j] * X[j]; X2[i] = s; }
        std::memcpy(X, X2, sizeof(X));
      }
      T.pos.x += X[3]; T.pos.y += X[4]; T.pos.z += X[5];
      T.rot = T.rot * DeltaQ(V3<float>(X[0], X[1], X[2]));
      if (!std::isfinite(T.pos.x)) T.pos.x = 0;
      if (!std::isfinite(T.pos.y)) T.pos.y = 0;
      if (!std::isfinite(T.pos.z)) T.pos.z = 0;
      float delta_r = RadToDeg(R0.angularDistance(T.rot));
      float delta_t = std::sqrt(std::pow(X[3] * 100, 2) + std::pow(X[4] * 100, 2) + std::pow(X[5] * 100, 2));
      if (delta_r < 0.05 && delta_t < 0.05) break;  // delta_*_abort_ doubles (PointMapping.cc:75-76)
    }
  }

  // ---- Estimator.cc:1361-1646
  void BuildLocalMap(SolveReport *rep) {
    double t0 = now_ms();
    feature_frames.assign(W + 1, {});
    int pivot = W - Wo;
    Twist<double> lb = transform_lb.cast<double>();
    Twist<double> T_pivot = LidarPose(pivot, lb);
    auto relTransform = [&](int i) {
      Twist<double> T_li = LidarPose(i, lb);
      Twist<float> tf = (T_pivot.inverse() * T_li).cast<float>();
      // .transform(): Affine3f with linear = rot.normalized().toRotationMatrix()
      return tf;
    };
    auto transformCloud = [](const Cloud &in, const Twist<float> &tf, Cloud &out) {
      M3<float> R = tf.linear();
      out.resize(in.size());
      for (size_t k = 0; k < in.size(); ++k) {
        const P4 &p = in[k];
        // pcl::transformPointCloud: m00*x + m01*y + m02*z + m03
        out[k].x = R(0, 0) * p.x + R(0, 1) * p.y + R(0, 2) * p.z + tf.pos.x;
        out[k].y = R(1, 0) * p.x + R(1, 1) * p.y + R(1, 2) * p.z + tf.pos.y;
        out[k].z = R(2, 0) * p.x + R(2, 1) * p.y + R(2, 2) * p.z + tf.pos.z;
        out[k].i = p.i;
      }
    };
    if (!init_local_map) {  // A.15
      Cloud tmp, tr;
      for (int i = 0; i <= pivot; ++i) {
        transformCloud(surf_stack[i], relTransform(i), tr);
        tmp.insert(tmp.end(), tr.begin(), tr.end());
      }
      surf_stack[pivot] = tmp;
      init_local_map = true;
    }
    std::vector<Transformf> local_transforms;
    Cloud local;
    for (int i = 0; i < W + 1; ++i) {
      Twist<float> tf = relTransform(i);
      // Transform local_transform = transform_pivot_i  (Twist(Affine): quaternion of the matrix, normalised)
      local_transforms.push_back(Transformf::FromAffine(tf.linear(), tf.pos));
      if (i < pivot) continue;
      if (i != W) {
        if (i == pivot) { local.insert(local.end(), surf_stack[i].begin(), surf_stack[i].end()); continue; }
        Cloud tr;
        transformCloud(surf_stack[i], tf, tr);
        for (P4 &p : tr) p.i = float(i);
        local.insert(local.end(), tr.begin(), tr.end());
      }
    }
    VoxelGrid(local, cfg.surf_filter_size, local_map_filtered);
    double t1 = now_ms();
    KdTree tree;
    tree.Build(local_map_filtered);
    ms_features_acc = 0;
    for (int idx = 0; idx < W + 1; ++idx) {
      double tf0 = now_ms();
      if (idx > pivot) {
        if (idx != W || !cfg.imu_factor) {
          CalculateFeatures(tree, local_map_filtered, surf_stack[idx], local_transforms[idx], cfg.min_match_sq_dis, cfg.min_plane_dis,
                            cfg.keep_features, feature_frames[idx]);
        } else {
          Transformf T = local_transforms[idx];
          CalculateLaserOdom(tree, local_map_filtered, surf_stack[idx], T, feature_frames[idx]);
          laser_odom_transform = T;
        }
      }
      ms_features_acc += now_ms() - tf0;
    }
    if (rep) { rep->ms_build_map = t1 - t0; rep->ms_features = ms_features_acc; rep->n_local_map = int(local_map_filtered.size()); rep->laser_odom_iters = laser_odom_iters; rep->laser_odom_kz = laser_odom_kz; }
  }

  void VectorToProblem(WindowProblem &P) const {  // Estimator.cc:2440-2477
    int pivot = W - Wo;
    P.Wo = Wo;
    P.pose.resize(Wo + 1); P.sb.resize(Wo + 1);
    for (int i = 0, oi = pivot; i <= Wo; ++i, ++oi) {
      Qd q = Qd::FromMatrix(Rs[oi]);
      P.pose[i] = {Ps[oi].x, Ps[oi].y, Ps[oi].z, q.x, q.y, q.z, q.w};
      P.sb[i] = {Vs[oi].x, Vs[oi].y, Vs[oi].z, Bas[oi].x, Bas[oi].y, Bas[oi].z, Bgs[oi].x, Bgs[oi].y, Bgs[oi].z};
    }
    P.ex = {transform_lb.pos.x, transform_lb.pos.y, transform_lb.pos.z, transform_lb.rot.x, transform_lb.rot.y, transform_lb.rot.z, transform_lb.rot.w};
  }

  void DoubleToVector(const WindowProblem &P) {  // Estimator.cc:2479-2568
    int pivot = W - Wo;
    V3d origin_P0 = Ps[pivot];
    V3d origin_R0 = R2ypr(Rs[pivot]);
    Qd q0(P.pose[0][6], P.pose[0][3], P.pose[0][4], P.pose[0][5]);
    M3d R00 = q0.normalized().toRotationMatrix();
    V3d origin_R00 = R2ypr(R00);
    double y_diff = origin_R0.x - origin_R00.x;
    M3d rot_diff = ypr2R(V3d(y_diff, 0, 0));
    if (std::fabs(std::fabs(origin_R0.y) - 90) < 1.0 || std::fabs(std::fabs(origin_R00.y) - 90) < 1.0)
      rot_diff = Rs[pivot] * R00.transpose();
    {
      Twist<double> trans_pivot{Qd::FromMatrix(Rs[pivot]), Ps[pivot]};
      M3d R_opt_pivot = rot_diff * R00;
      Twist<double> trans_opt_pivot{Qd::FromMatrix(R_opt_pivot), origin_P0};
      for (int idx = 0; idx < pivot; ++idx) {
        Twist<double> trans_idx{Qd::FromMatrix(Rs[idx]), Ps[idx]};
        Twist<double> t = trans_opt_pivot * trans_pivot.inverse() * trans_idx;
        Ps[idx] = t.pos;
        Rs[idx] = t.rot.normalized().toRotationMatrix();
      }
    }
    for (int i = 0, oi = pivot; i <= Wo; ++i, ++oi) {
      Qd qi(P.pose[i][6], P.pose[i][3], P.pose[i][4], P.pose[i][5]);
      Rs[oi] = rot_diff * qi.normalized().toRotationMatrix();
      Ps[oi] = rot_diff * V3d(P.pose[i][0] - P.pose[0][0], P.pose[i][1] - P.pose[0][1], P.pose[i][2] - P.pose[0][2]) + origin_P0;
      Vs[oi] = rot_diff * V3d(P.sb[i][0], P.sb[i][1], P.sb[i][2]);
      Bas[oi] = V3d(P.sb[i][3], P.sb[i][4], P.sb[i][5]);
      Bgs[oi] = V3d(P.sb[i][6], P.sb[i][7], P.sb[i][8]);
    }
    transform_lb.pos = V3d(P.ex[0], P.ex[1], P.ex[2]).cast<float>();
    transform_lb.rot = Qd(P.ex[6], P.ex[3], P.ex[4], P.ex[5]).cast<float>();
  }

  // ---- Estimator.cc:1648-2438
  bool SolveOptimization(SolveReport *rep) {
    if (cir_buf_count < W && cfg.imu_factor) return false;
    double t_total0 = now_ms();
    SolveReport local_rep;
    SolveReport &R = rep ? *rep : local_rep;
    R = SolveReport();
    bool turn_off = true;
    BuildLocalMap(&R);
    double t_prep0 = now_ms();
    int pivot = W - Wo;
    WindowProblem P;
    VectorToProblem(P);
    P.ex_constant = (cfg.extrinsic_stage == 0 || !cfg.opt_extrinsic);
    P.use_imu = cfg.imu_factor;
    P.use_lidar = cfg.point_distance_factor;
    P.shard_rank = shard_rank; P.shard_world = shard_world; P.allreduce = allreduce; P.allreduce_user = allreduce_user;
    P.pim.assign(Wo, nullptr);
    if (cfg.imu_factor)
      for (int i = 0; i < Wo; ++i) {
        auto &pi = pre_integrations[pivot + i + 1];
        if (pi && pi->sum_dt_ <= 10.0) P.pim[i] = pi;  // :1799 skip when sum_dt_ > 10
      }
    P.feats.assign(Wo + 1, {});
    R.n_lidar = 0;
    if (cfg.point_distance_factor)
      for (int i = 1; i <= Wo; ++i) { P.feats[i] = feature_frames[pivot + i]; R.n_lidar += int(P.feats[i].size()); }
    if (cfg.marginalization_factor && last_marg) P.prior = last_marg;
    if (cfg.prior_factor) {
      P.use_prior_factor = true;
      Twist<double> t = transform_lb.cast<double>();
      P.prior_pos = t.pos; P.prior_rot = t.rot;
    }
    R.ms_prepare = now_ms() - t_prep0;
    // pre-solve group costs and convergence_flag_ logic (:1924-1984)
    {
      Layout lay = SolveLayout(P);
      GroupCosts gc;
      EvaluateProblem(P, lay, 1 | 2 | 4, false, nullptr, nullptr, &gc);
      R.cost_pim = gc.pim; R.cost_ppp = gc.ppp; R.cost_marg = gc.marg;
      if (cfg.imu_factor) turn_off = gc.pim > 1e3;
      double ratio = gc.marg / (gc.ppp + gc.pim);
      if (!convergence_flag && !turn_off && ratio <= 2 && ratio != 0) convergence_flag = true;
      if (!convergence_flag) {
        P.ex_constant = true;
        last_marg.reset();
        P.prior.reset();
      }
    }
    double t_opt0 = now_ms();
    // sharded mode: the number of collectives per solve must not depend on a per-rank clock (include/lio_c.h)
    SolveSummary s = SolveDogleg(P, cfg.max_num_iterations, (shard_world > 1 && allreduce) ? -1.0 : cfg.max_solver_time);
    R.ms_opt = now_ms() - t_opt0;
    R.iterations = s.iterations; R.successful = s.successful; R.termination = s.termination;
    R.initial_cost = s.initial_cost; R.final_cost = s.final_cost; R.trace = s.cost_trace;
    DoubleToVector(P);
    R.turn_off = turn_off; R.convergence_flag = convergence_flag;
    // marginalization (:2040-2275)
    if (cfg.marginalization_factor && !turn_off) {
      double tm0 = now_ms();
      WindowProblem M;
      VectorToProblem(M);
      M.ex_constant = false;
      M.use_imu = cfg.imu_factor; M.use_lidar = cfg.point_distance_factor;
      M.shard_rank = shard_rank; M.shard_world = shard_world; M.allreduce = allreduce; M.allreduce_user = allreduce_user;
      M.pim.assign(Wo, nullptr);
      if (cfg.imu_factor) {
        auto &pi = pre_integrations[pivot + 1];
        if (pi && pi->sum_dt_ < 10.0) M.pim[0] = pi;  // :2076
      }
      M.feats = P.feats;
      M.prior = last_marg;
      last_marg = Marginalize(M, 4);
      R.marginalized = true;
      R.ms_marg = now_ms() - tm0;
    }
    R.ms_total = now_ms() - t_total0;
    return true;
  }

  // ---- Estimator.cc:2570-2666
  void SlideWindow() {
    if (init_local_map) {
      int pivot = W - Wo;
      Twist<double> lb = transform_lb.cast<double>();
      Twist<double> T_pivot = LidarPose(pivot, lb);
      int i = pivot + 1;
      Twist<double> T_li = LidarPose(i, lb);
      Twist<float> tf = (T_li.inverse() * T_pivot).cast<float>();
      M3<float> Rm = tf.linear();
      const Cloud &src = surf_stack[pivot];
      Cloud filtered;
      size_t drop = size_surf_stack[0];
      for (size_t k = drop; k < src.size(); ++k) {
        const P4 &p = src[k];
        P4 o;
        o.x = Rm(0, 0) * p.x + Rm(0, 1) * p.y + Rm(0, 2) * p.z + tf.pos.x;
        o.y = Rm(1, 0) * p.x + Rm(1, 1) * p.y + Rm(1, 2) * p.z + tf.pos.y;
        o.z = Rm(2, 0) * p.x + Rm(2, 1) * p.y + Rm(2, 2) * p.z + tf.pos.z;
        o.i = p.i;
        filtered.push_back(o);
      }
      filtered.insert(filtered.end(), surf_stack[i].begin(), surf_stack[i].end());
      surf_stack[i] = filtered;
    }
    pushState(cir_buf_count);
  }
};

}  // namespace orc
