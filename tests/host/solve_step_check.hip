// Stand-alone check (host code only, built with hipcc by tests/test_solve_step.py): the device-resident dogleg step
// (csrc/solve_step.h) replayed by its one-thread host executor against the host solver it replaces (host_solver.h:
// solve_dogleg) on a synthetic sliding window — Wo optimised frames, IMU factors between them, a few thousand point-plane
// factors per frame, an extrinsic prior, and a marginalization prior produced by host_solver.h's marginalize() on the
// previous window.  Both run the same problem; the check prints iteration count, successful steps, termination code, the
// cost traces and the largest parameter difference, and exits non-zero when they disagree.
//
// What this covers: every statement of launch B (fold, L S L^T, assembly order, the trust-region state machine, blocked
// L D L^T with the right-hand side carried along, back-substitution, dogleg step, Plus) and of the aux row of launch A.
// What it cannot cover: the device-only register / MFMA forms of the three panel routines and barrier placement — those are
// checked on the GPU (tests/test_gpu_dev_solver.py).
#include <cstdio>
#include <random>

#include "solve_step.h"
using namespace lio;

namespace {
std::mt19937 rng(7);
double nrm(double s = 1.0) { return s * std::normal_distribution<double>(0, 1)(rng); }

struct FrameData { std::vector<double> pts, coef; std::vector<uint8_t> valid; };   // pts xyz (frame i), coef (w, d) in the pivot frame

struct Truth { std::vector<std::array<double, 7>> pose; std::vector<std::array<double, 9>> sb; std::array<double, 7> ex; };

// exact CPU moments of one frame at T_{pivot<-i} = (R, t): the sums the device kernels accumulate (solve_kernels.hip)
void cpu_moments(const FrameData &fd, const double R[9], const double t[3], int b0, int b1, double out[LIO_MOMENT_OUT]) {
  for (int k = 0; k < LIO_MOMENT_OUT; ++k) out[k] = 0;
  double lg = 0, cnt = 0;
  for (int s = b0; s < b1; ++s) {
    if (!fd.valid[s]) continue;
    const double *p = &fd.pts[3 * s], *c = &fd.coef[4 * s];
    const double qx = R[0] * p[0] + R[1] * p[1] + R[2] * p[2] + t[0], qy = R[3] * p[0] + R[4] * p[1] + R[5] * p[2] + t[1],
                 qz = R[6] * p[0] + R[7] * p[1] + R[8] * p[2] + t[2];
    const double r = c[0] * qx + c[1] * qy + c[2] * qz + c[3];
    const double sw = 1.0 / std::sqrt(1.0 + r * r);
    double z[16] = {0};
    for (int a = 0; a < 3; ++a) { z[4 * a] = sw * c[a] * p[0]; z[4 * a + 1] = sw * c[a] * p[1]; z[4 * a + 2] = sw * c[a] * p[2]; z[4 * a + 3] = sw * c[a]; }
    z[12] = sw * c[3];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) out[i * 16 + j] += z[i] * z[j];
    lg += std::log(1.0 + r * r); cnt += 1;
  }
  out[256] = 0.5 * lg; out[257] = cnt;
}

struct Window {
  int Wo;
  WindowParams P;
  std::vector<FrameData> fr;   // [1..Wo]
  WindowSystem sys;
};

void eval_lidar(const Window &w, const WindowParams &P, std::vector<FrameMoments> &m) {
  for (int i = 1; i <= w.Wo; ++i) {
    double R[9], t[3], out[LIO_MOMENT_OUT];
    relative_lidar_pose(P.pose[0].data(), P.pose[i].data(), P.ex.data(), R, t);
    cpu_moments(w.fr[i], R, t, 0, int(w.fr[i].valid.size()), out);
    std::memcpy(m[i].S, out, 256 * sizeof(double));
    m[i].cost = out[256]; m[i].count = out[257];
  }
}

// the device path, emulated: launch A (moments partials + aux row) and launch B (solve_step) until done
struct DevRun { DevState st; int launches = 0; };
bool run_device_emulated(Window &w, int max_iterations, bool conv_flag_in, DevRun &out, WindowParams &Pout) {
  const int Wo = w.Wo, bpf = 7;
  DevProblem pb;
  std::vector<double> prior_mats;
  if (!ds_pack_problem(w.sys, w.P, max_iterations, bpf, conv_flag_in, true, pb, &prior_mats)) return false;
  if (prior_mats.empty()) prior_mats.resize(4);
  DevState &st = out.st;
  ds_init_state(w.P, st);
  std::vector<double> partials(size_t(Wo) * bpf * LIO_MOMENT_OUT), imu_out(size_t(Wo) * DS_IMU_OUT), lmap(size_t(Wo) * DS_LMAP_OUT),
      prior_out(size_t(pb.n_prior) + 8), exprior_out(DS_EXP_OUT), Hcur(size_t(pb.n_pad) * pb.ld), S_buf(size_t(2) * Wo * LIO_MOMENT_OUT),
      lds(ds_lds_doubles(pb.n_pad, Wo)), aux_lds(2048);
  HostExec x;
  StepBuffers B{prior_mats.data(), partials.data(), imu_out.data(), lmap.data(), prior_out.data(), exprior_out.data(), Hcur.data(), S_buf.data(), nullptr};
  for (int k = 0; k <= max_iterations + 1; ++k) {
    if (st.done) break;
    // ---- launch A
    for (int i = 0; i < Wo; ++i) {
      const FrameData &fd = w.fr[i + 1];
      const int ns = int(fd.valid.size());
      for (int b = 0; b < bpf; ++b)
        cpu_moments(fd, st.cand_Rt[i], st.cand_Rt[i] + 9, int((long long)ns * b / bpf), int((long long)ns * (b + 1) / bpf),
                    &partials[(size_t(i) * bpf + b) * LIO_MOMENT_OUT]);
      aux_imu(x, pb.pim[i], st.cand.pose[i], st.cand.sb[i], st.cand.pose[i + 1], st.cand.sb[i + 1], &imu_out[size_t(i) * DS_IMU_OUT], aux_lds.data());
      aux_lmap(x, st.cand.pose[0], st.cand.pose[i + 1], st.cand.ex, &lmap[size_t(i) * DS_LMAP_OUT], aux_lds.data());
    }
    if (pb.have_prior) aux_prior(x, pb, prior_mats.data(), st.cand, prior_out.data(), aux_lds.data());
    if (pb.use_ex_prior) aux_exprior(x, pb, st.cand, exprior_out.data());
    // ---- launch B
    solve_step(x, pb, st, B, lds.data());
    ++out.launches;
  }
  Pout = w.P;
  ds_unpack_params(st.x, Pout);
  return true;
}

Truth make_truth(int nframes, std::vector<std::shared_ptr<Preintegration>> &pims, const PimNoise &noise) {
  Truth T;
  T.pose.resize(nframes); T.sb.resize(nframes);
  T.ex = {0.05, -0.02, -0.08, 0.01, -0.02, 0.015, 0};
  { double s = 0; for (int k = 3; k < 6; ++k) s += T.ex[k] * T.ex[k]; T.ex[6] = std::sqrt(1 - s); }
  T.pose[0] = {1.0, 2.0, 0.5, 0.02, -0.01, 0.3, 0};
  { double s = 0; for (int k = 3; k < 6; ++k) s += T.pose[0][k] * T.pose[0][k]; T.pose[0][6] = std::sqrt(1 - s); }
  T.sb[0] = {1.5, 0.3, 0.0, 0.01, -0.02, 0.015, 0.001, -0.002, 0.0015};
  pims.assign(nframes, nullptr);
  const V3d g(0, 0, -noise.g_norm);
  for (int f = 1; f < nframes; ++f) {
    V3d ba(T.sb[f - 1][3], T.sb[f - 1][4], T.sb[f - 1][5]), bg(T.sb[f - 1][6], T.sb[f - 1][7], T.sb[f - 1][8]);
    V3d a0(0.3 + nrm(0.2), nrm(0.2), 9.8 + nrm(0.2)), w0(nrm(0.05), nrm(0.05), 0.2 + nrm(0.05));
    auto pm = std::make_shared<Preintegration>(a0, w0, ba, bg, noise);
    for (int k = 0; k < 40; ++k) pm->push_back(0.005, V3d(0.3 + 0.2 * std::sin(0.1 * k + f) + nrm(0.02), nrm(0.05), 9.8 + nrm(0.05)),
                                               V3d(nrm(0.01), nrm(0.01), 0.2 + 0.05 * std::cos(0.07 * k) + nrm(0.01)));
    pims[f] = pm;
    // the state the factor is exactly satisfied by (ImuFactor residual = 0), plus a little process noise
    V3d Pi, Pj; Qd Qi;
    unpack_pose(T.pose[f - 1].data(), Pi, Qi);
    V3d Vi(T.sb[f - 1][0], T.sb[f - 1][1], T.sb[f - 1][2]);
    const double dt = pm->sum_dt;
    Pj = Pi + Vi * dt + 0.5 * g * dt * dt + rotate(Qi, pm->dp);
    Qd Qj = normalized(Qi * pm->dq);
    V3d Vj = Vi + g * dt + rotate(Qi, pm->dv);
    T.pose[f] = {Pj.x + nrm(1e-3), Pj.y + nrm(1e-3), Pj.z + nrm(1e-3), Qj.x, Qj.y, Qj.z, Qj.w};
    T.sb[f] = {Vj.x + nrm(1e-3), Vj.y + nrm(1e-3), Vj.z, T.sb[f - 1][3] + nrm(1e-4), T.sb[f - 1][4], T.sb[f - 1][5], T.sb[f - 1][6], T.sb[f - 1][7] + nrm(1e-5), T.sb[f - 1][8]};
  }
  return T;
}

// window over truth frames [f0, f0 + Wo]: lidar factors generated in the pivot (f0) lidar frame at the TRUE poses
Window make_window(const Truth &T, const std::vector<std::shared_ptr<Preintegration>> &pims, int f0, int Wo, int pts_per_frame, bool ex_free,
                   double noise_p, double noise_r) {
  Window w;
  w.Wo = Wo;
  w.P.Wo = Wo; w.P.pose.resize(Wo + 1); w.P.sb.resize(Wo + 1);
  w.P.ex = T.ex; w.P.ex_constant = !ex_free;
  w.fr.resize(Wo + 1);
  for (int i = 0; i <= Wo; ++i) { w.P.pose[i] = T.pose[f0 + i]; w.P.sb[i] = T.sb[f0 + i]; }
  for (int i = 1; i <= Wo; ++i) {
    double R[9], t[3];
    relative_lidar_pose(T.pose[f0].data(), T.pose[f0 + i].data(), T.ex.data(), R, t);
    FrameData &fd = w.fr[i];
    for (int s = 0; s < pts_per_frame; ++s) {
      // a plane in the pivot frame, a point on it (plus noise / a few outliers), expressed in frame i
      V3d wv(nrm(), nrm(), nrm());
      if (s % 3 == 0) wv = V3d(nrm(0.05), nrm(0.05), 1);     // ground-like
      wv = wv / norm(wv);
      V3d q(nrm(15), nrm(15), nrm(2));
      const double d = -dot(wv, q) + nrm(0.02) + (s % 97 == 0 ? 0.5 : 0.0);
      V3d dq(q.x - t[0], q.y - t[1], q.z - t[2]);
      V3d p(R[0] * dq.x + R[3] * dq.y + R[6] * dq.z, R[1] * dq.x + R[4] * dq.y + R[7] * dq.z, R[2] * dq.x + R[5] * dq.y + R[8] * dq.z);
      const double sc = 0.5 + 0.5 * std::uniform_real_distribution<double>(0, 1)(rng);
      fd.pts.insert(fd.pts.end(), {double(float(p.x)), double(float(p.y)), double(float(p.z))});
      fd.coef.insert(fd.coef.end(), {double(float(sc * wv.x)), double(float(sc * wv.y)), double(float(sc * wv.z)), double(float(sc * d))});
      fd.valid.push_back(s % 11 != 0);
    }
  }
  // perturb the starting point
  for (int i = 0; i <= Wo; ++i) {
    for (int k = 0; k < 3; ++k) w.P.pose[i][k] += nrm(noise_p);
    Qd q(w.P.pose[i][6], w.P.pose[i][3], w.P.pose[i][4], w.P.pose[i][5]);
    q = normalized(q * deltaQ(V3d(nrm(noise_r), nrm(noise_r), nrm(noise_r))));
    w.P.pose[i][3] = q.x; w.P.pose[i][4] = q.y; w.P.pose[i][5] = q.z; w.P.pose[i][6] = q.w;
    for (int k = 0; k < 3; ++k) w.P.sb[i][k] += nrm(noise_p);
  }
  w.sys.Wo = Wo; w.sys.use_lidar = true;
  w.sys.pim.assign(Wo, nullptr);
  for (int i = 0; i < Wo; ++i) w.sys.pim[i] = pims[f0 + i + 1];
  w.sys.use_prior_factor = true;
  w.sys.prior_pos = V3d(T.ex[0], T.ex[1], T.ex[2]); w.sys.prior_rot = Qd(T.ex[6], T.ex[3], T.ex[4], T.ex[5]);
  return w;
}

double param_gap(const WindowParams &a, const WindowParams &b) {
  double mx = 0;
  for (int i = 0; i <= a.Wo; ++i) {
    for (int k = 0; k < 7; ++k) mx = std::max(mx, std::fabs(a.pose[i][k] - b.pose[i][k]));
    for (int k = 0; k < 9; ++k) mx = std::max(mx, std::fabs(a.sb[i][k] - b.sb[i][k]));
  }
  for (int k = 0; k < 7; ++k) mx = std::max(mx, std::fabs(a.ex[k] - b.ex[k]));
  return mx;
}

int compare(const char *name, Window &w, int max_it, bool conv_in) {
  // host solver
  Window wh = w;
  wh.sys.lidar_eval = [&wh](const WindowParams &P, std::vector<FrameMoments> &m) { eval_lidar(wh, P, m); };
  WindowParams Ph = wh.P;
  SolveSummary sh = solve_dogleg(wh.sys, Ph, max_it, -1.0, nullptr);
  // emulated device path
  DevRun dr;
  WindowParams Pd;
  if (!run_device_emulated(w, max_it, conv_in, dr, Pd)) { std::printf("%s: does not fit the device path\n", name); return 1; }
  const DevState &st = dr.st;
  double tgap = 0;
  const int nt = std::min<int>(int(sh.trace.size()), st.ntrace);
  for (int k = 0; k < nt; ++k) tgap = std::max(tgap, std::fabs(sh.trace[k] - st.trace[k]) / std::fabs(sh.trace[k]));
  const double pg = param_gap(Ph, Pd);
  std::printf("%s: n=%d host it=%d succ=%d term=%d | dev it=%d succ=%d term=%d need_host=%d launches=%d | trace %zu/%d rel gap %.2e | param gap %.2e | cost %.9g -> %.9g\n",
              name, WindowSystem::solve_layout(w.P).dim, sh.iterations, sh.successful, sh.termination, st.it, st.successful, st.termination, st.need_host,
              dr.launches, sh.trace.size(), st.ntrace, tgap, pg, sh.initial_cost, sh.final_cost);
  int bad = 0;
  if (sh.iterations != st.it || sh.successful != st.successful || sh.termination != st.termination || int(sh.trace.size()) != st.ntrace) bad = 1;
  // same decisions; values agree to the rounding the two factorisations (L D L^T vs Cholesky) and summation orders leave
  if (!(tgap < 1e-6) || !(pg < 1e-7)) bad = 1;
  if (st.need_host) bad = 1;
  return bad;
}
}  // namespace

int main() {
  PimNoise noise;
  noise.acc_n = 0.2; noise.gyr_n = 0.02; noise.acc_w = 0.0002; noise.gyr_w = 2.0e-5; noise.g_norm = 9.8;
  int bad = 0;
  for (int Wo : {5, 2, 7}) {
    std::vector<std::shared_ptr<Preintegration>> pims;
    Truth T = make_truth(Wo + 3, pims, noise);
    for (int ex_free = 0; ex_free <= 1; ++ex_free) {
      if (Wo == 7 && ex_free) continue;   // n = 132 -> padded 144: not on the device path (checked below)
      // window 1: no marginalization prior yet
      Window w1 = make_window(T, pims, 0, Wo, 1500, ex_free, 0.02, 0.003);
      char nm[64];
      std::snprintf(nm, sizeof nm, "Wo=%d ex_free=%d no-prior", Wo, ex_free);
      bad |= compare(nm, w1, 10, true);
      // marginalize window 1 at the host solution -> prior for window 2
      Window wh = w1;
      wh.sys.lidar_eval = [&wh](const WindowParams &P, std::vector<FrameMoments> &m) { eval_lidar(wh, P, m); };
      WindowParams Ph = wh.P;
      solve_dogleg(wh.sys, Ph, 10, -1.0, nullptr);
      WindowSystem ms = wh.sys;
      for (int i = 1; i < Wo; ++i) ms.pim[i] = nullptr;
      auto prior = marginalize(ms, Ph);
      Window w2 = make_window(T, pims, 1, Wo, 1500, ex_free, 0.0, 0.0);
      for (int i = 0; i < Wo; ++i) { w2.P.pose[i] = Ph.pose[i + 1]; w2.P.sb[i] = Ph.sb[i + 1]; }   // the slid window starts from the solved states
      for (int k = 0; k < 3; ++k) w2.P.pose[Wo][k] += 0.03 * (k + 1);
      w2.P.ex = Ph.ex;
      w2.sys.prior = prior;
      std::snprintf(nm, sizeof nm, "Wo=%d ex_free=%d with-prior", Wo, ex_free);
      bad |= compare(nm, w2, 10, true);
      // tight iteration budget and a far start (rejected steps, radius shrinking)
      Window w3 = make_window(T, pims, 1, Wo, 400, ex_free, 0.3, 0.05);
      w3.sys.prior = prior;
      std::snprintf(nm, sizeof nm, "Wo=%d ex_free=%d far-start", Wo, ex_free);
      bad |= compare(nm, w3, 10, true);
    }
  }
  {  // the shape change of the first solves (convergence_flag_ still false with a prior present) must be handed back to the host
    std::vector<std::shared_ptr<Preintegration>> pims;
    Truth T = make_truth(8, pims, noise);
    Window w1 = make_window(T, pims, 0, 5, 300, true, 0.02, 0.003);
    DevRun dr; WindowParams Pd;
    run_device_emulated(w1, 10, false, dr, Pd);
    std::printf("conv_flag_in=0, ex free: need_host=%d conv_out=%d turn_off=%d\n", dr.st.need_host, dr.st.conv_flag_out, dr.st.turn_off);
    if (!(dr.st.need_host == 1 || dr.st.conv_flag_out == 1)) bad = 1;
  }
  std::printf(bad ? "FAIL\n" : "OK\n");
  return bad;
}
