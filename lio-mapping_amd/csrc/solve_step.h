// solve_step.h — the trust-region iteration of Estimator::SolveOptimization (Estimator.cc:1909-1990: ceres::Solve with
// DOGLEG / DENSE_SCHUR, Appendix B.3) as device-resident code.
//
// One dogleg iteration is two launches on the estimator's stream, with no host round trip between them:
//
//   A  k_lidar_moments_dev   (solve_kernels.hip)   S_i = sum rho' z z^T of the lidar factors at the CANDIDATE poses held in
//                                                   device memory, per-block partials; one extra grid row evaluates, in
//                                                   parallel with it, everything that depends on the candidate but not on
//                                                   the points: the Wo ImuFactor blocks (ImuFactor.h:53-168) whitened and
//                                                   squared (30x30 each), the marginalization prior's gradient and cost
//                                                   (MarginalizationFactor.cc:343-393), the extrinsic PriorFactor
//                                                   (PriorFactor.cc:35-67) and the 18x13 lidar linear maps L_i;
//   B  k_solve_step           (solve_device.hip)   ONE workgroup: folds the partials, expands L S L^T, assembles H and g in
//                                                   LDS, decides on the pending candidate (step quality, radius, mu:
//                                                   Ceres' TrustRegionMinimizer), factors H + mu D^2 = L D L^T in LDS
//                                                   (16-wide panels: a register-resident diagonal block on one wave,
//                                                   row-parallel triangular solves, fp64-MFMA trailing updates), forms the
//                                                   dogleg step and writes the next candidate (poses + T_{pivot<-i}).
//
// The host enqueues the whole chain, reads back the final parameters, the summary and the moments at the accepted point.
//
// Everything here is written ONCE for two executors: the kernel's (512 threads, LDS, wave shuffles) and a one-thread host
// emulation (tests/host/solve_step_check.hip) that replays the same statements sequentially — the control flow, index
// arithmetic and numerics of the kernel are therefore checked on the CPU against host_solver.h's solve_dogleg; only the
// few `if constexpr (X::kDevice)` blocks (register-resident panel, MFMA tiles, shuffles) are device-only.
#pragma once
#include "host_solver.h"

namespace lio {

#define DS_MAX_WO 7
#define DS_MAX_NPAD 128      // padded tangent dimension the LDS-resident factorisation takes (15 (Wo + 1) + 6, rounded up to 16)
#define DS_NB 16
#define DS_PART 512    // doubles of the factorisation's scratch: pivot exchange 64 | staging block 256; the 32 x 16 slices of the back-substitution
#ifndef DS_THREADS
#define DS_THREADS 256     // threads of the step kernel's workgroup: one wave per SIMD, 512 registers per lane (at 512 threads the kernel spilled 230 VGPRs to scratch: 118 -> 153 us per step, profiles/r5_g_*; tools/r5/build_variant.sh builds the other form)
#endif
#define DS_MAX_KEEP (DS_MAX_WO + 3)
#define DS_IMU_OUT 932       // 900 J^T J + 30 J^T r + cost + present
#define DS_LMAP_OUT 248      // 234 L + 13 l + pad
#define DS_EXP_OUT 44        // 36 + 6 + cost + pad

struct DevPim {
  double dp[3], dq[4], dv[3], ba[3], bg[3], g[3], sum_dt;
  int present, pad;
  double jac[225];
  double sqrt_info[225];
};

struct DevParams {
  double pose[DS_MAX_WO + 1][7];
  double sb[DS_MAX_WO + 1][9];
  double ex[7];
};

// constant during one solve
struct DevProblem {
  int Wo, n, n_pad, ld, ex_col, n_prior, have_prior, use_ex_prior, max_iterations, bpf, conv_flag_in, imu_on;
  int n_keep;
  int keep_kind[DS_MAX_KEEP], keep_index[DS_MAX_KEEP], keep_size[DS_MAX_KEEP], keep_idx[DS_MAX_KEEP], keep_x0[DS_MAX_KEEP];
  int prior_col[DS_MAX_NPAD + 8];      // tangent column of H -> column of the prior (-1: none)
  double prior_x0[DS_MAX_KEEP * 9];
  double ex_prior_pos[3], ex_prior_rot[4];   // rot as w, x, y, z
  DevPim pim[DS_MAX_WO];
};

struct DevState {
  DevParams x, cand;
  double cand_Rt[DS_MAX_WO][12];
  double x_cost, x_norm, radius, mu, alpha, dogleg_norm, gmax, model_change, step_norm, n_lidar;
  int reuse, invalid, it, successful, termination, done, need_host, started;
  int turn_off, conv_flag_out, s_cur, ntrace;
  double trace[40];
  double costs0[4];      // marg, pim, ppp, extrinsic prior at the initial point (Estimator.cc:1924-1954)
  double scale[DS_MAX_NPAD], diag[DS_MAX_NPAD], grad[DS_MAX_NPAD], gn[DS_MAX_NPAD], g[DS_MAX_NPAD];
};

// prior_mats layout: JtJ (np x np) | lin_jac (np x np) | lin_res (np) | Jtr0 (np)
LIO_HD size_t ds_prior_mats_size(int np) { return size_t(2) * np * np + 2 * np; }
// LDS doubles the step kernel needs for (n_pad, Wo)
LIO_HD size_t ds_lds_doubles(int n_pad, int Wo) {
  return size_t(n_pad) * (n_pad + 1) + 10 * size_t(n_pad) + size_t(Wo) * 344 + DS_PART + size_t(n_pad) * DS_NB + 64 + 32;   // (n_pad * 16: L11^-1 of every panel)
}

// ------------------------------------------------------------------------------------------------ executors
struct HostExec {
  static constexpr bool kDevice = false;
  static constexpr int WT = 1;   // lanes per "wave"
  int tid = 0, nthr = 1, lane = 0, wave = 0, nwave = 1;
  void sync() const {}
  void sync_lds() const {}
  void wsync() const {}
  double wsum(double v) const { return v; }
  double wmax(double v) const { return v; }
  double rcp(double d) const { return 1.0 / d; }
  void stamp(long long *, int) const {}
};

// block-wide sum of K per-thread partials (every thread gets the totals); `red` holds >= 8 * K doubles
template <class X, int K>
LIO_HD void block_sum(const X &x, double *red, double (&v)[K]) {
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = x.wsum(v[k]);
  if (x.lane == 0)
#pragma unroll
    for (int k = 0; k < K; ++k) red[x.wave * K + k] = v[k];
  x.sync_lds();
#pragma unroll
  for (int k = 0; k < K; ++k) { double s = 0; for (int w = 0; w < x.nwave; ++w) s += red[w * K + k]; v[k] = s; }
  x.sync_lds();
}
template <class X>
LIO_HD double block_max(const X &x, double *red, double v) {
  v = x.wmax(v);
  if (x.lane == 0) red[x.wave] = v;
  x.sync_lds();
  double s = red[0];
  for (int w = 1; w < x.nwave; ++w) s = s > red[w] ? s : red[w];
  x.sync_lds();
  return s;
}

// ------------------------------------------------------------------------------------------------ aux row of launch A
// ImuFactor i at (pose_i, sb_i, pose_j, sb_j): whitened J (15 x 30, tangent columns [pose_i 6 | sb_i 9 | pose_j 6 | sb_j 9]),
// J^T J, J^T r, 0.5 |r|^2  -> out[DS_IMU_OUT].  lds: >= 15*30*2 + 32 + 4*136 = 1476 doubles.  The summation orders are those of
// WindowSystem::evaluate (host_solver.h) so both paths produce the same blocks.
template <class X>
LIO_HD void aux_imu(const X &x, const DevPim &pm, const double *pose_i, const double *sb_i, const double *pose_j, const double *sb_j,
                    double *out, double *lds, long long *prof = nullptr) {
  if (!pm.present) {
    for (int k = x.tid; k < DS_IMU_OUT; k += x.nthr) out[k] = 0.0;
    return;
  }
  double *Jraw = lds, *Jw = lds + 450, *rraw = lds + 900, *rw = lds + 916;
  PimCore c;
  for (int k = 0; k < 3; ++k) { c.dp[k] = pm.dp[k]; c.dv[k] = pm.dv[k]; c.ba[k] = pm.ba[k]; c.bg[k] = pm.bg[k]; c.g[k] = pm.g[k]; }
  for (int k = 0; k < 4; ++k) c.dq[k] = pm.dq[k];
  c.sum_dt = pm.sum_dt; c.jac = pm.jac;
  // five serial jobs (four raw Jacobian blocks, the residual).  Device: one job per WAVE — on neighbouring lanes of one wave the five
  // different code paths ran one after the other (the wave executes every divergent branch); the residual shares a wave with the
  // lightest block.  Host emulation: one thread walks them.
  for (int job = 0; job < 5; ++job) {
    const int owner = X::kDevice ? (job < 4 ? (x.WT * job) % x.nthr : (x.WT * 3 + x.WT / 2) % x.nthr) : job % x.nthr;
    if (x.tid != owner) continue;
    if (job < 4) {
      // the raw block goes through LDS (x.kDevice: as a thread-private array it is indexed at run time and lives in scratch memory —
      // 56 k of the aux row's 73 k clocks were this phase, profiles/r5_i_*)
      double Jpriv[X::kDevice ? 1 : 135];
      double *J = X::kDevice ? lds + 932 + job * 136 : Jpriv;
      imu_raw_jacobian(c, job, pose_i, sb_i, pose_j, sb_j, J);
      const int cols = (job & 1) ? 9 : 7, use = (job & 1) ? 9 : 6, off = job == 0 ? 0 : (job == 1 ? 6 : (job == 2 ? 15 : 21));
      for (int k = 0; k < 15; ++k)
        for (int q = 0; q < use; ++q) Jraw[k * 30 + off + q] = J[k * cols + q];
    } else {
      V3d Pi, Pj; Qd Qi, Qj;
      unpack_pose(pose_i, Pi, Qi); unpack_pose(pose_j, Pj, Qj);
      pim_residual(c, Pi, Qi, V3d(sb_i[0], sb_i[1], sb_i[2]), V3d(sb_i[3], sb_i[4], sb_i[5]), V3d(sb_i[6], sb_i[7], sb_i[8]), Pj, Qj,
                   V3d(sb_j[0], sb_j[1], sb_j[2]), V3d(sb_j[3], sb_j[4], sb_j[5]), V3d(sb_j[6], sb_j[7], sb_j[8]), rraw);
    }
  }
  x.sync();
  x.stamp(prof, 65);
  const double *S = pm.sqrt_info;
  for (int e = x.tid; e < 465; e += x.nthr) {
    if (e < 450) {
      const int i = e / 30, col = e % 30;
      double o = 0.0;
      for (int k = i; k < 15; ++k) o += S[i * 15 + k] * Jraw[k * 30 + col];
      Jw[e] = o;
    } else {
      const int i = e - 450;
      double s = 0;
      for (int k = i; k < 15; ++k) s += S[i * 15 + k] * rraw[k];
      rw[i] = s;
    }
  }
  x.sync();
  x.stamp(prof, 66);
  for (int e = x.tid; e < 931; e += x.nthr) {
    if (e < 900) {
      const int a = e / 30, b = e % 30;
      double h = 0.0;
      for (int k = 0; k < 15; ++k) h += Jw[k * 30 + a] * Jw[k * 30 + b];
      out[e] = h;
    } else if (e < 930) {
      const int a = e - 900;
      double gsum = 0.0;
      for (int k = 0; k < 15; ++k) gsum += Jw[k * 30 + a] * rw[k];
      out[e] = gsum;
    } else {
      double cost = 0;
      for (int k = 0; k < 15; ++k) cost += rw[k] * rw[k];
      out[930] = 0.5 * cost;
      out[931] = 1.0;
    }
  }
}

// 18x13 linear map of frame i's lidar factors (host_solver.h: lidar_linear_maps) -> out[DS_LMAP_OUT]; lds >= 4*54 + 12 doubles
template <class X>
LIO_HD void aux_lmap(const X &x, const double *pose_p, const double *pose_i, const double *pose_ex, double *out, double *lds) {
  double *J4 = lds, *r4 = lds + 216;
  for (int b = x.tid; b < 4; b += x.nthr) {
    const LidarMapPrep m = lidar_map_prepare(pose_p, pose_i, pose_ex);
    lidar_map_probe(m, b, J4 + b * 54, r4 + b * 3);
  }
  x.sync();
  for (int e = x.tid; e < 247; e += x.nthr) out[e] = lidar_map_entry(J4, r4, e);
}

// marginalization prior at the candidate (WindowSystem::evaluate bit0): out[0..np) = J^T r, out[np] = cost; lds >= 2 np doubles
template <class X>
LIO_HD void aux_prior(const X &x, const DevProblem &pb, const double *mats, const DevParams &P, double *out, double *lds) {
  const int np = pb.n_prior;
  double *dx = lds, *rr = lds + np;
  const double *JtJ = mats, *lin_jac = mats + size_t(np) * np, *lin_res = lin_jac + size_t(np) * np, *Jtr0 = lin_res + np;
  for (int b = x.tid; b < pb.n_keep; b += x.nthr) {
    const int kind = pb.keep_kind[b], idx = pb.keep_index[b], sz = pb.keep_size[b], o = pb.keep_idx[b];
    const double *xv = kind == 0 ? P.pose[idx] : (kind == 1 ? P.sb[idx] : P.ex);
    const double *x0 = pb.prior_x0 + pb.keep_x0[b];
    if (sz != 7) { for (int k = 0; k < sz; ++k) dx[o + k] = xv[k] - x0[k]; }
    else {
      for (int k = 0; k < 3; ++k) dx[o + k] = xv[k] - x0[k];
      Qd q0(x0[6], x0[3], x0[4], x0[5]), q(xv[6], xv[3], xv[4], xv[5]);
      Qd dq = qinverse(q0) * q;
      V3d v = 2.0 * normalized(dq).vec();
      if (dq.w < 0) v = -v;
      for (int k = 0; k < 3; ++k) dx[o + 3 + k] = v[k];
    }
  }
  x.sync();
  for (int e = x.tid; e < 2 * np; e += x.nthr) {
    if (e < np) {
      double s = lin_res[e];
      for (int j = 0; j < np; ++j) s += lin_jac[size_t(e) * np + j] * dx[j];
      rr[e] = s;
    } else {
      const int i = e - np;
      double s = Jtr0[i];
      for (int j = 0; j < np; ++j) s += JtJ[size_t(i) * np + j] * dx[j];
      out[i] = s;
    }
  }
  x.sync();
  if (x.tid == 0) {
    double cost = 0;
    for (int i = 0; i < np; ++i) cost += rr[i] * rr[i];
    out[np] = 0.5 * cost;
  }
}

// extrinsic PriorFactor (WindowSystem::evaluate bit3): out[0..36) J^T J, [36..42) J^T r, [42] cost
template <class X>
LIO_HD void aux_exprior(const X &x, const DevProblem &pb, const DevParams &P, double *out) {
  if (x.tid != 0) return;
  double r[6], J[42];
  prior_factor(V3d(pb.ex_prior_pos[0], pb.ex_prior_pos[1], pb.ex_prior_pos[2]),
               Qd(pb.ex_prior_rot[0], pb.ex_prior_rot[1], pb.ex_prior_rot[2], pb.ex_prior_rot[3]), P.ex, r, J);
  double cost = 0;
  for (int k = 0; k < 6; ++k) cost += r[k] * r[k];
  for (int a = 0; a < 6; ++a) {
    for (int b = 0; b < 6; ++b) { double s = 0; for (int k = 0; k < 6; ++k) s += J[k * 7 + a] * J[k * 7 + b]; out[a * 6 + b] = s; }
    double s = 0; for (int k = 0; k < 6; ++k) s += J[k * 7 + a] * r[k];
    out[36 + a] = s;
  }
  out[42] = 0.5 * cost;
}

// ------------------------------------------------------------------------------------------------ launch B: pieces
struct StepBuffers {
  const double *prior_mats;    // ds_prior_mats_size(np)
  const double *partials;      // Wo * bpf * LIO_MOMENT_OUT
  const double *imu_out;       // Wo * DS_IMU_OUT
  const double *lmap;          // Wo * DS_LMAP_OUT
  const double *prior_out;     // np + 1
  const double *exprior_out;   // DS_EXP_OUT
  double *Hcur;                // n_pad * ld: scaled H at the accepted point
  double *S_buf;               // 2 * Wo * LIO_MOMENT_OUT: moments at the accepted point / at the candidate (st.s_cur picks)
  long long *prof;             // optional: 32 shader-clock stamps of the last launch B (LIO_DEBUG_TIMING)
};

// ---- one window of a batch (device array; est_batch.hip fills it): everything the two launches of an iteration and the
// marginalization launches read for that window
struct DevMarg {
  DevParams x;                       // the parameters after DoubleToVector / VectorToDouble: the point the prior is linearised at
  int active;                        // 0: this window does not marginalise this time (turn_off, no marginalization_factor, host fallback)
  int Wo, m, n;                      // dropped (pose 0 [+ speed-bias 0]) and kept tangent sizes (MarginalizationFactor.cc:185-311)
  int has_imu, have_prior;
  int pose_col[DS_MAX_WO + 1], sb_col[2], ex_col;   // columns of the marginalization's layout (-1: block absent)
  int prior_col[DS_MAX_NPAD];        // column of [dropped | kept] -> column of the OLD prior (-1: none)
};
struct BatchSolve {
  int active;                        // 0: the window is not solved on the device this time
  int nframes, bpf;
  MomentFrame fr[DS_MAX_WO];         // slot_off: offset into the batch's slot arrays; R, t are overwritten from the state
  const DevProblem *pb;
  DevState *st;
  const double *prior_mats;          // the window's prior on the device (ds_prior_mats_size(np) doubles)
  double *partials, *imu_out, *lmap, *prior_out, *exprior_out, *Hcur, *S_buf;
  long long *prof;
  // marginalization (launch_bw_marginalize, marg_kernels.hip)
  const DevMarg *marg;
  double *marg_imu, *marg_lmap, *marg_prior_out, *marg_A, *marg_info;
  double *next_prior_mats;           // where the new prior goes (same layout as prior_mats)
};

// The batch's allocations, handed to the batched kernels BY VALUE next to the descriptors: pointers that arrive as kernel arguments are
// global to the compiler, pointers read from a descriptor are not (dev.h: rebase).
struct BatchBases { double *slab; double *partials; DevState *st; DevProblem *pb; DevMarg *mg; };

struct StepLds {
  double *A, *hdiag, *gz, *invd, *scale, *diag, *grad, *gn, *g, *step, *tmp, *zb, *part, *xinv, *red;
  int *ctl;
};
LIO_HD StepLds carve_lds(double *base, int n_pad, int Wo) {
  StepLds l;
  double *p = base;
  l.A = p; p += size_t(n_pad) * (n_pad + 1);
  l.hdiag = p; p += n_pad; l.gz = p; p += n_pad; l.invd = p; p += n_pad; l.scale = p; p += n_pad; l.diag = p; p += n_pad;
  l.grad = p; p += n_pad; l.gn = p; p += n_pad; l.g = p; p += n_pad; l.step = p; p += n_pad; l.tmp = p; p += n_pad;
  l.zb = p; p += size_t(Wo) * 344;
  l.part = p; p += DS_PART;
  l.xinv = p; p += size_t(n_pad) * DS_NB;
  l.red = p; p += 64;
  l.ctl = reinterpret_cast<int *>(p);
  return l;
}

// ---- control block shared through LDS (thread 0 writes, everybody reads after a barrier)
struct StepCtl {
  double radius, mu, alpha, dogleg_norm, gmax, x_cost, x_norm, model_change, step_norm, cand_total;
  int mode, done, a_valid, fact_ok, reuse, invalid, it, successful, termination, valid_step, lin_ok, lower_fresh;   // lower_fresh: A's lower triangle mirrors the upper
};
enum { DS_MODE_INIT = 0, DS_MODE_ACCEPT = 1, DS_MODE_REJECT = 2 };

// H(i, j) of a symmetric matrix whose UPPER triangle (row-major, leading dimension ld) and diagonal vector are intact
LIO_HD double ds_hsym(const double *Hm, int ld, const double *hd, int i, int j) { return i == j ? hd[i] : (i < j ? Hm[i * ld + j] : Hm[j * ld + i]); }

// sum_i v_i (H v)_i for the symmetric H of ds_hsym, rows >= n excluded.  Device: four lanes per row (columns q, q + 4, ...; the
// four partial sums combined as (p0 + p1) + (p2 + p3)) — one lane per row was a chain of n dependent multiply-adds on 96 of the
// block's 512 threads, 12 us of the step twice per iteration (profiles/r5_e_step_phases.txt).  The thread's contribution is
// ADDED to acc (the caller block-sums it).
template <class X>
LIO_HD void ds_vHv(const X &x, const double *Hm, int ld, const double *hd, int n, const double *v, double &acc) {
  if constexpr (X::kDevice) {
    const int q = x.tid & 3;
    for (int i0 = 0; i0 < n; i0 += x.nthr / 4) {   // (uniform trip count: the quad permutes below need every lane of the wave)
      const int i = i0 + (x.tid >> 2);
      const int ic = i < n ? i : 0;
      const double hii = hd[ic];
      double sres = 0.0;
      for (int j0 = q; j0 < n; j0 += 32) {          // eight terms' loads in flight (unconditional, clamped), then the eight multiply-adds in order
        double hv[8], vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = j0 + 4 * u, jc = j < n ? j : ic;
          hv[u] = Hm[ic < jc ? ic * ld + jc : jc * ld + ic];
          vv[u] = v[jc];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = j0 + 4 * u;
          const double h = (j == ic) ? hii : hv[u];
          sres += (j < n) ? h * vv[u] : 0.0;
        }
      }
      sres = x.pair_sum4(sres);
      if (i < n && q == 0) acc += v[i] * sres;
    }
  } else {
    for (int i = x.tid; i < n; i += x.nthr) {
      double sres = 0;
      for (int j = 0; j < n; ++j) sres += ds_hsym(Hm, ld, hd, i, j) * v[j];
      acc += v[i] * sres;
    }
  }
}

// ------------------------------------------------------------------------------------------------ blocked L D L^T in LDS
// Lower triangle of A (n_pad x n_pad, leading dimension ld; rows/cols >= n are an identity pad) is replaced by unit L with D
// on the diagonal; the right-hand side gz is carried along as an extra row, so that it leaves as z = D^-1 L^-1 g; the
// back-substitution L^T x = z then overwrites gz with the solution.  The STRICT UPPER triangle is never touched (it still
// holds the matrix, which the caller needs for H v products afterwards).  Returns 0 when a pivot is not positive.

// 16x16 diagonal block at p.  Reference form: one lane walks the block; the device executor keeps row r of the block in the
// registers of lane r and broadcasts pivots / columns with v_readlane (same arithmetic, same order).
template <class X>
LIO_HD int ds_panel_factor(const X &x, double *A, int ld, int p, double *invd, double *scratch, double *xinv) {
  if (x.wave != 0) return 1;
  if constexpr (X::kDevice) {
    return x.panel_factor_regs(A, ld, p, invd, scratch, xinv + size_t(p) * DS_NB);
  } else {
    for (int j = 0; j < DS_NB; ++j) {
      const double d = A[(p + j) * ld + p + j];
      if (!(d > 0.0)) return 0;
      const double inv = x.rcp(d);
      invd[p + j] = inv;
      for (int r = j + 1; r < DS_NB; ++r) {
        const double l = A[(p + r) * ld + p + j] * inv;
        for (int c = j + 1; c <= r; ++c) A[(p + r) * ld + p + c] -= l * A[(p + c) * ld + p + j];   // column j still holds L d
      }
      for (int r = j + 1; r < DS_NB; ++r) A[(p + r) * ld + p + j] *= inv;
    }
    return 1;
  }
}

// (not inlined into the step kernel: inside that one function — 256 VGPRs and spills — the factorisation's unrolled panel code
// shared its registers with everything else that is live across it)
#if defined(__HIP_DEVICE_COMPILE__) && defined(LIO_DS_LDLT_NOINLINE)
#define LIO_DS_NOINLINE __attribute__((noinline))
#else
#define LIO_DS_NOINLINE
#endif
template <class X>
LIO_DS_NOINLINE LIO_HD int ds_ldlt_solve(const X &x, double *A, int ld, int npad, double *gz, double *invd, double *part, double *xinv, int *flag, long long *prof = nullptr) {
  const int nblk = npad / DS_NB;
  for (int kb = 0; kb < nblk; ++kb) {
    const int p = kb * DS_NB, q = p + DS_NB;
    // ---- phase 1: diagonal block (wave 0)
    if (kb == 0) x.stamp(prof, 16);
    const int ok = ds_panel_factor(x, A, ld, p, invd, part, xinv);
    if (x.tid == 0) *flag = ok;
    x.sync_lds();
    if (kb == 0) x.stamp(prof, 17);
    if (!*flag) return 0;   // uniform: every thread reads the same LDS word after the barrier
    // ---- phase 2: rows below the block and the right-hand side: T = A21 L11^-T (forward substitution along the columns),
    // stored as L = T D^-1
    const int nrows = npad - q;
    if constexpr (X::kDevice) {
      x.panel_trsm_mfma(A, ld, npad, p, gz, invd, xinv + size_t(p) * DS_NB);   // against L11^-1, which the panel wave left in its slot of xinv
    } else {
      for (int t = x.tid; t <= nrows; t += x.nthr) {
        const bool rhs = (t == nrows);
        double *row = rhs ? (gz + p) : (A + (q + t) * ld + p);
        double tr[DS_NB];
        for (int j = 0; j < DS_NB; ++j) tr[j] = row[j];
        for (int j = 1; j < DS_NB; ++j) {
          const double *l11 = A + (p + j) * ld + p;
          for (int k = 0; k < j; ++k) tr[j] -= tr[k] * l11[k];
        }
        for (int j = 0; j < DS_NB; ++j) row[j] = tr[j] * invd[p + j];
      }
    }
    x.sync_lds();
    if (kb == 0) x.stamp(prof, 18);
    if (q >= npad) break;
    // ---- phase 3: trailing update  A22 -= L21 D L21^T  (lower triangle), rhs -= L21 D z
    for (int c = q + x.tid; c < npad; c += x.nthr) {
      double sres = 0;
      const double *lc = A + (c) * ld + p;
#pragma unroll
      for (int j = 0; j < DS_NB; ++j) sres += gz[p + j] * (lc[j] * A[(p + j) * ld + p + j]);
      gz[c] -= sres;
    }
    if constexpr (X::kDevice) {
      x.trailing_update_mfma(A, ld, npad, p);
    } else {
      for (int r = q; r < npad; ++r)
        for (int c = q; c <= r; ++c) {
          double sres = 0;
          for (int j = 0; j < DS_NB; ++j) sres += A[(r) * ld + p + j] * (A[(c) * ld + p + j] * A[(p + j) * ld + p + j]);
          A[(r) * ld + c] -= sres;
        }
    }
    x.sync_lds();
    if (kb == 0) x.stamp(prof, 19);
  }
  x.stamp(prof, 20);
  // ---- back-substitution  L^T x = z, block by block from the bottom
  for (int kb = nblk - 1; kb >= 0; --kb) {
    const int p = kb * DS_NB, q = p + DS_NB;
    // contributions of the rows below: z_j -= sum_{r >= q} L[r][p + j] x_r ; 32 slices of rows per column, fixed order
    {
      const int j = x.tid & 15;
      for (int sl = x.tid >> 4; sl < 32; sl += (x.nthr + 15) / 16) {
        double acc = 0;
        for (int r = q + sl; r < npad; r += 32) acc += A[(r) * ld + p + j] * gz[r];
        part[sl * 16 + j] = acc;
        if (x.nthr < 16) {   // host emulation: one thread walks every column
          for (int jj = 1; jj < 16; ++jj) {
            double a2 = 0;
            for (int r = q + sl; r < npad; r += 32) a2 += A[(r) * ld + p + jj] * gz[r];
            part[sl * 16 + jj] = a2;
          }
        }
      }
    }
    x.sync_lds();
    if (x.wave == 0) {
      if constexpr (X::kDevice) {
        x.panel_backsolve_regs(A, ld, p, gz, part, xinv + size_t(p) * DS_NB);
      } else {
        double y[DS_NB];
        for (int j = 0; j < DS_NB; ++j) { double s2 = 0; for (int sl = 0; sl < 32; ++sl) s2 += part[sl * 16 + j]; y[j] = gz[p + j] - s2; }
        for (int k = DS_NB - 1; k >= 1; --k)
          for (int j = 0; j < k; ++j) y[j] -= A[(p + k) * ld + p + j] * y[k];
        for (int j = 0; j < DS_NB; ++j) gz[p + j] = y[j];
      }
    }
    x.sync_lds();
  }
  return 1;
}

// Frame blocks of the normal equations from the folded moments: LS = L S (18 x 13), then H_i = (L S) L^T (18 x 18) and
// g_i = (L S) l into zb[f * 344 + ..] (324 + 18 used).  S: Wo x LIO_MOMENT_OUT (row-major 16 x 16 tiles), lmap: Wo x DS_LMAP_OUT,
// LS: Wo x 234 doubles of scratch.  Shared by the step kernel and the marginalization's assembly.
template <class X>
LIO_HD void ds_lidar_blocks(const X &x, int Wo, const double *lmap, const double *S, double *LS, double *zb, long long *prof = nullptr) {
  for (int e = x.tid; e < Wo * 234; e += x.nthr) {
    const int f = e / 234, a = (e % 234) / 13, b = e % 13;
    const double *Lm = lmap + size_t(f) * DS_LMAP_OUT, *Sf = S + f * LIO_MOMENT_OUT;
    double o = 0.0;
    for (int k = 0; k < 13; ++k) o += Lm[a * 13 + k] * Sf[k * 16 + b];
    LS[e] = o;
  }
  x.sync_lds();
  x.stamp(prof, 2);
  for (int e = x.tid; e < Wo * 342; e += x.nthr) {
    const int f = e / 342, r = e % 342;
    const double *Lm = lmap + size_t(f) * DS_LMAP_OUT, *ls = LS + f * 234;
    double o = 0.0;
    if (r < 324) {
      const int a = r / 18, b = r % 18;
      for (int k = 0; k < 13; ++k) o += ls[a * 13 + k] * Lm[b * 13 + k];
    } else {
      const int a = r - 324;
      for (int k = 0; k < 13; ++k) o += ls[a * 13 + k] * Lm[234 + k];
    }
    zb[f * 344 + r] = o;
  }
  x.sync_lds();
  x.stamp(prof, 3);
}

// ------------------------------------------------------------------------------------------------ launch B
template <class X>
LIO_HD void solve_step(const X &x, const DevProblem &pb, DevState &st, const StepBuffers &B, double *lds_base) {
  const int n = pb.n, npad = pb.n_pad, ld = pb.ld, Wo = pb.Wo, np = pb.n_prior, exc = pb.ex_col;
  StepLds L = carve_lds(lds_base, npad, Wo);
  StepCtl &C = *reinterpret_cast<StepCtl *>(L.ctl);
  if (x.tid == 0) C.done = st.done;
  x.sync_lds();
  if (C.done) return;
  double *A = L.A;
  x.stamp(B.prof, 0);
  // ---- P1: fold the per-block partials of launch A (the order of k_moment_reduce: four interleaved chains, remainder on the
  // first, (v0 + v1) + (v2 + v3)); the folded moments also go to the candidate slot of S_buf
  double *Sx = A;                                   // Wo x 260, dead once the assembly starts
  double *LS = A + size_t(Wo) * LIO_MOMENT_OUT;     // Wo x 234
  double *S_cand = B.S_buf + size_t(1 - st.s_cur) * Wo * LIO_MOMENT_OUT;
  {
    const int bpf = pb.bpf, b4 = bpf & ~3;
    const int items = Wo * 258;
    // one lane per value, eight blocks' loads in flight per value and a thread's (up to three) values together (the first device form —
    // four lanes per value, one load at a time — was a chain of memory round trips: 14 us of the step at ten blocks per frame; one value
    // at a time still cost three round trips: 5 us); the sums and their order are the four chains above
    constexpr int IT = X::kDevice ? 3 : 1;
    for (int item0 = x.tid; item0 < items; item0 += IT * x.nthr) {
      const double *src[IT];
      double v[IT][4];
#pragma unroll
      for (int u = 0; u < IT; ++u) {
        const int item = item0 + u * x.nthr;
        const int it2 = item < items ? item : item0;
        src[u] = B.partials + size_t(it2 / 258) * bpf * LIO_MOMENT_OUT + it2 % 258;
        v[u][0] = v[u][1] = v[u][2] = v[u][3] = 0.0;
      }
      int b = 0;
      for (; b + 8 <= b4; b += 8) {
        double t[IT][8];
#pragma unroll
        for (int u = 0; u < IT; ++u)
#pragma unroll
          for (int q = 0; q < 8; ++q) t[u][q] = src[u][size_t(b + q) * LIO_MOMENT_OUT];
#pragma unroll
        for (int u = 0; u < IT; ++u)
#pragma unroll
          for (int q = 0; q < 8; ++q) v[u][q & 3] += t[u][q];
      }
      if (b < b4) {   // (b4 is a multiple of four: one more group of four)
        double t[IT][4];
#pragma unroll
        for (int u = 0; u < IT; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q) t[u][q] = src[u][size_t(b + q) * LIO_MOMENT_OUT];
#pragma unroll
        for (int u = 0; u < IT; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q) v[u][q] += t[u][q];
      }
      for (b = b4; b < bpf; ++b)
#pragma unroll
        for (int u = 0; u < IT; ++u) v[u][0] += src[u][size_t(b) * LIO_MOMENT_OUT];
#pragma unroll
      for (int u = 0; u < IT; ++u) {
        const int item = item0 + u * x.nthr;
        if (item < items) {
          const int f = item / 258, k = item % 258;
          const double o = (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
          Sx[f * LIO_MOMENT_OUT + k] = o; S_cand[f * LIO_MOMENT_OUT + k] = o;
          if (k >= 256) L.tmp[2 * f + (k - 256)] = o;   // cost and count of frame f, for the passes and the decision below (Sx is gone by then)
        }
      }
    }
  }
  x.sync_lds();
  x.stamp(B.prof, 1);
  // ---- P2, P3: H_i = (L S) L^T (18 x 18), g_i = (L S) l per frame
  ds_lidar_blocks(x, Wo, B.lmap, Sx, LS, L.zb, B.prof);
  // ---- P4: assemble the candidate's H (full, unscaled) and g in the order of WindowSystem::evaluate — prior, ImuFactor 0..Wo-1,
  // lidar frames 1..Wo, extrinsic prior — as BLOCK passes with a barrier between them: every entry still receives its
  // contributions in that order, but a pass is a plain coalesced add of one factor's block (the first form visited every entry
  // once and walked all factors from there: 18 entries per thread, each a chain of dependent global loads — 50 of the step's
  // 155 us, profiles/r5_c_step_phases.txt).  Thread layout of the dense passes: 128 columns x (threads / 128) rows, no division.
  const double *JtJ = B.prior_mats;
  constexpr int CW = X::kDevice ? 128 : 1;
  const int tcol = x.tid % CW, trow = x.tid / CW, rstep = x.nthr >= CW ? x.nthr / CW : 1;
  int *pcol = reinterpret_cast<int *>(L.part);   // the prior's column of every tangent column (LDS copy; `part` is idle until the back-substitution)
  for (int i = x.tid; i < npad; i += x.nthr) pcol[i] = (pb.have_prior && i < n) ? pb.prior_col[i] : -1;
  x.sync_lds();
  x.stamp(B.prof, 15);
  // the ImuFactor blocks of this thread's entries, requested now (one memory round trip under the prior pass instead of one per block)
  constexpr int IMU_Q = X::kDevice ? (930 + DS_THREADS - 1) / DS_THREADS : 1;
  double imv[X::kDevice ? DS_MAX_WO : 1][IMU_Q];
  double imflag[X::kDevice ? DS_MAX_WO : 1];
  if (X::kDevice) {
#pragma unroll
    for (int i = 0; i < DS_MAX_WO; ++i) {
      const double *im = B.imu_out + size_t(i < Wo ? i : 0) * DS_IMU_OUT;
      imflag[i] = im[931];
#pragma unroll
      for (int q = 0; q < IMU_Q; ++q) { const int e = x.tid + q * DS_THREADS; imv[i][q] = im[e < 930 ? e : 0]; }
    }
  }
  x.stamp(B.prof, 24);
  {  // prior pass: column tcol of rows trow, trow + rstep, ...: the rows' prior columns first (one batch of LDS reads), then 24 rows' loads
     // in flight, unconditional from clamped 32-bit indices (profiles/r5_i_*, r5_j_*: with a branch and an LDS read in front of every
     // load the pass cost 28 k clocks)
    constexpr int PQ = X::kDevice ? 24 : 8;
    for (int c = tcol; c < npad; c += CW) {
      const int pc = pcol[c];
      for (int r0 = trow; r0 < npad; r0 += PQ * rstep) {
        int pr[PQ];
#pragma unroll
        for (int q = 0; q < PQ; ++q) { const int r = r0 + q * rstep; pr[q] = pcol[r < npad ? r : 0]; }
        double v[PQ];
#pragma unroll
        for (int q = 0; q < PQ; ++q) {
          const bool on = (pr[q] >= 0) & (pc >= 0) & (r0 + q * rstep < npad);
          const double t = JtJ[on ? pr[q] * np + pc : 0];
          v[q] = on ? t : 0.0;
        }
#pragma unroll
        for (int q = 0; q < PQ; ++q) {
          const int r = r0 + q * rstep;
          if (r < npad) A[r * ld + c] = (r == c && r >= n) ? 1.0 : v[q];   // (rows / columns of the pad have no prior column: zero, identity on the diagonal)
        }
      }
    }
    for (int r = x.tid; r < npad; r += x.nthr) { const int pr = pcol[r]; const double t = B.prior_out[pr >= 0 ? pr : 0]; L.gz[r] = pr >= 0 ? t : 0.0; }
  }
  x.stamp(B.prof, 25);
  x.sync_lds();
  x.stamp(B.prof, 11);
  // ImuFactor i spans tangent columns [15 i, 15 i + 30): the even factors touch disjoint entries, and so do the odd ones — two passes
  // (an entry shared by factors i and i + 1 receives the even one first)
  for (int parity = 0; parity < 2; ++parity) {
    if (X::kDevice) {
      // (static register indices: the loop over the frames is unrolled to its bound)
#pragma unroll
      for (int ii = 0; ii < DS_MAX_WO; ++ii) {
        if ((ii & 1) == parity && ii < Wo && imflag[ii] != 0.0) {
#pragma unroll
          for (int q = 0; q < IMU_Q; ++q) {
            const int e = x.tid + q * DS_THREADS;
            if (e < 900) { const int lr = e / 30, lc = e % 30; A[(15 * ii + lr) * ld + 15 * ii + lc] += imv[ii][q]; }
            else if (e < 930) L.gz[15 * ii + (e - 900)] += imv[ii][q];
          }
        }
      }
    } else {
      for (int i = parity; i < Wo; i += 2) {
        const double *im = B.imu_out + size_t(i) * DS_IMU_OUT;
        if (im[931] == 0.0) continue;
        for (int e = x.tid; e < 930; e += x.nthr) {
          if (e < 900) { const int lr = e / 30, lc = e % 30; A[(15 * i + lr) * ld + 15 * i + lc] += im[e]; }
          else L.gz[15 * i + (e - 900)] += im[e];
        }
      }
    }
    x.sync_lds();
  }
  x.stamp(B.prof, 14);
  // lidar frame f + 1 touches (pose_0, pose_{f+1}, extrinsic): local rows 0..5, 6..11, 12..17.  ONE pass: the entries that involve
  // pose_{f+1} belong to that frame alone; the 12 x 12 (+ 12 of g) entries among pose_0 and the extrinsic are shared by all frames —
  // one thread each walks the frames in order and adds the extrinsic PriorFactor behind them.
  {
    auto glob = [&](int f, int a) { return a < 6 ? a : (a < 12 ? 15 * (f + 1) + (a - 6) : (exc >= 0 ? exc + (a - 12) : -1)); };
    for (int item = x.tid; item < Wo * 342; item += x.nthr) {
      const int f = item / 342, e = item % 342;
      if (L.tmp[2 * f + 1] == 0.0) continue;
      const double val = L.zb[f * 344 + e];
      if (e < 324) {
        const int a = e / 18, b = e % 18;
        if ((a < 6 || a >= 12) && (b < 6 || b >= 12)) continue;          // shared: below
        const int ra = glob(f, a), cb = glob(f, b);
        if (ra >= 0 && cb >= 0) A[ra * ld + cb] += val;
      } else {
        const int a = e - 324;
        if (a < 6 || a >= 12) continue;
        L.gz[glob(f, a)] += val;
      }
    }
    const bool exp_on = pb.use_ex_prior && exc >= 0;
    for (int item = x.tid; item < 156; item += x.nthr) {
      const bool is_g = item >= 144;
      const int sa = is_g ? item - 144 : item / 12, sb = is_g ? 0 : item % 12;       // indices among (pose_0 | extrinsic)
      const int a = sa < 6 ? sa : sa + 6, b = sb < 6 ? sb : sb + 6;                   // local indices of the 18
      const int ra = glob(0, a), cb = glob(0, b);
      if (ra < 0 || cb < 0) continue;
      double acc = is_g ? L.gz[ra] : A[ra * ld + cb];
      for (int f = 0; f < Wo; ++f)
        if (L.tmp[2 * f + 1] != 0.0) acc += L.zb[f * 344 + (is_g ? 324 + a : a * 18 + b)];
      if (exp_on && sa >= 6 && (is_g || sb >= 6)) acc += B.exprior_out[is_g ? 36 + (sa - 6) : (sa - 6) * 6 + (sb - 6)];
      if (is_g) L.gz[ra] = acc; else A[ra * ld + cb] = acc;
    }
    x.sync_lds();
  }
  // NOTE: Sx / LS aliased A and are gone now; S_cand (global) keeps the moments.
  x.stamp(B.prof, 4);
  // ---- P5: decide on the pending candidate (thread 0)
  if (x.tid == 0) {
    double marg = pb.have_prior ? B.prior_out[np] : 0.0, pim = 0.0, ppp = 0.0, cnt = 0.0;
    for (int i = 0; i < Wo; ++i) if (B.imu_out[size_t(i) * DS_IMU_OUT + 931] != 0.0) pim += B.imu_out[size_t(i) * DS_IMU_OUT + 930];
    for (int i = 0; i < Wo; ++i) { ppp += L.tmp[2 * i]; cnt += L.tmp[2 * i + 1]; }
    const double exprior = pb.use_ex_prior ? B.exprior_out[42] : 0.0;
    const double total = marg + pim + ppp + exprior;
    C.radius = st.radius; C.mu = st.mu; C.alpha = st.alpha; C.dogleg_norm = st.dogleg_norm; C.gmax = st.gmax; C.x_cost = st.x_cost;
    C.x_norm = st.x_norm; C.model_change = st.model_change; C.step_norm = st.step_norm; C.cand_total = total;
    C.reuse = st.reuse; C.invalid = st.invalid; C.it = st.it; C.successful = st.successful; C.termination = st.termination;
    C.a_valid = 0; C.fact_ok = 1; C.valid_step = 0; C.lin_ok = 1; C.lower_fresh = 0;
    if (!st.started) {
      st.costs0[0] = marg; st.costs0[1] = pim; st.costs0[2] = ppp; st.costs0[3] = exprior;
      st.n_lidar = cnt;
      // group costs -> convergence_flag_ (Estimator.cc:1956-1984)
      const int turn_off = pb.imu_on ? (pim > 1e3) : 1;
      const double ratio = marg / (ppp + pim);
      int conv = pb.conv_flag_in;
      if (!conv && !turn_off && ratio <= 2 && ratio != 0) conv = 1;
      st.turn_off = turn_off; st.conv_flag_out = conv;
      if (!conv && (exc >= 0 || pb.have_prior)) { st.need_host = 1; st.done = 1; C.done = 1; }   // the problem changes shape: host path
      C.mode = DS_MODE_INIT;
      C.x_cost = total; st.trace[0] = total; st.ntrace = 1;
      C.radius = 1e4; C.mu = 1e-8; C.alpha = 0; C.dogleg_norm = 0; C.reuse = 0; C.invalid = 0; C.it = 0; C.successful = 0; C.termination = 0;
      st.started = 1;
    } else {
      st.n_lidar = cnt;
      bool fin = false;
      if (C.step_norm <= 1e-8 * (C.x_norm + 1e-8)) { C.termination = 1; fin = true; }
      const double cost_change = C.x_cost - total;
      if (!fin && fabs(cost_change) <= 1e-6 * C.x_cost) { C.termination = 2; fin = true; }
      if (fin) {
        if (st.ntrace < 40) st.trace[st.ntrace++] = C.x_cost;
        st.termination = C.termination; st.done = 1; C.done = 1;
      } else {
        const double rho = cost_change / C.model_change;
        if (rho > 1e-3) {
          C.mode = DS_MODE_ACCEPT;
          C.x_cost = total;
          ++C.successful;
          if (rho < 0.25) C.radius *= 0.5;
          if (rho > 0.75) C.radius = C.radius > 3.0 * C.dogleg_norm ? C.radius : 3.0 * C.dogleg_norm;
          C.mu = (2.0 * C.mu / 10.0) > 1e-8 ? (2.0 * C.mu / 10.0) : 1e-8;
          C.reuse = 0;
          st.s_cur = 1 - st.s_cur;   // the candidate's moments become the accepted point's
        } else {
          C.mode = DS_MODE_REJECT;
          C.radius *= 0.5; C.reuse = 1;
        }
        if (st.ntrace < 40) st.trace[st.ntrace++] = C.x_cost;
      }
    }
  }
  x.sync_lds();
  if (C.done) { if (x.tid == 0) { st.it = C.it; st.successful = C.successful; } return; }
  const int mode = C.mode;
  x.stamp(B.prof, 5);
  // ---- P6: make (H, g, x) the current point
  DevParams &P = st.x;
  if (mode == DS_MODE_ACCEPT) {
    const double *src = reinterpret_cast<const double *>(&st.cand);
    double *dst = reinterpret_cast<double *>(&st.x);
    for (int k = x.tid; k < int(sizeof(DevParams) / sizeof(double)); k += x.nthr) dst[k] = src[k];
    x.sync();   // (st.x is read back from global memory by other threads)
  }
  if (mode != DS_MODE_REJECT) {
    // gradient max-norm through the ambient Plus (TrustRegionMinimizer: || Plus(x, -g) - x ||_inf) and |x|
    double mx = 0.0, sq = 0.0;
    for (int t = x.tid; t < (Wo + 1) * 10 + 1; t += x.nthr) {
      const int f = t / 10, w = t % 10;
      if (f <= Wo) {
        if (w == 0) {
          double negd[6], out[7];
          for (int k = 0; k < 6; ++k) negd[k] = -L.gz[15 * f + k];
          pose_plus(P.pose[f], negd, out);
          for (int k = 0; k < 7; ++k) { const double d = P.pose[f][k] - out[k]; mx = fabs(d) > mx ? fabs(d) : mx; sq += P.pose[f][k] * P.pose[f][k]; }
        } else {
          const int k = w - 1;
          const double pp = P.sb[f][k] + (-L.gz[15 * f + 6 + k]);
          const double d = P.sb[f][k] - pp;
          mx = fabs(d) > mx ? fabs(d) : mx; sq += P.sb[f][k] * P.sb[f][k];
        }
      } else if (exc >= 0) {
        double negd[6], out[7];
        for (int k = 0; k < 6; ++k) negd[k] = -L.gz[exc + k];
        pose_plus(P.ex, negd, out);
        for (int k = 0; k < 7; ++k) { const double d = P.ex[k] - out[k]; mx = fabs(d) > mx ? fabs(d) : mx; sq += P.ex[k] * P.ex[k]; }
      }
    }
    const double gmax = block_max(x, L.red, mx);
    double sv[1] = {sq};
    block_sum<X, 1>(x, L.red, sv);
    if (x.tid == 0) { C.gmax = gmax; C.x_norm = sqrt(sv[0]); }
    x.stamp(B.prof, 22);
    // Jacobi scaling: fixed at the first linearisation (scale = 1 / (1 + sqrt(H_ii)))
    for (int i = x.tid; i < npad; i += x.nthr) {
      double sc = 1.0;
      if (i < n) sc = (mode == DS_MODE_INIT) ? 1.0 / (1.0 + sqrt(A[(i) * ld + i])) : st.scale[i];
      L.scale[i] = sc;
      if (mode == DS_MODE_INIT && i < n) st.scale[i] = sc;
    }
    x.sync_lds();
    x.stamp(B.prof, 23);
    // the UPPER triangle is scaled and mirrored into the lower one in the same pass (the factorisation below then only has to set the
    // diagonal); eight entries in flight; only the upper triangle goes to global memory (ds_hsym reads nothing else)
    for (int c = tcol; c < n; c += CW) {
      const double sc_c = L.scale[c];
      for (int r0 = trow; r0 <= c; r0 += 8 * rstep) {
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int r = r0 + q * rstep, rc = r <= c ? r : c; v[q] = A[rc * ld + c] * (L.scale[rc] * sc_c); }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int r = r0 + q * rstep;
          if (r <= c) {
            A[r * ld + c] = v[q];
            A[c * ld + r] = v[q];
            B.Hcur[r * ld + c] = v[q];
            if (r == c) L.hdiag[r] = v[q];
          }
        }
      }
    }
    for (int i = x.tid; i < npad; i += x.nthr) {
      const double gs = i < n ? L.gz[i] * L.scale[i] : 0.0;
      L.g[i] = gs;
      if (i < n) st.g[i] = gs; else L.hdiag[i] = 1.0;
    }
    if (x.tid == 0) { C.a_valid = 1; C.lower_fresh = 1; }
  } else {
    for (int i = x.tid; i < npad; i += x.nthr) {
      const bool in = i < n;
      L.scale[i] = in ? st.scale[i] : 1.0; L.g[i] = in ? st.g[i] : 0.0; L.diag[i] = in ? st.diag[i] : 1.0;
      L.grad[i] = in ? st.grad[i] : 0.0; L.gn[i] = in ? st.gn[i] : 0.0;
      L.hdiag[i] = in ? B.Hcur[(i) * ld + i] : 1.0;
    }
  }
  x.sync_lds();
  x.stamp(B.prof, 6);
  // ---- P7: the minimizer loop up to the next candidate (TrustRegionMinimizer::Minimize + DoglegStrategy::ComputeStep)
  for (int guard = 0; guard < 64; ++guard) {
    // Thread 0 owns the control block; every other thread only reads it, and between such a read and thread 0's next write of the
    // same word lies a barrier.  (Round 6: two places broke that rule — C.reuse was set right behind the test below, C.done behind the
    // test at the loop's end — and a wave delayed by blocks of another stream sharing its SIMD read the NEW value, took the other branch
    // and met the workgroup's barriers out of step: windows diverged at random once two loop groups ran side by side.)
    if (x.tid == 0 && !C.done) {   // (done: five invalid steps in a row, set at the end of the previous pass)
      if (C.it >= pb.max_iterations) { C.termination = 0; C.done = 1; }
      else if (C.gmax <= 1e-10) { C.termination = 3; C.done = 1; }
      else if (C.radius <= 1e-32) { C.termination = 1; C.done = 1; }
      else ++C.it;
    }
    x.sync_lds();
    if (C.done) break;
    bool h_in_lds = C.a_valid != 0;   // H of the current point: in LDS, or only in global memory (LDS holds a rejected candidate's)
    if (!C.reuse) {
      if (!h_in_lds) {   // a linearisation is needed at the accepted point but LDS holds a rejected candidate: reload
        for (int r = trow; r < npad; r += rstep)
          for (int c = tcol; c < npad; c += CW) {
            if (r < n && c < n) { if (c >= r) A[(r) * ld + c] = B.Hcur[(r) * ld + c]; }   // (the lower triangle is rebuilt from the upper before the factorisation)
            else A[(r) * ld + c] = (r == c) ? 1.0 : 0.0;
          }
        x.sync_lds();   // (every thread has read a_valid above before it changes)
        if (x.tid == 0) C.a_valid = 1;
        h_in_lds = true;
      }
      // diag, gradient in the scaled space, Cauchy step length alpha = |g|^2 / |J g|^2
      double gq[2] = {0.0, 0.0};
      for (int i = x.tid; i < npad; i += x.nthr) {
        double dg = 1.0, gr = 0.0, tt = 0.0;
        if (i < n) {
          double h = L.hdiag[i];
          h = h > 1e-6 ? h : 1e-6; h = h < 1e32 ? h : 1e32;
          dg = sqrt(h); gr = L.g[i] / dg; tt = gr / dg;
          gq[0] += gr * gr;
          st.diag[i] = dg; st.grad[i] = gr;
        }
        L.diag[i] = dg; L.grad[i] = gr; L.tmp[i] = tt;
      }
      x.sync_lds();
      if (x.tid == 0) { C.reuse = 1; C.lin_ok = 0; }   // (behind the barrier: every thread has tested C.reuse; lin_ok is read behind the next one)
      if (h_in_lds) ds_vHv(x, A, ld, L.hdiag, n, L.tmp, gq[1]); else ds_vHv(x, B.Hcur, ld, L.hdiag, n, L.tmp, gq[1]);   // (two calls: an LDS pointer and a global one, never a flat access)
      block_sum<X, 2>(x, L.red, gq);
      if (x.tid == 0) C.alpha = gq[0] / gq[1];
      x.stamp(B.prof, 7);
      // Gauss-Newton step of the regularised system; mu grows until the factorisation succeeds (DoglegStrategy::ComputeGaussNewtonStep)
      for (int attempt = 0; attempt < 12; ++attempt) {
        x.sync_lds();
        if (!(C.mu < 1.0) || C.lin_ok) break;
        // lower triangle <- upper triangle (unless the scaling pass has just mirrored it), regularised diagonal, right-hand side
        if (C.lower_fresh) {
          for (int r = x.tid; r < npad; r += x.nthr) A[r * ld + r] = (r < n) ? L.hdiag[r] + L.diag[r] * L.diag[r] * C.mu : 1.0;
        } else {
          for (int c = tcol; c < npad; c += CW)
            for (int r0 = trow; r0 < npad; r0 += 8 * rstep) {   // eight LDS reads in flight, then the eight writes
              double v[8];
#pragma unroll
              for (int q = 0; q < 8; ++q) {
                const int r = r0 + q * rstep;
                v[q] = A[(c) * ld + (r < npad ? r : c)];   // (unconditional: a load under a branch waits for the one before it)
              }
#pragma unroll
              for (int q = 0; q < 8; ++q) {
                const int r = r0 + q * rstep;
                if (r < npad && c <= r) A[(r) * ld + c] = (c < r) ? v[q] : ((r < n) ? L.hdiag[r] + L.diag[r] * L.diag[r] * C.mu : 1.0);
              }
            }
        }
        for (int i = x.tid; i < npad; i += x.nthr) L.gz[i] = L.g[i];
        x.sync_lds();
        int ok = ds_ldlt_solve(x, A, ld, npad, L.gz, L.invd, L.part, L.xinv, &C.fact_ok, B.prof);
        if (x.tid == 0) C.lower_fresh = 0;   // (the lower triangle now holds L; every thread read the flag before the factorisation's barriers)
        x.stamp(B.prof, 21);
        int fin = 1;
        if (ok) for (int i = x.tid; i < n; i += x.nthr) if (!(fabs(L.gz[i]) <= 1.7e308)) fin = 0;
        double fv[1] = {double(fin)};
        block_sum<X, 1>(x, L.red, fv);
        if (x.tid == 0) {
          if (ok && fv[0] == double(x.nthr)) C.lin_ok = 1; else C.mu *= 10.0;
        }
      }
      x.sync_lds();
      x.stamp(B.prof, 8);
      if (C.lin_ok) {
        for (int i = x.tid; i < npad; i += x.nthr) {
          const double v = i < n ? L.gz[i] * (-L.diag[i]) : 0.0;
          L.gn[i] = v;
          if (i < n) st.gn[i] = v;
        }
      }
      x.sync_lds();
    } else if (x.tid == 0) C.lin_ok = 1;
    x.sync_lds();
    if (C.lin_ok) {
      double nq[3] = {0.0, 0.0, 0.0};
      for (int i = x.tid; i < n; i += x.nthr) { nq[0] += L.grad[i] * L.grad[i]; nq[1] += L.gn[i] * L.gn[i]; nq[2] += L.grad[i] * L.gn[i]; }
      block_sum<X, 3>(x, L.red, nq);
      const double gnorm = sqrt(nq[0]), gnn = sqrt(nq[1]), gdot = nq[2];
      const double radius = C.radius, alpha = C.alpha;
      int kind;
      double beta = 0.0, dnorm = 0.0;
      if (gnn <= radius) { kind = 0; dnorm = gnn; }
      else if (gnorm * alpha >= radius) { kind = 1; dnorm = radius; }
      else {
        kind = 2;
        const double b_dot_a = -alpha * gdot, a_sq = (alpha * gnorm) * (alpha * gnorm);
        const double bma = a_sq - 2 * b_dot_a + gnn * gnn;
        const double cc = b_dot_a - a_sq;
        const double dd = sqrt(cc * cc + bma * (radius * radius - a_sq));
        beta = (cc <= 0) ? (dd - cc) / bma : (radius * radius - a_sq) / (dd + cc);
      }
      double sn[1] = {0.0};
      for (int i = x.tid; i < npad; i += x.nthr) {
        double sv = 0.0;
        if (i < n) {
          if (kind == 0) sv = L.gn[i];
          else if (kind == 1) sv = -(radius / gnorm) * L.grad[i];
          else sv = (-alpha * (1.0 - beta)) * L.grad[i] + beta * L.gn[i];
          sn[0] += sv * sv;
        }
        L.step[i] = sv;
      }
      if (kind == 2) { block_sum<X, 1>(x, L.red, sn); dnorm = sqrt(sn[0]); } else x.sync_lds();
      for (int i = x.tid; i < n; i += x.nthr) L.step[i] = L.step[i] / L.diag[i];
      x.sync_lds();
      x.stamp(B.prof, 9);
      double mq[2] = {0.0, 0.0};
      if (h_in_lds) ds_vHv(x, A, ld, L.hdiag, n, L.step, mq[1]); else ds_vHv(x, B.Hcur, ld, L.hdiag, n, L.step, mq[1]);
      for (int i = x.tid; i < n; i += x.nthr) mq[0] += L.step[i] * L.g[i];
      block_sum<X, 2>(x, L.red, mq);
      x.stamp(B.prof, 10);
      if (x.tid == 0) {
        C.dogleg_norm = dnorm;
        C.model_change = -(mq[0] + 0.5 * mq[1]);
        C.valid_step = (C.model_change > 0) ? 1 : 0;
      }
    } else if (x.tid == 0) C.valid_step = 0;
    x.sync_lds();
    if (C.valid_step) break;
    if (x.tid == 0) {   // (the others read nothing of this before the barrier at the top of the next pass, which also ends the loop when done)
      if (++C.invalid >= 5) { C.termination = 5; C.done = 1; }
      else { C.mu *= 10.0; C.reuse = 0; if (st.ntrace < 40) st.trace[st.ntrace++] = C.x_cost; }
    }
  }
  x.sync_lds();
  x.stamp(B.prof, 12);
  // ---- P8: write the candidate (Plus), its ambient step norm and the relative lidar poses launch A will read
  if (!C.done) {
    for (int t = x.tid; t < (Wo + 1) * 10 + 1; t += x.nthr) {
      const int f = t / 10, w = t % 10;
      if (f <= Wo) {
        if (w == 0) {
          double d[6];
          for (int k = 0; k < 6; ++k) d[k] = L.step[15 * f + k] * L.scale[15 * f + k];
          pose_plus(P.pose[f], d, st.cand.pose[f]);
        } else {
          const int k = w - 1;
          st.cand.sb[f][k] = P.sb[f][k] + L.step[15 * f + 6 + k] * L.scale[15 * f + 6 + k];
        }
      } else {
        if (exc >= 0) {
          double d[6];
          for (int k = 0; k < 6; ++k) d[k] = L.step[exc + k] * L.scale[exc + k];
          pose_plus(P.ex, d, st.cand.ex);
        } else {
          for (int k = 0; k < 7; ++k) st.cand.ex[k] = P.ex[k];
        }
      }
    }
    x.sync();   // (st.cand is read back from global memory by other threads)
    double sq[1] = {0.0};
    for (int t = x.tid; t < (Wo + 1) * 16 + 7; t += x.nthr) {
      double d = 0;
      if (t < (Wo + 1) * 16) { const int f = t / 16, k = t % 16; d = k < 7 ? P.pose[f][k] - st.cand.pose[f][k] : P.sb[f][k - 7] - st.cand.sb[f][k - 7]; }
      else if (exc >= 0) d = P.ex[t - (Wo + 1) * 16] - st.cand.ex[t - (Wo + 1) * 16];
      sq[0] += d * d;
    }
    block_sum<X, 1>(x, L.red, sq);
    for (int i = x.tid; i < Wo; i += x.nthr) relative_lidar_pose(st.cand.pose[0], st.cand.pose[i + 1], st.cand.ex, st.cand_Rt[i], st.cand_Rt[i] + 9);
    if (x.tid == 0) { C.step_norm = sqrt(sq[0]); C.invalid = 0; }
  }
  x.sync_lds();
  x.stamp(B.prof, 13);
  if (x.tid == 0) {
    st.radius = C.radius; st.mu = C.mu; st.alpha = C.alpha; st.dogleg_norm = C.dogleg_norm; st.gmax = C.gmax; st.x_cost = C.x_cost;
    st.x_norm = C.x_norm; st.model_change = C.model_change; st.step_norm = C.step_norm;
    st.reuse = C.reuse; st.invalid = C.invalid; st.it = C.it; st.successful = C.successful; st.termination = C.termination;
    if (C.done) st.done = 1;
  }
}

// ------------------------------------------------------------------------------------------------ host side: packing
// WindowSystem / WindowParams (host_solver.h) -> the POD problem the kernels read.  false: the problem does not fit the
// device-resident path (too many optimised frames for the LDS-resident factorisation, no lidar factors) -> host solver.
// prior_mats == nullptr: the caller already holds the prior's matrices on the device
inline bool ds_pack_problem(const WindowSystem &sys, const WindowParams &P, int max_iterations, int bpf, bool conv_flag_in, bool imu_on,
                            DevProblem &pb, std::vector<double> *prior_mats) {
  const Layout lay = WindowSystem::solve_layout(P);
  if (P.Wo < 1 || P.Wo > DS_MAX_WO || !sys.use_lidar) return false;
  const int npad = (lay.dim + DS_NB - 1) / DS_NB * DS_NB;
  if (npad > DS_MAX_NPAD) return false;
  std::memset(&pb, 0, sizeof(pb));
  pb.Wo = P.Wo; pb.n = lay.dim; pb.n_pad = npad; pb.ld = npad + 1; pb.ex_col = lay.ex;
  pb.max_iterations = max_iterations; pb.bpf = bpf; pb.conv_flag_in = conv_flag_in ? 1 : 0; pb.imu_on = imu_on ? 1 : 0;
  for (int i = 0; i < DS_MAX_NPAD + 8; ++i) pb.prior_col[i] = -1;
  if (prior_mats) prior_mats->clear();
  if (sys.prior) {
    const MargPrior &pr = *sys.prior;
    if (int(pr.keep.size()) > DS_MAX_KEEP) return false;
    pb.have_prior = 1; pb.n_prior = pr.n; pb.n_keep = int(pr.keep.size());
    int off = 0;
    for (size_t b = 0; b < pr.keep.size(); ++b) {
      const KeepBlock &kb = pr.keep[b];
      pb.keep_kind[b] = kb.kind; pb.keep_index[b] = kb.index; pb.keep_size[b] = kb.size; pb.keep_idx[b] = kb.idx; pb.keep_x0[b] = off;
      for (int k = 0; k < kb.size; ++k) pb.prior_x0[off + k] = pr.x0[b][k];
      off += kb.size;
      const int col = kb.kind == 0 ? lay.pose[kb.index] : (kb.kind == 1 ? lay.sb[kb.index] : lay.ex);
      if (col < 0) continue;
      const int la = kb.size == 7 ? 6 : kb.size;
      for (int i = 0; i < la; ++i) pb.prior_col[col + i] = kb.idx + i;
    }
    if (prior_mats) {
      const size_t np = size_t(pr.n);
      prior_mats->resize(ds_prior_mats_size(pr.n));
      std::memcpy(prior_mats->data(), pr.JtJ.a.data(), sizeof(double) * np * np);
      std::memcpy(prior_mats->data() + np * np, pr.lin_jac.a.data(), sizeof(double) * np * np);
      std::memcpy(prior_mats->data() + 2 * np * np, pr.lin_res.data(), sizeof(double) * np);
      std::memcpy(prior_mats->data() + 2 * np * np + np, pr.Jtr0.data(), sizeof(double) * np);
    }
  }
  pb.use_ex_prior = sys.use_prior_factor ? 1 : 0;
  pb.ex_prior_pos[0] = sys.prior_pos.x; pb.ex_prior_pos[1] = sys.prior_pos.y; pb.ex_prior_pos[2] = sys.prior_pos.z;
  pb.ex_prior_rot[0] = sys.prior_rot.w; pb.ex_prior_rot[1] = sys.prior_rot.x; pb.ex_prior_rot[2] = sys.prior_rot.y; pb.ex_prior_rot[3] = sys.prior_rot.z;
  for (int i = 0; i < P.Wo; ++i) {
    DevPim &d = pb.pim[i];
    if (!sys.pim[i]) continue;
    const Preintegration &pm = *sys.pim[i];
    const double *S = pm.sqrt_info();
    if (!S) return false;
    const PimCore c = pm.core();
    std::memcpy(d.dp, c.dp, sizeof(d.dp)); std::memcpy(d.dq, c.dq, sizeof(d.dq)); std::memcpy(d.dv, c.dv, sizeof(d.dv));
    std::memcpy(d.ba, c.ba, sizeof(d.ba)); std::memcpy(d.bg, c.bg, sizeof(d.bg)); std::memcpy(d.g, c.g, sizeof(d.g));
    d.sum_dt = c.sum_dt; d.present = 1;
    std::memcpy(d.jac, pm.jac, sizeof(d.jac));
    std::memcpy(d.sqrt_info, S, sizeof(d.sqrt_info));
  }
  return true;
}
inline void ds_pack_params(const WindowParams &P, DevParams &d) {
  std::memset(&d, 0, sizeof(d));
  for (int i = 0; i <= P.Wo; ++i) {
    for (int k = 0; k < 7; ++k) d.pose[i][k] = P.pose[i][k];
    for (int k = 0; k < 9; ++k) d.sb[i][k] = P.sb[i][k];
  }
  for (int k = 0; k < 7; ++k) d.ex[k] = P.ex[k];
}
inline void ds_unpack_params(const DevParams &d, WindowParams &P) {
  for (int i = 0; i <= P.Wo; ++i) {
    for (int k = 0; k < 7; ++k) P.pose[i][k] = d.pose[i][k];
    for (int k = 0; k < 9; ++k) P.sb[i][k] = d.sb[i][k];
  }
  for (int k = 0; k < 7; ++k) P.ex[k] = d.ex[k];
}
// state before launch A_0: the initial point is its own first "candidate"
inline void ds_init_state(const WindowParams &P, DevState &st) {
  std::memset(&st, 0, sizeof(st));
  ds_pack_params(P, st.x);
  st.cand = st.x;
  for (int i = 0; i < P.Wo; ++i) relative_lidar_pose(st.cand.pose[0], st.cand.pose[i + 1], st.cand.ex, st.cand_Rt[i], st.cand_Rt[i] + 9);
}

}  // namespace lio
