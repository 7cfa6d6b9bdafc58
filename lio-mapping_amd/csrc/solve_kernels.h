// solve_kernels.h — device side of the Gauss-Newton normal-equation build for the lidar factors
// (PivotPointPlaneFactor, src/factor/PivotPointPlaneFactor.cc:43-137, with CauchyLoss(1.0),
// Estimator.cc:1664,1867-1876) — the one dense contraction of the solve, on the fp64 matrix cores.
#pragma once
#include <cstdint>

#include "cloud_kernels.h"

namespace lio {

// Per residual the 18 Jacobian entries and the residual are LINEAR in z = [w (x) [p;1]; d] (13 values):
// j = L z, r = l^T z with L, l depending only on the (pivot, frame i, extrinsic) poses.  Hence
//   sum rho' j j^T = L (sum rho' z z^T) L^T ,   sum rho' j r = L (sum rho' z z^T) l
// and the K = N_res contraction is S_i = sum_k rho'_k z_k z_k^T — a 16x16xK product (13 padded to 16)
// that maps onto v_mfma_f64_16x16x4_f64 with no wasted tiles.  rho'_k needs the residual at the current
// poses, computed per lane in fp64 from T_{pivot<-i}.
#define LIO_MOMENT_OUT 260  // 256 S entries (row-major 16x16) + cost + count + 2 pad

struct MomentFrame {
  const float4 *stack;
  int M;
  int slot_off;
  int nslots;
  int slot_begin, slot_end;  // this rank's share of [0, nslots) (factor sharding); the whole range on one GPU
  double R[9];  // R_{lp,i} row-major
  double t[3];  // P_{lp,i}
};
struct MomentArgs {
  MomentFrame fr[LIO_MAX_FRAMES];
  int nframes;
  int blocks_per_frame;
  int form;   // 0: MFMA or VALU form by chunks per wave, 1: MFMA, 2: VALU (lio_est_config.moments_form)
};

// partials: nframes * blocks_per_frame * LIO_MOMENT_OUT doubles; out: nframes * LIO_MOMENT_OUT doubles
// tickets: nframes ints, zero before the first launch (the kernel re-zeroes them) -> the fold runs inside the same launch;
// tickets == nullptr -> separate k_moment_reduce launch.
void launch_lidar_moments(const MomentArgs &a, const uint8_t *valid, const float4 *coef, double *partials, int *tickets, double *out,
                          hipStream_t s, const HostSignal &sig = HostSignal());
int moment_blocks_per_frame(int max_slots);
int moment_blocks_per_frame_batched(int max_slots, int nframes);
// same pass over `nframes` frame descriptors held in device memory (any number of windows in one launch)
void launch_lidar_moments_batched(const MomentFrame *d_frames, int nframes, int blocks_per_frame, int max_slots, const uint8_t *valid,
                                  const float4 *coef, double *partials, double *out, hipStream_t s, int form = 0);

// One iteration of the device-resident dogleg (solve_step.h): launch A (moments at the candidate + aux row) and launch B
// (k_solve_step), both on `s`, no host interaction.  The StepBuffers' partials must be `partials`.
struct DevProblem;
struct DevState;
struct StepBuffers;
void launch_solve_iteration(const MomentArgs &a, const uint8_t *valid, const float4 *coef, double *partials, const DevProblem *pb, DevState *st,
                            const StepBuffers &B, double *imu_out, double *lmap, double *prior_out, double *exprior_out, int n_pad, hipStream_t s);

// test hook: x = A^-1 b through the blocked LDS L D L^T of launch B (A symmetric positive definite, n <= 128, row-major).
// 1 ok, 0 a pivot was not positive, -1 the upper triangle was disturbed, -2 does not fit.
int ldlt_solve_device(const double *A, const double *b, int n, double *x_host, hipStream_t s);

}  // namespace lio
