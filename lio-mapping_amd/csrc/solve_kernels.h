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

// partials: nframes * blocks_per_frame * LIO_MOMENT_OUT doubles; out: nframes * LIO_MOMENT_OUT doubles (k_moment_reduce folds them)
void launch_lidar_moments(const MomentArgs &a, const uint8_t *valid, const float4 *coef, double *partials, double *out, hipStream_t s,
                          const HostSignal &sig = HostSignal());
int moment_blocks_per_frame(int max_slots);

// ---- resident form: ONE launch per solve.  The worker blocks keep their residuals in registers and wait for
// every linearisation point (Wo relative poses + a sequence number) on a doorbell; each pass a wave pushes its cached residuals
// through the fp64 MFMA tile, the block parks a compact record (the 13 x 13 upper triangle, cost, count: 93 doubles) in HBM behind
// a per-block flag; block 0 of each frame waits for its frame's flags, folds the blocks in k_moment_reduce's order (two lanes per
// value) and posts the frame's moments (the triangle mirrored into the padded 16 x 16 tile) + a completion word to coherent host
// memory.  One extra block — the
// relay — is the only poller of host memory: measured on the MI355X box (tools/micro/pingpong.hip) a host -> kernel -> host
// round trip is 2.3 us with one polling block and 13-19 us with a hundred, so the relay republishes the doorbell in HBM and the
// workers poll that copy.  Arithmetic per block = k_lidar_moments (MFMA form) at the same blocks per frame and the fold =
// k_moment_reduce's, so the moments are bit-identical to the two-launch path in that configuration.
#define LIO_RES_OUT 264            // doubles per frame in the host landing zone: [0, 91) the 13 x 13 upper triangle of S (row-major), 91 cost,
                                   // 92 count, [258, 264) diagnostics
#define LIO_RES_NTRI 91
#define LIO_RES_DOOR 16            // doubles per frame in the doorbell: [R0..R6, seq | R7, R8, t0, t1, t2, 0, 0, seq] (two cache lines)
#define LIO_RES_EXPIRED 0xFFFFFFFFu
// the doorbell value that ends a launch: minus the sequence number of its first pass — unique per launch, so whatever an earlier
// launch left in the HBM copy of the doorbell (its STOP, its last sequence number) means nothing to this one and needs no clearing
#define LIO_RES_STOP(first_seq) (-double(first_seq))
#define LIO_RES_MAX_BLOCKS 256     // size of the per-block record array; the admission limit is resident_max_blocks() (asked of the device)
struct ResidentArgs {
  const double *door;        // host, coherent
  double *out;               // host, coherent: frame f's folded moments at f * LIO_RES_OUT (compact: see LIO_RES_OUT)
  unsigned *words;           // host, coherent: one completion word per frame, [LIO_MAX_FRAMES] = the relay's (LIO_RES_EXPIRED on a timeout),
                             // [LIO_MAX_FRAMES + 1] = the relay's echo of every sequence number it has seen
  unsigned first_seq;        // the sequence number of the first pass this launch serves
  long long timeout_ticks;   // wall-clock ticks (hipDeviceAttributeWallClockRate) without a doorbell before the kernel gives up
  double *relay;             // device: the doorbell republished by the relay block (same layout)
  double *block_part;        // device: block b's record at b * LIO_MOMENT_OUT; slot LIO_MOMENT_OUT - 1 = the pass it belongs to (the flag)
  int diag;                  // 1: every phase stamp goes into the frame record (LIO_DEBUG_TIMING), 0: only the pass time
};
// worker blocks the current device keeps co-resident beside the relay (occupancy x CUs - 1, <= LIO_RES_MAX_BLOCKS; 0 without a device)
int resident_max_blocks(int per_lane);
// blocks per frame so that a lane holds at most `per_lane` residuals (0 when the window does not fit resident_max_blocks)
int resident_blocks_per_frame(int max_slots, int nframes, int per_lane);
void launch_lidar_moments_resident(const MomentArgs &a, const ResidentArgs &ra, int per_lane, const uint8_t *valid, const float4 *coef, hipStream_t s);
int moment_blocks_per_frame_batched(int max_slots, int nframes);
// same pass over `nframes` frame descriptors held in device memory (any number of windows in one launch)
void launch_lidar_moments_batched(const MomentFrame *d_frames, int nframes, int blocks_per_frame, int max_slots, const uint8_t *valid,
                                  const float4 *coef, double *partials, double *out, hipStream_t s, int form = 0);

// One iteration of the device-resident dogleg (solve_step.h) for every window of a batch: launch A (moments at the candidate + aux
// row) and launch B (one workgroup per window), both on `s`, no host interaction.  BatchSolve is declared in solve_step.h.
struct BatchSolve;
struct BatchBases;
// the three launches of an iteration one by one (the aux row depends on the candidate only: it can run beside the moments on a stream of its own)
void launch_bw_aux(const BatchSolve *bs, const BatchBases &bb, int B, int max_wo, int threads_or_0, hipStream_t s);
void launch_bw_moments(const BatchSolve *bs, const BatchBases &bb, int B, int max_bpf, int max_wo, const uint8_t *valid, const float4 *coef, hipStream_t s);
void prepare_bw_step_kernel();   // per device, before the first launch_bw_step on it
void launch_bw_step(const BatchSolve *bs, const BatchBases &bb, int B, int max_wo, int max_npad, hipStream_t s);
void launch_bw_solve_iteration(const BatchSolve *bs, const BatchBases &bb, int B, int max_bpf, int max_wo, int max_npad, int aux_threads_or_0, const uint8_t *valid, const float4 *coef,
                               hipStream_t s);
// blocks per frame of a window's moments pass inside a batch: a function of the window's own slot counts only
int batch_blocks_per_frame(int max_slots);

// test hook: x = A^-1 b through the blocked LDS L D L^T of launch B (A symmetric positive definite, n <= 128, row-major).
// 1 ok, 0 a pivot was not positive, -1 the upper triangle was disturbed, -2 does not fit.
int ldlt_solve_device(const double *A, const double *b, int n, double *x_host, hipStream_t s);

}  // namespace lio
