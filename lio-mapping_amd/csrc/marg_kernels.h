// marg_kernels.h — MarginalizationInfo::Marginalize's dense tail on the device (MarginalizationFactor.cc:271-302):
// Amm^+ through the eigendecomposition of the marginalised block, the Schur complement on the fp64 matrix cores, the
// eigendecomposition of the complement, and the square-root factors linearized_jacobians / linearized_residuals.
#pragma once
#include <hip/hip_runtime.h>

namespace lio {

#define MARG_MAX_M 16   // marginalised parameters: pose (6) + speed-bias (9) = 15
#define MARG_MAX_N 80   // kept parameters: 15 + 6 Wo; opt windows above 10 marginalise on the host

// Owns the device buffers and a stream of its own (the marginalization runs on the estimator's worker thread).
class MargSchurDev {
 public:
  explicit MargSchurDev(int device);
  ~MargSchurDev();
  MargSchurDev(const MargSchurDev &) = delete;
  MargSchurDev &operator=(const MargSchurDev &) = delete;
  // A: (m + n)^2 row-major, b: m + n (host).  Outputs (host): lin_jac n x n row-major with row k = sqrt(s_k) v_k^T, lin_res n
  // (s ascending, entries with s_k <= eps zeroed), evals n.  Returns false when the shape is out of range (caller: host path).
  bool Run(const double *A, const double *b, int m, int n, double eps, double *lin_jac, double *lin_res, double *evals, int *sweeps);
  double last_ms() const { return last_ms_; }

 private:
  int device_;
  hipStream_t stream_ = nullptr;
  double *d_in_ = nullptr, *d_out_ = nullptr;   // [A | b], [lin_jac | lin_res | evals | info]
  double *h_io_ = nullptr;                      // pinned staging for both
  double last_ms_ = 0;
};

// Marginalization of every window of a batch on `s` (two launches; BatchSolve: solve_step.h); the new priors stay on the device
struct BatchSolve;
struct BatchBases;
void prepare_bw_marg_kernel();   // per device, before the first launch_bw_marginalize on it
void launch_bw_marginalize(const BatchSolve *bs, const BatchBases &bb, int B, int max_wo, int max_n, hipStream_t s);

}  // namespace lio
