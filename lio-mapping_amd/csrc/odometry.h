// odometry.h — PointOdometry (LOAM scan-to-scan step) on the GPU.
// Reference: src/point_processor/PointOdometry.cc:237-292 (TransformToStart/End), :294-683 (Process).
#pragma once
#include "cloud_kernels.h"
#include "hmath.h"

namespace lio {

class OdometryDev {
 public:
  OdometryDev(float scan_period, int io_ratio, int max_iter, bool no_deskew);
  ~OdometryDev();
  void Process(const float *sharp, size_t n_sharp, const float *less_sharp, size_t n_ls, const float *flat, size_t n_flat, const float *less_flat,
               size_t n_lf);
  size_t GetLastCloud(int which, float *out);

  Rigid<float> transform_es_, transform_sum_;
  int iterations_done_ = 0, last_num_sel_ = 0;
  int last_kz_ = 0;                         // degeneracy test of iteration 0 (:584-615): leading update components masked
  std::vector<Rigid<float>> es_trace_;      // transform_es_ after every iteration of the last Process (lio_odom_get_iteration_trace)
  bool enable_odom_ = true;

 private:
  float scan_period_, time_factor_;
  int io_ratio_, max_iter_;
  bool no_deskew_, inited_ = false;
  hipStream_t stream_ = nullptr;
  DBuf<float4> sharp_, flat_, less_sharp_, less_flat_, last_corner_, last_surf_;
  size_t n_last_corner_ = 0, n_last_surf_ = 0;
  DBuf<int> idx_;              // 2*nc + 3*ns correspondence indices
  DBuf<OdomState> d_state_;
  DBuf<double> d_partials_;
  DBuf<float> d_trace_;        // 8 floats per iteration, written by k_odo_update
  std::vector<float> h_trace_;
  void BuildGrids();
  KnnGrid grid_c_, grid_s_;
  DBuf<float> partial_c_, partial_s_;
  DBuf<VoxParams> bounds_;
  VoxParams *h_bounds_ = nullptr;  // pinned
  OdomState *h_state_ = nullptr;   // pinned, coherent: the state's mailbox
  unsigned *h_flag_ = nullptr;     // its completion word
  unsigned seq_ = 0;
};

}  // namespace lio
