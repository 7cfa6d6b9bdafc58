// two_stream_visibility.hip — does kernel k + 1 of a stream always see what kernel k of the SAME stream wrote, while another
// stream runs the same kind of chain beside it?  (round 6: identical windows of a batch diverged only with two loop groups.)
// The chain of the batched trust-region loop in miniature, per "window" w:
//   k_write  grid (5, 5, W): reads the window's record (written by the previous k_check), writes its 25 x 260-double partial rows = pass
//   k_check  grid (W), one workgroup per window: reads the 25 rows, counts values != pass, bumps the record
// Stream A owns windows [0, W/2), stream B [W/2, W) of the SAME arrays (records and rows packed without padding, as the product
// had them) or of arrays padded to 256 B per window.  Errors are counted per kind.
// build: hipcc --offload-arch=gfx950 -O3 -o two_stream_visibility two_stream_visibility.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define ROWS 25
#define ROW 260

struct Rec { double pass; double cand[12]; double filler[50]; };   // 504 B: not a multiple of the 128-B line

__global__ void __launch_bounds__(256) k_write(Rec *rec, double *rows, size_t rec_stride, size_t row_stride, int w0, int pass, unsigned *err) {
  const int w = w0 + blockIdx.z;
  const Rec *r = reinterpret_cast<const Rec *>(reinterpret_cast<const char *>(rec) + size_t(w) * rec_stride);
  const double seen = r->pass, c = r->cand[blockIdx.y];
  if (threadIdx.x == 0 && (seen != double(pass) || c != double(pass) + blockIdx.y)) atomicAdd(err + 0, 1u);   // the record of the previous k_check
  double *dst = rows + size_t(w) * row_stride + (size_t(blockIdx.y) * 5 + blockIdx.x) * ROW;
  for (int k = threadIdx.x; k < 258; k += 256) dst[k] = double(pass) + 0.5 * seen;
}
__global__ void __launch_bounds__(256) k_check(Rec *rec, const double *rows, size_t rec_stride, size_t row_stride, int w0, int pass, unsigned *err) {
  const int w = w0 + blockIdx.x;
  Rec *r = reinterpret_cast<Rec *>(reinterpret_cast<char *>(rec) + size_t(w) * rec_stride);
  const double *src = rows + size_t(w) * row_stride;
  int bad = 0;
  for (int i = threadIdx.x; i < ROWS * 258; i += 256) {
    const double v = src[size_t(i / 258) * ROW + i % 258];
    if (v != 1.5 * double(pass)) ++bad;
  }
  if (bad) atomicAdd(err + 1, unsigned(bad));
  if (r->pass != double(pass) && threadIdx.x == 0) atomicAdd(err + 2, 1u);   // own record from the previous pass
  __syncthreads();
  if (threadIdx.x == 0) r->pass = double(pass + 1);
  if (threadIdx.x < 12) r->cand[threadIdx.x] = double(pass + 1) + threadIdx.x;
  if (threadIdx.x >= 64 && threadIdx.x < 114) r->filler[threadIdx.x - 64] = double(pass);
}

int main(int argc, char **argv) {
  const int W = 64, passes = 11, reps = argc > 1 ? atoi(argv[1]) : 200;
  std::vector<hipStream_t> st(10);
  for (auto &s : st) CK(hipStreamCreate(&s));
  unsigned *err; CK(hipMalloc(&err, 16));
  for (int padded = 0; padded < 2; ++padded) {
    const size_t rec_stride = padded ? 512 : sizeof(Rec);
    const size_t row_stride = padded ? size_t(ROWS * ROW + 31) / 32 * 32 : size_t(ROWS * ROW);
    char *rec; double *rows;
    CK(hipMalloc(&rec, rec_stride * W)); CK(hipMalloc(&rows, row_stride * W * sizeof(double)));
    for (int mode = 0; mode < 3; ++mode) {   // 0: one stream over all windows, 1: two streams (0, 1), 2: two streams (1, 5)
      CK(hipMemset(err, 0, 16));
      for (int rep = 0; rep < reps; ++rep) {
        std::vector<Rec> h(W);
        for (int w = 0; w < W; ++w) { h[w].pass = 0; for (int k = 0; k < 12; ++k) h[w].cand[k] = k; }
        for (int w = 0; w < W; ++w) CK(hipMemcpyAsync(rec + w * rec_stride, &h[w], sizeof(Rec), hipMemcpyHostToDevice, st[0]));
        CK(hipStreamSynchronize(st[0]));
        const int G = mode == 0 ? 1 : 2;
        for (int g = 0; g < G; ++g) {
          hipStream_t s = mode == 0 ? st[0] : (mode == 1 ? st[g] : st[1 + 4 * g]);
          const int w0 = W * g / G, n = W / G;
          for (int p = 0; p < passes; ++p) {
            hipLaunchKernelGGL(k_write, dim3(5, 5, n), dim3(256), 0, s, reinterpret_cast<Rec *>(rec), rows, rec_stride, row_stride, w0, p, err);
            hipLaunchKernelGGL(k_check, dim3(n), dim3(256), 0, s, reinterpret_cast<Rec *>(rec), rows, rec_stride, row_stride, w0, p, err);
          }
        }
        CK(hipDeviceSynchronize());
      }
      unsigned h[4]; CK(hipMemcpy(h, err, 16, hipMemcpyDeviceToHost));
      std::printf("%s mode %d (%s): %d reps x %d passes x %d windows: stale record seen by k_write %u, stale row values seen by k_check %u, stale own record %u\n",
                  padded ? "padded  " : "unpadded", mode, mode == 0 ? "one stream" : (mode == 1 ? "streams 0,1" : "streams 1,5"), reps, passes, W, h[0], h[1], h[2]);
    }
    CK(hipFree(rec)); CK(hipFree(rows));
  }
  return 0;
}
