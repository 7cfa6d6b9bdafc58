// Host harness: the product's StartOriFilter (csrc/pointproc.h, PointProcessor.cc:348-387) fed from stdin.
// Each input line: measured ring0_front (radians; "nan" allowed for ring0_front).  Output: the start azimuth used, one per line.
#include <cstdio>
#include <cstdlib>

#include "pointproc.h"

int main(int argc, char **argv) {
  const double rad_diff = argc > 1 ? std::atof(argv[1]) : 0.2;
  lio::StartOriFilter f;
  double m, r;
  while (std::scanf("%lf %lf", &m, &r) == 2) std::printf("%.9g\n", double(f.Update(float(m), float(r), rad_diff)));
  return 0;
}
