#!/bin/bash
# builds lio-mapping_amd/csrc/liblio_hip_<tag>.so with extra compiler flags (A/B builds of compile-time constants); load it with LIO_HIP_LIB=<path>
# usage: tools/r5/build_variant.sh t256 -DDS_THREADS=256
set -e
tag=$1; shift
cd "$(dirname "$0")/../../lio-mapping_amd/csrc"
out=/tmp/lio_variant_$tag; mkdir -p $out
for f in seg_sort cloud_kernels batch_kernels solve_kernels pointproc odometry mapping kf_batch marg_kernels estimator est_batch rccl_comm capi; do
  if [ ! -f $out/$f.o ] || [ $f.hip -nt $out/$f.o ] || [ solve_step.h -nt $out/$f.o ] || [ solve_device.h -nt $out/$f.o ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mavx2 -fPIC -Wno-unused-function -Wno-unused-result "$@" -c $f.hip -o $out/$f.o &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o liblio_hip_$tag.so $out/*.o -ldl -Wl,-rpath,/opt/rocm/lib
ls -la liblio_hip_$tag.so
