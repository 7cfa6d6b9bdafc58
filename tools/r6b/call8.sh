# round 6 (second session), call 8: k_bw_moments / k_lidar_moments_batched A/B — 13 doubles per residual in the transpose buffer (LDS 34.8 -> 26.6 KB per block), 5 / 6 waves per SIMD asked of the compiler
R=$PWD
mkdir -p $R/gpurun_out/r6b
{
for tag in "" zb13 zb13w5 zb13w6 w5; do
  lib=$R/lio-mapping_amd/csrc/liblio_hip${tag:+_$tag}.so
  echo "== ${tag:-shipped}"
  LIO_HIP_LIB=$lib timeout 300 python tools/batched_moments.py 64 512 2>&1 | tail -2
  for B in 512 64; do LIO_HIP_LIB=$lib timeout 300 python tools/batch_profile.py $B 6 2>&1 | grep -o "B [0-9]*: [0-9]* solves/s\|'dev_loop': [0-9.]*" | tr '\n' ' '; echo; done
done
} > $R/gpurun_out/r6b/call8.log 2>&1
cat $R/gpurun_out/r6b/call8.log
