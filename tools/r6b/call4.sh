# round 6 (second session), call 4: where the QL eigensolver's clocks go (stamps of thread 0)
R=$PWD
mkdir -p $R/gpurun_out/r6b
{
for B in 8; do LIO_DEBUG_DIGEST=1 timeout 300 python tools/batch_profile.py $B 8 2>&1 | grep -E "^B |digest\]" | cut -c1-400; done
} > $R/gpurun_out/r6b/call4.log 2>&1
cat $R/gpurun_out/r6b/call4.log
