# round 6 (second session), call 12: more loop groups with the interleaved enqueue (limit raised to 8)
R=$PWD
mkdir -p $R/gpurun_out/r6b
{
for V in "LIO_BW_GROUPS=1" "LIO_BW_GROUPS=2" "LIO_BW_GROUPS=4" "LIO_BW_GROUPS=6" "LIO_BW_GROUPS=8"; do for B in 16 32 64 128 256 512; do echo "== $V B=$B"; env $V timeout 300 python tools/batch_profile.py $B 8 2>&1 | grep -o "B [0-9]*: [0-9]* solves/s\|'dev_loop': [0-9.]*" | tr '\n' ' '; echo; done; done
} > $R/gpurun_out/r6b/call12.log 2>&1
cat $R/gpurun_out/r6b/call12.log
