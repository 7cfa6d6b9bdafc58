# round 6, call 2: kernel-boundary visibility with two streams (micro) + which termination the diverging windows report
mkdir -p gpurun_out/r6
{
timeout 300 tools/micro/two_stream_visibility 300
python tools/r6/diag_determinism.py 64 10 1
} > gpurun_out/r6/call2.log 2>&1
tail -40 gpurun_out/r6/call2.log
