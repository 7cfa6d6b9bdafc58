# round 6 (second session), call 1: several lio_est_batch objects in flight from several host threads
mkdir -p gpurun_out/r6b
{
for cfg in "8 1" "8 2 16 1" "8 2" "64 1" "64 2 8 1" "64 2" "64 4 8 1" "512 1 4" "512 2 4 1" "512 2 4" "512 4 4 1"; do
  timeout 300 python tools/batches_in_flight.py $cfg 2>&1 | tail -1
done
} > gpurun_out/r6b/call1.log 2>&1
cat gpurun_out/r6b/call1.log
