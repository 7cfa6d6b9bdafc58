# round 6, call 12: kernel trace + PMC traffic of the batched step with the segmented sort (64 and 512 windows)
mkdir -p gpurun_out/r6
R=$(pwd); O=$R/gpurun_out/r6
cd /tmp && export TMPDIR=/tmp
for B in 64 512; do
  (timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_b$B -o b -- python $R/tools/batch_profile.py $B 3 > /dev/null 2>&1)
  python $R/profiles/summarize_rocpd.py /tmp/prof_b$B/b_results.db > $O/batch${B}_kernel_stats.md
done
(timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/prof_f -o f -- python $R/tools/batch_profile.py 64 2 > /dev/null 2>&1)
(timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/prof_w -o w -- python $R/tools/batch_profile.py 64 2 > /dev/null 2>&1)
python $R/profiles/pmc_summary.py /tmp/prof_f/f_results.db /tmp/prof_w/w_results.db $O/batch64_pmc.json > $O/batch64_pmc_hbm_traffic.md 2>&1
cd $R
for B in 8 64 512; do python tools/batch_profile.py $B 6 2>&1 | grep -v "amdgpu.ids" | cut -c1-500; done > $O/call12_batch_profile.txt
cat $O/call12_batch_profile.txt | cut -c1-330
head -24 $O/batch512_kernel_stats.md | cut -c1-170
