# round 3: kernel durations and SQ / LDS / memory counters of the run-length and entropy kernels (VERDICT r2 "missing" 4)
set -o pipefail
mkdir -p /root/repo/gpurun_out
OUT=/root/repo/gpurun_out/r03_pmc_stages.txt
: > $OUT
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/kt_st
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_st -- python3 /root/repo/tools/prof_run_stages.py 8192 > /root/repo/gpurun_out/kt_st.log 2>&1 || { tail -5 /root/repo/gpurun_out/kt_st.log; exit 1; }
cp "$(find /tmp/kt_st -name '*kernel_stats.csv' | head -1)" /root/repo/gpurun_out/r03_kernel_stats_stages.csv
grep -v "at::native\|elementwise\|vectorized" /root/repo/gpurun_out/r03_kernel_stats_stages.csv | cut -c1-160
A="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES"
B="SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM"
for p in A B; do
  eval "C=\$$p"
  rm -rf /tmp/pm_st$p
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d /tmp/pm_st$p -- python3 /root/repo/tools/prof_run_stages.py 8192 > /root/repo/gpurun_out/pm_st$p.log 2>&1 || { tail -5 /root/repo/gpurun_out/pm_st$p.log; exit 1; }
  echo "== pass $p" >> $OUT
  python3 /root/repo/tools/pmc_summary.py /tmp/pm_st$p all >> $OUT
done
cat $OUT
