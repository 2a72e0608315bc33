# round 3: SQ counters of the VALU encoder and the persistent f16 row-pass variant on one box
set -o pipefail
mkdir -p /root/repo/gpurun_out
OUT=/root/repo/gpurun_out/r03_pmc_variants.txt
: > $OUT
cd /tmp; export TMPDIR=/tmp
C="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
for tr in valu xdl; do
  export JPEGAMD_TRANSFORM=$tr
  rm -rf /tmp/pm_$tr
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d /tmp/pm_$tr -- python3 /root/repo/tools/prof_run.py 8192 > /root/repo/gpurun_out/pm_$tr.log 2>&1 || { tail -5 /root/repo/gpurun_out/pm_$tr.log; exit 1; }
  echo "== JPEGAMD_TRANSFORM=$tr" >> $OUT
  python3 /root/repo/tools/pmc_summary.py /tmp/pm_$tr | grep encode >> $OUT
done
cat $OUT
