set -e
cd /tmp; export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL"
B="SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH SQ_IFETCH"
for tr in valu xdl; do
  export JPEGAMD_TRANSFORM=$tr
  for p in A B; do
    eval "C=\$$p"
    rm -rf /tmp/pm_$tr$p
    timeout -s KILL 90 rocprofv3 --pmc $C --output-format csv -d /tmp/pm_$tr$p -- python3 /root/repo/tools/prof_run.py 8192 > /root/repo/gpurun_out/pm_$tr$p.log 2>&1
    echo "== $tr $p" >> /root/repo/gpurun_out/pmc_detail.txt
    python3 /root/repo/tools/pmc_summary.py /tmp/pm_$tr$p | grep encode >> /root/repo/gpurun_out/pmc_detail.txt
  done
done
cat /root/repo/gpurun_out/pmc_detail.txt
