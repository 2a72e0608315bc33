# one rocprofv3 --pmc pass per counter group over tools/prof_run.py (a handful of launches)
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
: > /root/repo/gpurun_out/pmc_sq.txt
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rm -rf /tmp/pm_$i
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d /tmp/pm_$i -- python3 /root/repo/tools/prof_run.py 8192 > /root/repo/gpurun_out/pm_$i.log 2>&1 || { tail -5 /root/repo/gpurun_out/pm_$i.log; exit 1; }
  python3 /root/repo/tools/pmc_summary.py /tmp/pm_$i >> /root/repo/gpurun_out/pmc_sq.txt
done
cat /root/repo/gpurun_out/pmc_sq.txt
