# round 3: one SQ counter pass of tools/prof_run.py per environment setting given ("NAME=value"), one box
set -o pipefail
mkdir -p /root/repo/gpurun_out
OUT=/root/repo/gpurun_out/r03_pmc_env.txt
: > $OUT
cd /tmp; export TMPDIR=/tmp
C="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY"
i=0
for cfg in "$@"; do
  i=$((i+1))
  export "$cfg"
  rm -rf /tmp/pm_env$i
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d /tmp/pm_env$i -- python3 /root/repo/tools/prof_run.py 8192 > /root/repo/gpurun_out/pm_env$i.log 2>&1 || { tail -5 /root/repo/gpurun_out/pm_env$i.log; exit 1; }
  echo "== $cfg" >> $OUT
  python3 /root/repo/tools/pmc_summary.py /tmp/pm_env$i | grep encode >> $OUT
done
cat $OUT
