# one rocprofv3 --pmc pass over tools/prof_run.py with the environment as given
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm_one
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d /tmp/pm_one -- python3 /root/repo/tools/prof_run.py 8192 > /root/repo/gpurun_out/pm_one.log 2>&1 || { tail -5 /root/repo/gpurun_out/pm_one.log; exit 1; }
python3 /root/repo/tools/pmc_summary.py /tmp/pm_one
