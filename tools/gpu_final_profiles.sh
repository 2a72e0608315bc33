# The round's evidence on the tree as it is, each step under its own time limit -> gpurun_out/final/
set -o pipefail
OUT=/root/repo/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. per-kernel durations of the default bench (fused kernels + the run-length variant's kernels)
rm -rf /tmp/kt_bench
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_bench -- python3 /root/repo/bench.py --steps 20 --no-cpu > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err || { tail -20 $OUT/bench_under_rocprof.err; exit 1; }
cp "$(find /tmp/kt_bench -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_bench.csv || exit 1
# 2. the same for image -> .bin -> image from host memory (run-length, entropy coder and decoder kernels)
rm -rf /tmp/kt_pipe
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_pipe -- python3 /root/repo/tools/pipeline_repeat.py 8192 > $OUT/pipeline_under_rocprof.txt 2> $OUT/pipeline_under_rocprof.err || { tail -20 $OUT/pipeline_under_rocprof.err; exit 1; }
cp "$(find /tmp/kt_pipe -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_pipeline.csv || exit 1
# 3. SQ counters of the two fused kernels, one pass per group
: > $OUT/pmc_sq.txt
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rm -rf /tmp/pm_$i
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d /tmp/pm_$i -- python3 /root/repo/tools/prof_run.py 8192 > $OUT/pm_$i.log 2>&1 || { tail -5 $OUT/pm_$i.log; exit 1; }
  python3 /root/repo/tools/pmc_summary.py /tmp/pm_$i >> $OUT/pmc_sq.txt || exit 1
done
# 4. HBM traffic (FETCH_SIZE, WRITE_SIZE in passes of their own)
bash /root/repo/tools/gpu_pmc_hbm.sh > $OUT/pmc_hbm.log 2>&1 || { tail -5 $OUT/pmc_hbm.log; exit 1; }
cd /root/repo
# 5. the bench line with its CPU leg, and the 2-rank rehearsal of the launcher on this one GPU
timeout -k 10 600 python3 bench.py > $OUT/bench_final.json 2> $OUT/bench_final.err || { tail -20 $OUT/bench_final.err; exit 1; }
timeout -k 10 300 python3 bench.py --gpus 2 --no-cpu > $OUT/bench_gpus2_rehearsal.json 2> $OUT/bench_gpus2_rehearsal.err || { tail -20 $OUT/bench_gpus2_rehearsal.err; exit 1; }
cut -c1-160 $OUT/kernel_stats_bench.csv; cut -c1-160 $OUT/kernel_stats_pipeline.csv; cat $OUT/pmc_sq.txt; tail -3 $OUT/pmc_hbm.log; cut -c1-400 $OUT/bench_final.json; cut -c1-300 $OUT/bench_gpus2_rehearsal.json
