# Round 3's evidence on the tree as it is, each step under its own time limit -> gpurun_out/r03_final/
# (copied to profiles/r03_f_* and r03_pmc_hbm_traffic.* afterwards)
set -o pipefail
OUT=/root/repo/gpurun_out/r03_final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. per-kernel durations: the default bench command, the headline input alone, image -> .bin -> image
rm -rf /tmp/kt_bench /tmp/kt_head /tmp/kt_pipe
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_bench -- python3 /root/repo/bench.py --steps 20 --no-cpu > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err || { tail -20 $OUT/bench_under_rocprof.err; exit 1; }
cp "$(find /tmp/kt_bench -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_bench.csv || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_head -- python3 /root/repo/bench.py --steps 20 --no-cpu --headline-only > $OUT/bench_headline_under_rocprof.json 2> $OUT/bench_headline_under_rocprof.err || { tail -20 $OUT/bench_headline_under_rocprof.err; exit 1; }
cp "$(find /tmp/kt_head -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_headline.csv || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_pipe -- python3 /root/repo/tools/pipeline_repeat.py 8192 > $OUT/pipeline_under_rocprof.txt 2> $OUT/pipeline_under_rocprof.err || { tail -20 $OUT/pipeline_under_rocprof.err; exit 1; }
cp "$(find /tmp/kt_pipe -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_pipeline.csv || exit 1
# 2. SQ counters of the two fused kernels, one pass per group of eight
: > $OUT/pmc_sq.txt
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rm -rf /tmp/pm_$i
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d /tmp/pm_$i -- python3 /root/repo/tools/prof_run.py 8192 > $OUT/pm_$i.log 2>&1 || { tail -5 $OUT/pm_$i.log; exit 1; }
  python3 /root/repo/tools/pmc_summary.py /tmp/pm_$i >> $OUT/pmc_sq.txt || exit 1
done
python3 - <<'PY'
import ast, json, collections
out = collections.defaultdict(dict)
for line in open("/root/repo/gpurun_out/r03_final/pmc_sq.txt"):
    k, d = line.split(" ", 1)
    out[k].update(ast.literal_eval(d))
json.dump({"8192x8192_qf50": out}, open("/root/repo/gpurun_out/r03_final/pmc_sq.json", "w"), indent=1)
PY
# 3. the same two groups for the run-length and entropy kernels (image -> pairs -> .bin -> pairs -> image on the device)
: > $OUT/pmc_stages.txt
A="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES"
B="SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM"
for p in A B; do
  eval "C=\$$p"
  rm -rf /tmp/pm_st$p
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d /tmp/pm_st$p -- python3 /root/repo/tools/prof_run_stages.py 8192 > $OUT/pm_st$p.log 2>&1 || { tail -5 $OUT/pm_st$p.log; exit 1; }
  echo "== pass $p" >> $OUT/pmc_stages.txt
  python3 /root/repo/tools/pmc_summary.py /tmp/pm_st$p all >> $OUT/pmc_stages.txt || exit 1
done
rm -rf /tmp/kt_st
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_st -- python3 /root/repo/tools/prof_run_stages.py 8192 > $OUT/kt_st.log 2>&1 || { tail -5 $OUT/kt_st.log; exit 1; }
cp "$(find /tmp/kt_st -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_stages.csv || exit 1
# 4. HBM traffic (FETCH_SIZE, WRITE_SIZE in passes of their own) -> gpurun_out/r03_pmc_hbm_traffic.{txt,json}
JPEGAMD_ROUND=r03 bash /root/repo/tools/gpu_pmc_hbm.sh > $OUT/pmc_hbm.log 2>&1 || { tail -5 $OUT/pmc_hbm.log; exit 1; }
cd /root/repo
# 5. the bench line with its CPU leg, the strips mode, and the 2-rank rehearsals of both launch modes on this one GPU
timeout -k 10 600 python3 bench.py > $OUT/bench_final.json 2> $OUT/bench_final.err || { tail -20 $OUT/bench_final.err; exit 1; }
timeout -k 10 300 python3 bench.py --mode strips > $OUT/strips_n1.json 2> $OUT/strips_n1.err || { tail -20 $OUT/strips_n1.err; exit 1; }
timeout -k 10 300 python3 bench.py --mode strips --gpus 2 > $OUT/strips_n2_rehearsal.json 2> $OUT/strips_n2.err || { tail -20 $OUT/strips_n2.err; exit 1; }
timeout -k 10 300 python3 bench.py --gpus 2 --no-cpu > $OUT/bench_gpus2_rehearsal.json 2> $OUT/bench_gpus2_rehearsal.err || { tail -20 $OUT/bench_gpus2_rehearsal.err; exit 1; }
grep -v "at::native\|elementwise\|vectorized" $OUT/kernel_stats_headline.csv | cut -c1-160; grep -v "at::native\|elementwise\|vectorized" $OUT/kernel_stats_stages.csv | cut -c1-160; cat $OUT/pmc_sq.txt; tail -3 $OUT/pmc_hbm.log; cut -c1-500 $OUT/bench_final.json; cut -c1-300 $OUT/strips_n1.json; cut -c1-300 $OUT/strips_n2_rehearsal.json
