# rocprofv3 kernel trace of the two fused kernels on the headline input alone -> gpurun_out/final/
set -o pipefail
OUT=/root/repo/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_head
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_head -- python3 /root/repo/bench.py --steps 20 --no-cpu --headline-only > $OUT/bench_headline_under_rocprof.json 2> $OUT/bench_headline_under_rocprof.err || { tail -20 $OUT/bench_headline_under_rocprof.err; exit 1; }
cp "$(find /tmp/kt_head -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_headline.csv || exit 1
grep -v "at::native" $OUT/kernel_stats_headline.csv | cut -c1-200
cut -c1-900 $OUT/bench_headline_under_rocprof.json
cd /root/repo && timeout -k 10 300 python3 bench.py --steps 20 --no-cpu --headline-only | cut -c1-1200
