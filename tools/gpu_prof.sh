# rocprofv3 kernel trace of the default bench (no CPU leg): per-kernel durations
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- python3 /root/repo/bench.py --steps 20 --no-cpu > /root/repo/gpurun_out/prof_bench.json 2> /root/repo/gpurun_out/prof_bench.err || { tail -20 /root/repo/gpurun_out/prof_bench.err; exit 1; }
f=$(find /tmp/prof_kt -name '*kernel_stats.csv' | head -1)
cp "$f" /root/repo/gpurun_out/kernel_stats.csv
cat /root/repo/gpurun_out/kernel_stats.csv | cut -c1-200
