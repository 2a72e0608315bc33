# the headline input alone under rocprofv3 --kernel-trace --stats, three times on one box, then once without the profiler
set -o pipefail
OUT=/root/repo/gpurun_out/r03_head
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do
  rm -rf /tmp/kt_head$i
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_head$i -- python3 /root/repo/bench.py --steps 20 --no-cpu --headline-only > $OUT/bench_$i.json 2> $OUT/bench_$i.err || { tail -20 $OUT/bench_$i.err; exit 1; }
  cp "$(find /tmp/kt_head$i -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_$i.csv || exit 1
  grep "image_kernel" $OUT/kernel_stats_$i.csv | cut -d, -f2-9 | sed 's/^.*)",//'
done
cd /root/repo && timeout -k 10 300 python3 bench.py --steps 20 --no-cpu --headline-only > $OUT/bench_plain.json 2> $OUT/bench_plain.err || exit 1
python3 - <<'PY'
import json
for n in ("1", "2", "3", "plain"):
    d = json.load(open(f"/root/repo/gpurun_out/r03_head/bench_{n}.json"))
    print(n, round(d["value"]), d["ms_per_step"], d["roofline"]["encode_ms"], d["roofline"]["decode_ms"])
PY
