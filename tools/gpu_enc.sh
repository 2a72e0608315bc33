set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_rle.py tests/test_shard.py tests/test_gpu_pipeline.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/t1.log 2>&1 || { tail -40 gpurun_out/t1.log; exit 1; }
tail -3 gpurun_out/t1.log
for mode in pertile persistent pertile persistent; do
JPEGAMD_GRID=$mode timeout -k 10 300 python bench.py --steps 20 --no-cpu > gpurun_out/bench_$mode.json 2> gpurun_out/bench_$mode.err || { tail -20 gpurun_out/bench_$mode.err; exit 1; }
python - $mode <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/bench_{sys.argv[1]}.json"))
print(sys.argv[1], round(d["value"]), d["roofline"]["encode_ms"], d["roofline"]["decode_ms"], d["worst_case_input"]["encode_ms"], d["rle_variant"]["ms"])
PY
done
