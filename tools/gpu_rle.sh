set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_rle.py tests/wip_no_alloc.py tests/test_gpu_pipeline.py tests/test_gpu_entropy.py -x -q --timeout 120 --timeout-method thread > gpurun_out/t1.log 2>&1 || { tail -30 gpurun_out/t1.log; exit 1; }
tail -3 gpurun_out/t1.log
timeout -k 10 300 python bench.py --steps 20 --no-cpu > gpurun_out/bench1.json 2> gpurun_out/bench1.err || { tail -20 gpurun_out/bench1.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench1.json"))
print(d["value"], d["roofline"]["encode_ms"], d["roofline"]["decode_ms"])
print(d["rle_variant"])
PY
