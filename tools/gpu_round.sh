# whole GPU suite + default bench (no CPU leg) -> gpurun_out/
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/t_all.log 2>&1 || { tail -40 gpurun_out/t_all.log; exit 1; }
tail -3 gpurun_out/t_all.log
timeout -k 10 300 python bench.py --steps 20 --no-cpu > gpurun_out/bench1.json 2> gpurun_out/bench1.err || { tail -20 gpurun_out/bench1.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench1.json"))
print(round(d["value"]), d["roofline"]["encode_ms"], d["roofline"]["decode_ms"], d["roofline"]["frac"], d["roofline"]["traffic"])
print(d["rle_variant"])
PY
