# A/B of library builds x environment settings on one box: arguments "name:ENV=val" (tools/ab/lib_<name>.so)
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2 3; do
for item in "$@"; do
  name=${item%%:*}; cfg=${item#*:}
  cp tools/ab/lib_$name.so jpeg_amd/libjpeg_amd.so || exit 1
  env $cfg timeout -k 10 300 python bench.py --steps 20 --no-cpu --headline-only > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -20 gpurun_out/ab.err; exit 1; }
  python - "$item" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab.json"))
print(sys.argv[1], round(d["value"]), "step %.4f" % d["ms_per_step"], "enc %.4f dec %.4f" % (d["roofline"]["encode_ms"], d["roofline"]["decode_ms"]))
PY
done
done
