# tools/partial_tile_probe.py under several library builds (tools/ab/lib_<name>.so), alternating
set -o pipefail
names=$1; shift
for rep in 1 2; do
for name in $names; do
  cp tools/ab/lib_$name.so jpeg_amd/libjpeg_amd.so || exit 1
  echo "== $name"
  timeout -k 10 300 python tools/partial_tile_probe.py "$@" || exit 1
done
done
