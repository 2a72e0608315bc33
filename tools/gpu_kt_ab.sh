set -o pipefail
cd /tmp && export TMPDIR=/tmp
export JPEGAMD_TIMING_ONLY=1
for name in base nostore; do
  cp /root/repo/tools/ab/lib_$name.so /root/repo/jpeg_amd/libjpeg_amd.so
  rm -rf /tmp/kt_$name
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$name -- python3 /root/repo/tools/prof_run_stages.py 8192 > /root/repo/gpurun_out/kt_$name.log 2>&1 || { tail -5 /root/repo/gpurun_out/kt_$name.log; exit 1; }
  echo "== $name"
  grep "hd_" "$(find /tmp/kt_$name -name '*kernel_stats.csv' | head -1)" | sed 's/(anonymous namespace):://g' | cut -c1-40,60-200
done
