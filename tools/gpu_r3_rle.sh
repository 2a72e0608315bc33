# round 3: run-length tests, then the bench line's rle_variant and a kernel trace of the chain
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_rle.py tests/test_zz_no_alloc.py tests/test_shard.py tests/test_gpu_pipeline.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/t_rle.log 2>&1 || { tail -40 gpurun_out/t_rle.log; exit 1; }
tail -2 gpurun_out/t_rle.log
bash tools/gpu_ab.sh JPEGAMD_TRANSFORM=valu || exit 1
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_rle
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_rle -- python3 /root/repo/tools/prof_run_stages.py 8192 > /root/repo/gpurun_out/kt_rle.log 2>&1 || { tail -5 /root/repo/gpurun_out/kt_rle.log; exit 1; }
grep "rle_\|rld_" "$(find /tmp/kt_rle -name '*kernel_stats.csv' | head -1)" | cut -c1-200
