# round 3: entropy / pipeline tests, then stage times through a context
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_entropy.py tests/test_gpu_entropy_decode.py tests/test_gpu_pipeline.py tests/test_harness.py tests/test_shard.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/t_ent.log 2>&1 || { tail -40 gpurun_out/t_ent.log; exit 1; }
tail -2 gpurun_out/t_ent.log
JPEGAMD_TRACE=1 timeout -k 10 300 python tools/pipeline_repeat.py 2>&1 | tail -16
