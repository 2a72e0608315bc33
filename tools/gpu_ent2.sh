set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_entropy.py tests/test_gpu_entropy_decode.py tests/test_gpu_pipeline.py tests/test_harness.py tests/test_shard.py -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/t1.log 2>&1 || { tail -40 gpurun_out/t1.log; exit 1; }
tail -3 gpurun_out/t1.log
JPEGAMD_TRACE=1 timeout -k 10 300 python tools/pipeline_repeat.py > gpurun_out/pipe.log 2>&1 || { tail -20 gpurun_out/pipe.log; exit 1; }
tail -12 gpurun_out/pipe.log
