set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_rle.py tests/wip_no_alloc.py -x -v --timeout 120 --timeout-method thread > gpurun_out/t1.log 2>&1 || { tail -30 gpurun_out/t1.log; exit 1; }
tail -3 gpurun_out/t1.log
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/t2.log 2>&1 || { tail -40 gpurun_out/t2.log; exit 1; }
tail -3 gpurun_out/t2.log
timeout -k 10 300 python bench.py --steps 20 > gpurun_out/bench1.json 2> gpurun_out/bench1.err || { tail -20 gpurun_out/bench1.err; exit 1; }
cat gpurun_out/bench1.json
timeout -k 10 200 python bench.py --gpus 2 --no-cpu --steps 10 > gpurun_out/bench2.json 2> gpurun_out/bench2.err || { tail -20 gpurun_out/bench2.err; exit 1; }
cat gpurun_out/bench2.json
