set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_rle.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/t1.log 2>&1 || { tail -40 gpurun_out/t1.log; exit 1; }
tail -3 gpurun_out/t1.log
timeout -k 10 60 ./tools/fdct_ubench || exit 1
bash tools/gpu_ab.sh X=0
