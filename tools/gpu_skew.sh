set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_shard.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/t1.log 2>&1 || { tail -30 gpurun_out/t1.log; exit 1; }
tail -2 gpurun_out/t1.log
for rep in 1 2; do for name in head skew; do
  cp tools/ab/lib_$name.so jpeg_amd/libjpeg_amd.so || exit 1
  echo $name; timeout -k 10 300 python tools/batch_launch_bench.py 64 2160 3840 | tail -1 || exit 1
done; done
bash tools/gpu_ab_lib.sh head skew
