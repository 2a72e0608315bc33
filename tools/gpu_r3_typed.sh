# round 3: the typed-load colour stage -- parity suite with it switched on, then A/B against the plain loads
set -o pipefail
mkdir -p gpurun_out
JPEGAMD_LOADS=typed timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_rle.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/t_typed.log 2>&1 || { tail -40 gpurun_out/t_typed.log; exit 1; }
tail -2 gpurun_out/t_typed.log
bash tools/gpu_ab.sh JPEGAMD_LOADS=plain JPEGAMD_LOADS=typed JPEGAMD_LOADS=plain JPEGAMD_LOADS=typed
