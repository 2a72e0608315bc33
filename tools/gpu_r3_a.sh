# round 3, first GPU call: the new parity tests, then valu vs xdl on one box
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 180 --timeout-method thread -k "variant or persistent_grid or matrix" > gpurun_out/t_new.log 2>&1 || { tail -40 gpurun_out/t_new.log; exit 1; }
tail -3 gpurun_out/t_new.log
bash tools/gpu_ab.sh JPEGAMD_TRANSFORM=valu JPEGAMD_TRANSFORM=xdl JPEGAMD_TRANSFORM=valu JPEGAMD_TRANSFORM=xdl
