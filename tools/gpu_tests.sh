# the whole GPU suite, then the no-alloc test alone in a fresh process
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/t_all.log 2>&1 || { tail -40 gpurun_out/t_all.log; exit 1; }
tail -3 gpurun_out/t_all.log
timeout -k 10 300 python -u -m pytest tests/test_zz_no_alloc.py -x -q --timeout 120 --timeout-method thread > gpurun_out/t_na.log 2>&1 || { tail -40 gpurun_out/t_na.log; exit 1; }
tail -2 gpurun_out/t_na.log
