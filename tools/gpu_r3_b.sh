# round 3: whole GPU suite, default bench line, strips mode at N = 1 and the N = 2 rehearsal
set -o pipefail
mkdir -p gpurun_out
bash tools/gpu_tests.sh || exit 1
timeout -k 10 600 python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err || { tail -20 gpurun_out/r03_bench.err; exit 1; }
cut -c1-3000 gpurun_out/r03_bench.json
timeout -k 10 300 python bench.py --mode strips --steps 10 > gpurun_out/r03_strips_n1.json 2> gpurun_out/r03_strips_n1.err || { tail -20 gpurun_out/r03_strips_n1.err; exit 1; }
cat gpurun_out/r03_strips_n1.json
timeout -k 10 300 python bench.py --mode strips --gpus 2 --steps 10 > gpurun_out/r03_strips_n2_rehearsal.json 2> gpurun_out/r03_strips_n2.err || { tail -20 gpurun_out/r03_strips_n2.err; exit 1; }
cat gpurun_out/r03_strips_n2_rehearsal.json
