# SURVEY 8d configs 2-5 beside the headline: size sweep, quality sweep, 4K batch, photograph-like input -> gpurun_out/sweeps/
set -o pipefail
OUT=gpurun_out/sweeps
mkdir -p $OUT
timeout -k 10 500 python tools/size_sweep.py > $OUT/size_sweep.txt 2> $OUT/size_sweep.err || { tail -5 $OUT/size_sweep.err; exit 1; }
timeout -k 10 500 python tools/qf_sweep.py > $OUT/qf_sweep.txt 2> $OUT/qf_sweep.err || { tail -5 $OUT/qf_sweep.err; exit 1; }
timeout -k 10 300 python tools/batch_launch_bench.py 64 2160 3840 > $OUT/batch_4k.txt 2> $OUT/batch_4k.err || { tail -5 $OUT/batch_4k.err; exit 1; }
timeout -k 10 300 python tools/batch_launch_bench.py 64 1024 1024 > $OUT/batch_1k.txt 2> $OUT/batch_1k.err || { tail -5 $OUT/batch_1k.err; exit 1; }
timeout -k 10 300 python tools/photo_step_bench.py > $OUT/photo.txt 2> $OUT/photo.err || { tail -5 $OUT/photo.err; exit 1; }
tail -4 $OUT/size_sweep.txt; tail -4 $OUT/qf_sweep.txt; cat $OUT/batch_4k.txt $OUT/batch_1k.txt $OUT/photo.txt
