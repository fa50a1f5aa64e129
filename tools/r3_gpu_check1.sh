#!/bin/bash
# round 3, first GPU pass: the new tests, the N = 2 launcher rehearsal over gloo (two ranks share the one GPU), the default bench line
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_batch_driver_gpu.py tests/test_hip_parity.py -x -q -m gpu -k "batch or stream_release or factor_and_solve or eight_streams" > gpurun_out/r3_newtests.log 2>&1
echo "new tests rc=$?" | tee -a gpurun_out/r3_newtests.log
tail -5 gpurun_out/r3_newtests.log
GPMP_BENCH_BACKEND=gloo GPMP_BENCH_DIST_N=16384 timeout -k 10 900 python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r3_bench_n2_gloo.log 2> gpurun_out/r3_bench_n2_gloo.err
echo "bench n2 gloo rc=$?" | tee -a gpurun_out/r3_bench_n2_gloo.err
tail -c 3000 gpurun_out/r3_bench_n2_gloo.log
timeout -k 10 900 python bench.py > gpurun_out/r3_bench_a.log 2> gpurun_out/r3_bench_a.err
echo "bench rc=$?" | tee -a gpurun_out/r3_bench_a.err
tail -c 6000 gpurun_out/r3_bench_a.log
