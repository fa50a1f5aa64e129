#!/bin/bash
# round 3, second GPU pass: distributed tests with the real kernels, the latency-kernel threshold in the Cholesky tail, one rank's
# share of config 5 incl. the many-right-hand-side solve with stubbed collectives
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_dist_gpu.py -x -q -m gpu > gpurun_out/r3_dist_gpu.log 2>&1
echo "dist gpu tests rc=$?"; tail -3 gpurun_out/r3_dist_gpu.log
for mx in 128 256 512 1024; do
  for n in 4096 8192; do
    echo "SMALL_NT_MAX=$mx n=$n" >> gpurun_out/r3_small_nt_max_sweep.log
    GPMP_GEMM_SMALL_NT_MAX=$mx python tools/potrf_ab.py GPMP_POTRF_W128_BELOW 0 0 $n >> gpurun_out/r3_small_nt_max_sweep.log 2>&1
  done
done
cat gpurun_out/r3_small_nt_max_sweep.log
timeout -k 10 600 python tools/dist_rank_emulation.py --size-n 131072 --grid 2x4 --coords 0,0 --solve-m 50000 > gpurun_out/r3_rank_emulation_solve.log 2>&1
echo "emulation rc=$?"; tail -c 3000 gpurun_out/r3_rank_emulation_solve.log
