#!/bin/bash
# round 4, first GPU pass: new tests (plumbing, per-device state, switches incl. two-level panels, distributed gradient with the
# staircase lauum), the potrf A/B of the two-level panels, the GEMM tile-order experiment (time; FETCH_SIZE in check2), one
# rank's share of config 5 (value + gradient) and of the headline step on the block-cyclic factor.
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_plumbing_gpu.py tests/test_switches_gpu.py -x -q -m gpu > gpurun_out/r4_newtests.log 2>&1
echo "new tests rc=$?" | tee -a gpurun_out/r4_newtests.log
tail -4 gpurun_out/r4_newtests.log
timeout -k 10 500 python -m pytest tests/test_dist_gpu.py -x -q -m gpu -k "grad or model or reml" > gpurun_out/r4_dist_tests.log 2>&1
echo "dist tests rc=$?" | tee -a gpurun_out/r4_dist_tests.log
tail -4 gpurun_out/r4_dist_tests.log
for v in 8192 16384 24576; do
  timeout -k 10 300 python tools/potrf_ab.py GPMP_POTRF_SUPER_ABOVE 0 $v 16384 32768 >> gpurun_out/r4_potrf_super_ab.log 2>&1 || exit 1
done
cat gpurun_out/r4_potrf_super_ab.log
hipcc -O2 --offload-arch=gfx950 -Iinclude tools/gemm_bench.cpp -Lgpmp_amd -lgpmp_hip -Wl,-rpath,$PWD/gpmp_amd -o tools/gemm_bench.bin || exit 1
for gm in 8 4 16; do
  echo "GPMP_GEMM_GM=$gm" >> gpurun_out/r4_gemm_gm.log
  GPMP_GEMM_GM=$gm timeout -k 10 200 ./tools/gemm_bench.bin 3 50 >> gpurun_out/r4_gemm_gm.log 2>&1 || exit 1
  GPMP_GEMM_GM=$gm timeout -k 10 200 ./tools/gemm_bench.bin 3 51 >> gpurun_out/r4_gemm_gm.log 2>&1 || exit 1
done
cat gpurun_out/r4_gemm_gm.log
timeout -k 10 400 python tools/dist_rank_emulation.py --size-n 131072 --grid 2x4 --coords 0,0 --grad > gpurun_out/r4_rank_emulation_grad.log 2>&1
echo "emulation grad rc=$?"; tail -c 2500 gpurun_out/r4_rank_emulation_grad.log
timeout -k 10 300 python tools/dist_rank_emulation.py --size-n 32768 --grid 2x4 --coords 0,0 --step-m 50000 > gpurun_out/r4_rank_emulation_step.log 2>&1
echo "emulation step rc=$?"; tail -c 2500 gpurun_out/r4_rank_emulation_step.log
