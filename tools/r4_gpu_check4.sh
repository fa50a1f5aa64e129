#!/bin/bash
# round 4, fourth GPU pass: lower-triangular tile sets in 8 x 8 super-tiles (GPMP_GEMM_TRI_BLOCK) -- correctness, the trailing-update
# shape alone (time + fabric-side traffic), the Cholesky A/B in one process.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out/prof_r4
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_switches_gpu.py tests/test_hip_parity.py -x -q -m gpu -k "cholesky or potrf or lauum or trtri or inverse" > gpurun_out/r4_tri_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r4_tri_tests.log
hipcc -O2 --offload-arch=gfx950 -Iinclude tools/gemm_bench.cpp -Lgpmp_amd -lgpmp_hip -Wl,-rpath,$R/gpmp_amd -o tools/gemm_bench.bin || exit 1
for tb in 0 1; do
  echo "GPMP_GEMM_TRI_BLOCK=$tb" >> gpurun_out/r4_tri_block.log
  GPMP_GEMM_TRI_BLOCK=$tb timeout -k 10 200 ./tools/gemm_bench.bin 5 51 >> gpurun_out/r4_tri_block.log 2>&1 || exit 1
  GPMP_GEMM_TRI_BLOCK=$tb timeout -k 10 200 ./tools/gemm_bench.bin 5 0 >> gpurun_out/r4_tri_block.log 2>&1 || exit 1
done
cat gpurun_out/r4_tri_block.log
timeout -k 10 400 python tools/potrf_ab.py GPMP_GEMM_TRI_BLOCK 0 1 8192 16384 32768 > gpurun_out/r4_potrf_tri_block_ab.log 2>&1 || exit 1
cat gpurun_out/r4_potrf_tri_block_ab.log
cd /tmp
for tb in 0 1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/p_tb
    GPMP_GEMM_TRI_BLOCK=$tb timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d /tmp/p_tb -- $R/tools/gemm_bench.bin 1 51 > /dev/null 2>&1 || { echo "pmc pass failed"; exit 1; }
    echo "GPMP_GEMM_TRI_BLOCK=$tb $c (KB raw; FETCH_SIZE x2 on gfx950; 4 dispatches: K = 1024 and K = 2048, warm-up + 1 each)" >> $R/gpurun_out/prof_r4/gemm_trailing_update_traffic_by_tri_block.txt
    python3 $R/tools/pmc_by_kernel.py $(ls /tmp/p_tb/*/*counter_collection.csv | head -1) $c | grep -i "gemm\|kernel" >> $R/gpurun_out/prof_r4/gemm_trailing_update_traffic_by_tri_block.txt
  done
done
cat $R/gpurun_out/prof_r4/gemm_trailing_update_traffic_by_tri_block.txt
