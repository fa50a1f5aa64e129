#!/bin/bash
# round 4: request distance of the LDS-direct GEMM's operand tiles -- GPMP_GEMM_EARLY_ISSUE = -1 (tile kt + 1 at the top of tile
# kt: 48 MFMAs of cover), 1 (tile kt + 2 right behind the barrier of tile kt: 64), 0 = shipped: early for K >= GPMP_GEMM_EARLY_MIN_K
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
GPMP_GEMM_EARLY_ISSUE=1 timeout -k 10 500 python -m pytest tests/test_hip_parity.py tests/test_switches_gpu.py -x -q -m gpu -k "cholesky or potrf or gemm or matmul or solve or predict" 2>&1 | tail -2
hipcc -O2 --offload-arch=gfx950 -Iinclude tools/gemm_bench.cpp -Lgpmp_amd -lgpmp_hip -Wl,-rpath,$R/gpmp_amd -o tools/gemm_bench.bin || exit 1
for e in -1 1 -1 1 -1 1; do echo "EARLY=$e"; GPMP_GEMM_EARLY_ISSUE=$e ./tools/gemm_bench.bin 3 50 | cut -c1-40,95-; GPMP_GEMM_EARLY_ISSUE=$e ./tools/gemm_bench.bin 5 51 | cut -c1-40,95-; GPMP_GEMM_EARLY_ISSUE=$e ./tools/gemm_bench.bin 5 0 | cut -c1-40,95-; done
timeout -k 10 300 python tools/potrf_ab.py GPMP_GEMM_EARLY_ISSUE -1 1 16384 32768
timeout -k 10 300 python tools/potrf_ab.py GPMP_GEMM_EARLY_ISSUE -1 0 16384 32768
