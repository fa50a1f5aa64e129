set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_config_sizes_gpu.py tests/test_fullsize_golden_gpu.py tests/test_plumbing_gpu.py -x -q -m gpu -k "solve or predict or trsm or config2 or config3 or matmul or gemm or leaves" 2>&1 | tail -3
hipcc -O2 --offload-arch=gfx950 -Iinclude tools/gemm_bench.cpp -Lgpmp_amd -lgpmp_hip -Wl,-rpath,$R/gpmp_amd -o tools/gemm_bench.bin || exit 1
for f in 0 1 0 1; do echo "FIT_N=$f"; GPMP_GEMM_FIT_N=$f ./tools/gemm_bench.bin 10 53 | cut -c1-40,60-72,95-; done
timeout -k 10 300 python tools/predict_ab.py GPMP_GEMM_FIT_N 0 1 50000 32768 16384 8192
timeout -k 10 300 python tools/predict_ab.py GPMP_GEMM_FIT_N 0 1 40000 32768 8192
