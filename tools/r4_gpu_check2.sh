#!/bin/bash
# round 4, second GPU pass: full-size golden tests (configs 3 / 4 against the committed vectors), fabric-side traffic of the
# solve-update GEMM under three tile-group heights (FETCH_SIZE / WRITE_SIZE passes), kernel stats of the Cholesky with and
# without two-level panels, one rank's value + gradient share again (device-resident scalars), the N = 2 rehearsal of
# bench.py over gloo on the one GPU (ranks_seen, strong_scaling_block_cyclic with the real kernels).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out/prof_r4
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_fullsize_golden_gpu.py -x -q -m gpu ${GOLDEN_K:+-k $GOLDEN_K} > gpurun_out/r4_fullsize_golden.log 2>&1
echo "fullsize golden rc=$?" | tee -a gpurun_out/r4_fullsize_golden.log
tail -15 gpurun_out/r4_fullsize_golden.log
hipcc -O2 --offload-arch=gfx950 -Iinclude tools/gemm_bench.cpp -Lgpmp_amd -lgpmp_hip -Wl,-rpath,$R/gpmp_amd -o tools/gemm_bench.bin || exit 1
cd /tmp
for gm in 8 4 16; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/p_gm
    GPMP_GEMM_GM=$gm timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d /tmp/p_gm -- $R/tools/gemm_bench.bin 1 52 > /dev/null 2>&1 || { echo "pmc pass failed gm=$gm $c"; exit 1; }
    echo "GPMP_GEMM_GM=$gm $c (KB raw; FETCH_SIZE x2 on gfx950)" >> $R/gpurun_out/prof_r4/gemm_solve_update_traffic_by_gm.txt
    python3 $R/tools/pmc_by_kernel.py $(ls /tmp/p_gm/*/*counter_collection.csv | head -1) $c | grep -i "gemm\|kernel" >> $R/gpurun_out/prof_r4/gemm_solve_update_traffic_by_gm.txt
  done
done
cat $R/gpurun_out/prof_r4/gemm_solve_update_traffic_by_gm.txt
for sup in 0 16384; do
  rm -rf /tmp/p_potrf
  GPMP_POTRF_SUPER_ABOVE=$sup timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_potrf -- python3 $R/tools/potrf_only.py 32768 > $R/gpurun_out/prof_r4/potrf_n32768_super${sup}.log 2>&1 || { echo "potrf stats failed"; exit 1; }
  cp $(ls /tmp/p_potrf/*/*kernel_stats.csv | head -1) $R/gpurun_out/prof_r4/potrf_n32768_super${sup}_kernel_stats.csv
done
head -8 $R/gpurun_out/prof_r4/potrf_n32768_super0_kernel_stats.csv; head -8 $R/gpurun_out/prof_r4/potrf_n32768_super16384_kernel_stats.csv
cd $R
timeout -k 10 400 python tools/dist_rank_emulation.py --size-n 131072 --grid 2x4 --coords 0,0 --grad > gpurun_out/r4_rank_emulation_grad2.log 2>&1
echo "emulation grad rc=$?"; tail -c 1200 gpurun_out/r4_rank_emulation_grad2.log
GPMP_BENCH_BACKEND=gloo GPMP_BENCH_DIST_N=16384 timeout -k 10 900 python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r4_bench_n2_gloo.log 2> gpurun_out/r4_bench_n2_gloo.err
echo "bench n2 gloo rc=$?" | tee -a gpurun_out/r4_bench_n2_gloo.err
tail -c 6000 gpurun_out/r4_bench_n2_gloo.log
tail -5 gpurun_out/r4_bench_n2_gloo.err
