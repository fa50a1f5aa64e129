#!/bin/bash
# round 4, fifth GPU pass: the final build -- bench line, rocprofv3 stats / PMC passes (v22), the whole GPU suite.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out/prof_r4
export TMPDIR=/tmp
timeout -k 10 900 python bench.py > gpurun_out/r4_bench_c.log 2> gpurun_out/r4_bench_c.err
echo "bench rc=$?" | tee -a gpurun_out/r4_bench_c.err
tail -c 3000 gpurun_out/r4_bench_c.log
timeout -k 10 900 bash tools/profile_r4.sh v22 > gpurun_out/r4_profile_v22.log 2>&1
echo "profile rc=$?"; tail -4 gpurun_out/r4_profile_v22.log
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4_gpu_suite3.log 2>&1
echo "suite rc=$?"; tail -3 gpurun_out/r4_gpu_suite3.log
