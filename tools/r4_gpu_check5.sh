#!/bin/bash
# round 4, fifth GPU pass: the final build -- bench line, rocprofv3 stats / PMC passes (v23), the whole GPU suite.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out/prof_r4
export TMPDIR=/tmp
timeout -k 10 900 python bench.py > gpurun_out/r4_bench_d.log 2> gpurun_out/r4_bench_d.err
echo "bench rc=$?" | tee -a gpurun_out/r4_bench_d.err
tail -c 3000 gpurun_out/r4_bench_d.log
timeout -k 10 900 bash tools/profile_r4.sh v23 > gpurun_out/r4_profile_v23.log 2>&1
echo "profile rc=$?"; tail -4 gpurun_out/r4_profile_v23.log
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4_gpu_suite4.log 2>&1
echo "suite rc=$?"; tail -3 gpurun_out/r4_gpu_suite4.log
