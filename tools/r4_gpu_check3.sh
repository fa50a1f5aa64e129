#!/bin/bash
# round 4, third GPU pass: config 3 at full size against the oracle vector (+ the config 4 vectors again), the kriging weights on
# the block-cyclic factor with the real kernels, ONE full-size CPU run of the headline step on this box's host (for
# cpu_baseline.model_over_measured), the default bench line, the rocprofv3 stats / PMC passes of the same command.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out/prof_r4
export TMPDIR=/tmp
# the config 3 vector: the pinned oracle at full size on THIS box's host cores (no GPU involved), then the tests against it
timeout -k 10 1100 python tests/golden/make_oracle_config3.py gpurun_out/oracle_config3_n32768.npz > gpurun_out/r4_make_oracle_config3.log 2>&1
echo "oracle config3 rc=$?"; tail -12 gpurun_out/r4_make_oracle_config3.log
cp gpurun_out/oracle_config3_n32768.npz tests/golden/ || exit 1
timeout -k 10 900 python -m pytest tests/test_fullsize_golden_gpu.py tests/test_dist_gpu.py -x -q -m gpu -k "config or model_surface" > gpurun_out/r4_fullsize_golden2.log 2>&1
echo "fullsize golden + dist model rc=$?" | tee -a gpurun_out/r4_fullsize_golden2.log
tail -15 gpurun_out/r4_fullsize_golden2.log
timeout -k 10 900 python tools/cpu_fullsize_step.py > gpurun_out/r4_cpu_fullsize_step.log 2>&1
echo "cpu full rc=$?"; tail -2 gpurun_out/r4_cpu_fullsize_step.log
mkdir -p profiles/r4 && grep '^{' gpurun_out/r4_cpu_fullsize_step.log > profiles/r4/cpu_fullsize_step.log     # the bench run below reads it
timeout -k 10 900 python bench.py > gpurun_out/r4_bench_a.log 2> gpurun_out/r4_bench_a.err
echo "bench rc=$?" | tee -a gpurun_out/r4_bench_a.err
tail -c 7000 gpurun_out/r4_bench_a.log
timeout -k 10 900 bash tools/profile_r4.sh v20 > gpurun_out/r4_profile.log 2>&1
echo "profile rc=$?"; tail -5 gpurun_out/r4_profile.log
