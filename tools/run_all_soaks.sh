#!/bin/bash
# Every opt-in random soak of the GPU suite in one go (on the GPU box through gpurun): `bash tools/run_all_soaks.sh SEED [SCALE]`.
# SCALE multiplies the draw counts (1 = about 10 minutes).  Each soak is an environment variable on a test of the suite
# ("Opt-in random soaks"); one summary line per soak, full output under gpurun_out/soaks/.
set -u
SEED=${1:-101}
S=${2:-1}
OUT=${GRAFT_REPO_ROOT:-.}/gpurun_out/soaks
mkdir -p $OUT
run() {   # name, env assignments..., -- pytest args
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" timeout -k 10 900 python3 -m pytest "$@" -m gpu -q -s > $OUT/$name.log 2>&1
  echo "$name: rc=$? $(tail -1 $OUT/$name.log)"
}
run model_sweep   GPMP_SWEEP_SEED=$SEED GPMP_SWEEP_CASES=$((240*S)) -- tests/test_random_sweep_gpu.py
run linalg        GPMP_LINALG_SOAK_SEED=$SEED GPMP_LINALG_SOAK_CASES=$((100*S)) -- tests/test_blocked_algorithms_gpu.py -k "random_soak and not c_abi"
run linalg_abi    GPMP_LINALG_ABI_SOAK_SEED=$SEED GPMP_LINALG_ABI_SOAK_CASES=$((100*S)) -- tests/test_blocked_algorithms_gpu.py -k c_abi
run batch         GPMP_BATCH_SOAK_SEED=$SEED GPMP_BATCH_SOAK_CASES=$((500*S)) -- tests/test_batch_driver_gpu.py -k random_soak
run drivers       GPMP_DRIVER_SOAK_SEED=$SEED GPMP_DRIVER_SOAK_CASES=$((300*S)) -- tests/test_c_abi_mean_drivers_gpu.py -k random_soak
run drivers_wide  GPMP_DRIVER_SOAK_WIDE=1 GPMP_DRIVER_SOAK_SEED=$SEED GPMP_DRIVER_SOAK_CASES=$((150*S)) -- tests/test_c_abi_mean_drivers_gpu.py -k random_soak
run dist_fabric   GPMP_DIST_SOAK_SEED=$SEED GPMP_DIST_SOAK_CASES=$((150*S)) -- tests/test_dist_gpu.py -k random_soak
run gram          GPMP_GRAM_SOAK_SEED=$SEED GPMP_GRAM_SOAK_CASES=$((500*S)) -- tests/test_hip_parity.py -k gram_entry_points_random_soak
run gemm          GPMP_GEMM_SOAK_SEED=$SEED GPMP_GEMM_SOAK_CASES=$((500*S)) -- tests/test_hip_parity.py -k gemm_entry_point_random_soak
