#!/bin/bash
# round 3, final pass with the final binary: full GPU suite, default bench line (live traffic, CPU baseline), committed profile passes
set -o pipefail
V=${1:-v19}
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r3_full_gpu_final.log 2>&1
echo "gpu suite rc=$?"; tail -2 gpurun_out/r3_full_gpu_final.log
timeout -k 10 1000 python bench.py > gpurun_out/r3_bench_final.log 2> gpurun_out/r3_bench_final.err
echo "bench rc=$?"
python - <<'PY'
import json
j = json.loads([x for x in open("gpurun_out/r3_bench_final.log") if x.startswith("{")][-1])
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["traffic"], j["extra"]["potrf"]["ms"], j["extra"]["config2"]["ms_per_step"], j["extra"]["config4"]["ms_per_value_and_gradient"], j["cpu_baseline"]["value"])
PY
bash tools/profile_r3.sh $V > gpurun_out/r3_profile_$V.log 2>&1
echo "profile rc=$?"; tail -3 gpurun_out/r3_profile_$V.log
