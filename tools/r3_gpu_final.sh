#!/bin/bash
# round 3, final pass with the final binary: default bench line (live traffic, CPU baseline), then the committed profile passes (v18)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python bench.py > gpurun_out/r3_bench_d.log 2> gpurun_out/r3_bench_d.err
echo "bench rc=$?"
python - <<'PY'
import json
j = json.loads([x for x in open("gpurun_out/r3_bench_d.log") if x.startswith("{")][-1])
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["traffic"], j["extra"]["potrf"]["ms"], j["extra"]["config2"]["ms_per_step"], j["extra"]["config4"]["ms_per_value_and_gradient"])
print(json.dumps(j["cpu_baseline"])[:1200])
PY
bash tools/profile_r3.sh v18 > gpurun_out/r3_profile_v18.log 2>&1
echo "profile rc=$?"; tail -3 gpurun_out/r3_profile_v18.log
