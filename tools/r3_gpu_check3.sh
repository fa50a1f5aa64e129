#!/bin/bash
# round 3, third GPU pass: the default bench line with roofline.traffic measured live, then the committed profile passes (v17) while
# the one full-size CPU run of the headline step runs on the host cores beside the counter passes (counters are not timing-critical)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python bench.py > gpurun_out/r3_bench_b.log 2> gpurun_out/r3_bench_b.err
echo "bench rc=$?"
python - <<'PY'
import json
l = [x for x in open("gpurun_out/r3_bench_b.log") if x.startswith("{")][-1]
j = json.loads(l)
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["traffic"], j["roofline"]["traffic_note"][:200])
PY
rm -f /tmp/gpmp_stats_pass_done
# the CPU run starts only when the (timing-critical) stats pass is over; it then shares the host with the counter passes
( while [ ! -e /tmp/gpmp_stats_pass_done ]; do sleep 2; done; python tools/cpu_fullsize_step.py > gpurun_out/r3_cpu_fullsize_step.log 2>&1; echo "cpu full rc=$?" >> gpurun_out/r3_cpu_fullsize_step.log ) &
CPU_PID=$!
bash tools/profile_r3.sh v17 > gpurun_out/r3_profile.log 2>&1
echo "profile rc=$?"; tail -3 gpurun_out/r3_profile.log
wait $CPU_PID
tail -2 gpurun_out/r3_cpu_fullsize_step.log
