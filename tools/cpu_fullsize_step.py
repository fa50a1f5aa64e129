#!/usr/bin/env python3
"""ONE full-size CPU run of the headline step (oracle = the reference's NumPy / SciPy call sequence, all host cores) beside
the figure bench.py ASSEMBLES from a bounded sample (`cpu_baseline`, `"assembled": true`): how far off is the model?
~8-10 minutes of host time, ~110 GB of host memory at n = 32768, m = 50000; no GPU.  Log: profiles/r3/cpu_fullsize_step.log

    python tools/cpu_fullsize_step.py [n] [m]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from oracle import gp_oracle as orc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
m = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
d = 8
threads = bench._host_threads()
from threadpoolctl import threadpool_limits  # noqa: E402

_limit = threadpool_limits(limits=threads)        # one BLAS thread per usable core, as bench.cpu_baseline
model_fig = bench.cpu_baseline(n, m, d, threads)
print(json.dumps({"assembled": {k: model_fig[k] for k in ("value", "measured_s", "extrapolated_s", "cpu_trsm_tflops")}}), flush=True)
xi, zi, xt, theta = bench.synth(n, m, d, 0)
om = orc.OracleModel(None, lambda x, y, t, pairwise=False: orc.maternp_covariance(x, y, 2, t, pairwise), None, theta, "zero")
t0 = time.perf_counter()
zpm, zpv = orc.predict(om, xi, zi, xt)
t1 = time.perf_counter()
print(json.dumps({"predict_s": t1 - t0}), flush=True)
nll = orc.negative_log_likelihood_zero_mean(om, theta, xi, zi)
t2 = time.perf_counter()
full = t2 - t0
assembled = m / model_fig["value"]
print(json.dumps({"tool": "cpu_fullsize_step", "n": n, "m": m, "d": d, "threads": threads, "predict_s": t1 - t0, "nll_s": t2 - t1,
                  "full_step_s": full, "points_per_s": m / full, "assembled_step_s": assembled, "assembled_points_per_s": model_fig["value"],
                  "assembled_over_measured": assembled / full, "nll": float(nll)}))
