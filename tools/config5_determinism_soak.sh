export PYTHONPATH=$PWD
mkdir -p gpurun_out/r5kk
for i in 1 2 3; do
  timeout -k 10 300 python tools/config5_full.py dist --threads --device-comm --grid 2x4 --grad-n 32768 --limit 250 --dist-out /tmp/d$i.npz > gpurun_out/r5kk/run$i.log 2>&1 || exit 1
  grep "factor:" gpurun_out/r5kk/run$i.log | cut -c1-160
done
python - <<PY
import numpy as np
a=[np.load(f"/tmp/d{i}.npz") for i in (1,2,3)]
keys=["nll","logdet","reml","L_sample","mean","var","lam_sample","uk_mean","uk_var","grad_value","grad"]
for k in keys:
    same=all(np.array_equal(a[0][k], x[k], equal_nan=True) for x in a[1:])
    print(k, "bit-identical over 3 runs:", same)
PY
