#!/usr/bin/env python3
"""REML parameter selection and prediction at n beyond one GPU's HBM -- the flow of the reference's example07
(gpmp/examples/gpmp_example07_nd_regression.py: noisy observations, constant mean, REML fit, prediction) on the 2-D
block-cyclic factor: one process per GPU, RCCL collectives owned by torch.distributed, local arithmetic in libgpmp_hip.so.

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/dist_reml_fit.py --size-n 131072
    python examples/dist_reml_fit.py --size-n 6000          (one GPU: a 1 x 1 grid; GPMP_EXAMPLE_BACKEND=gloo shares one GPU between ranks)
"""
import argparse
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size-n", dest="n", type=int, default=6000)
    ap.add_argument("--size-m", dest="m", type=int, default=2000)
    ap.add_argument("--dim-d", dest="d", type=int, default=4)
    ap.add_argument("--block", type=int, default=1024)
    ap.add_argument("--maxiter", type=int, default=20)
    a = ap.parse_args()
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29622")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("GPMP_EXAMPLE_BACKEND", "nccl")
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    os.environ["LOCAL_RANK"] = str(local)
    torch.cuda.set_device(local)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)

    from gpmp_amd.dist import BlockCyclicCholesky, ProcessGrid, fit_covparam
    from gpmp_amd.kernel import MaternCovariance

    n, m, d = a.n, a.m, a.d
    rng = np.random.default_rng(1234)                       # the same data on every rank (replicated inputs)
    x, xt = rng.random((n, d)), rng.random((m, d))
    f = lambda u: np.sin(2 * np.pi * u[:, 0]) + u[:, 1:].sum(axis=1)        # noqa: E731
    z = f(x) + 0.1 * rng.standard_normal(n)
    cov = MaternCovariance(2, noise=True)                   # theta = [log s2, log s2_noise, log 1/rho_1..d]
    P, Pt = np.ones((n, 1)), np.ones((m, 1))                # constant mean (linear predictor)
    th0 = np.concatenate(([0.0, math.log(0.05)], np.zeros(d)))
    grid = ProcessGrid(*ProcessGrid.default_shape(world))
    th, info = fit_covparam(grid, cov, x, z, th0, P=P, nb=a.block, options={"maxiter": a.maxiter})
    ch = BlockCyclicCholesky(grid, n, nb=a.block)
    ch.build_local_gram(cov, x, th, math.exp(th[1]))
    assert ch.factor() == 0
    mean, var, (j0, j1) = ch.predict(cov, x, z, xt, th, P=P, Pt=Pt)
    err = np.array([np.sum((mean - f(xt[j0:j1])) ** 2), j1 - j0], dtype=np.float64) if grid.r == 0 else np.zeros(2)
    t = torch.as_tensor(err)
    t = t.cuda() if backend == "nccl" else t
    dist.all_reduce(t)
    if rank == 0:
        rmse = math.sqrt(float(t[0]) / float(t[1]))
        print(f"dist_reml_fit: n={n} grid={grid.pr}x{grid.pc} evaluations={info['nfev']} reml={info['fun']:.6f} "
              f"sigma2={math.exp(th[0]):.4f} noise_sd={math.exp(0.5 * th[1]):.4f} rho={np.round(np.exp(-th[2:]), 3).tolist()} rmse={rmse:.4f}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
