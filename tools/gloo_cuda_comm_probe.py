#!/usr/bin/env python3
"""Do the DEVICE-RESIDENT communication branches of gpmp_amd/dist (what runs under RCCL: comm tensors stay on the GPU, scalars
and the few-column solve on the device, collectives called with CUDA tensors from the three streams) work end to end?  RCCL
refuses two ranks on one GPU, but gloo moves CUDA tensors too -- so: pr x pc ranks share the one GPU over gloo with
``backend`` forced to the RCCL code path, and factorisation, NLL, prediction with weights, leave-one-out and the REML value +
gradient are compared with the oracle.  (Stream / ordering semantics of RCCL itself are NOT what this checks.)

    python tools/gloo_cuda_comm_probe.py 2 2 [transport]
"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port, pr, pc, transport, out):
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gpmp_amd.num as gnp
        from gpmp_amd.dist import BlockCyclicCholesky, HipLocalOps, ProcessGrid
        from gpmp_amd.kernel import MaternCovariance
        from tests.helpers import make_xz, theta_aniso

        n, m, d, nb = 2100, 333, 4, 256
        x, z = make_xz(n, d, 11)
        xt, _ = make_xz(m, d, 12)
        th = theta_aniso(d, scale=0.5)
        cov = MaternCovariance(2)
        P = np.hstack((np.ones((n, 1)), x[:, :1]))
        Pt = np.hstack((np.ones((m, 1)), xt[:, :1]))
        ch = BlockCyclicCholesky(ProcessGrid(pr, pc), n, nb=nb, ops=HipLocalOps(), transport=transport)
        ch.backend = "nccl"                      # <- device-resident communication, as under RCCL
        ch.build_local_gram(cov, x, th, 10.0 * math.exp(th[0]) * gnp.eps)
        info = ch.factor()
        nll = ch.negative_log_likelihood(z)
        reml = ch.negative_log_restricted_likelihood(z, P)
        mean, var, (j0, j1), lam = ch.predict(cov, x, z, xt, th, P=P, Pt=Pt, return_lambdas=True)
        zloo, s2, eloo, idx = ch.loo(z, P)
        val, grad = ch.value_and_grad(x, z, th, 2, P=P)
        torch.cuda.synchronize()
        parts = [None] * world
        dist.all_gather_object(parts, (ch.grid.r, j0, j1, mean, var, ch.global_row_index(), lam.cpu().numpy(), idx, zloo))
        if rank == 0:
            zpm, zpv, L, zl = np.full(m, np.nan), np.full(m, np.nan), np.full((n, m), np.nan), np.full(n, np.nan)
            for (r, a, b, mu, v, rows, blk, ix, zz) in parts:
                zpm[a:b], zpv[a:b] = mu, v
                if len(rows) and b > a:
                    L[np.ix_(rows, np.arange(a, b))] = blk
                zl[ix] = zz
            np.savez(out, info=info, nll=nll, reml=reml, zpm=zpm, zpv=zpv, lam=L, zloo=zl, val=val, grad=grad)
    finally:
        dist.destroy_process_group()


def main():
    import socket

    import torch.multiprocessing as mp

    from oracle import gp_oracle as orc
    from tests.helpers import make_xz, theta_aniso

    pr, pc = int(sys.argv[1]), int(sys.argv[2])
    transport = sys.argv[3] if len(sys.argv) > 3 else "bcast"
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = "/tmp/gloo_cuda_probe.npz"
    mp.spawn(worker, args=(pr * pc, port, pr, pc, transport, out), nprocs=pr * pc, join=True)
    g = np.load(out)
    n, m, d = 2100, 333, 4
    x, z = make_xz(n, d, 11)
    xt, _ = make_xz(m, d, 12)
    th = theta_aniso(d, scale=0.5)
    mean_fn = lambda a, p: np.hstack((np.ones((len(a), 1)), a[:, :1]))  # noqa: E731
    kern = lambda a, b, t, pairwise=False: orc.maternp_covariance(a, b, 2, t, pairwise)  # noqa: E731
    om = orc.OracleModel(mean_fn, kern, None, th, "linear_predictor")
    oz = orc.OracleModel(None, kern, None, th, "zero")
    rm, rv, rl = orc.predict(om, x, z, xt, return_lambdas=True)
    rz, _, _ = orc.loo(om, x, z)
    errs = {"info": int(g["info"]),
            "nll_rel": abs(float(g["nll"]) - float(orc.negative_log_likelihood_zero_mean(oz, th, x, z))) / abs(float(g["nll"])),
            "reml_rel": abs(float(g["reml"]) - float(orc.negative_log_restricted_likelihood(om, th, x, z))) / abs(float(g["reml"])),
            "mean": float(np.max(np.abs(g["zpm"] - rm))), "var": float(np.max(np.abs(g["zpv"] - rv))),
            "lambda_rel": float(np.max(np.abs(g["lam"] - rl)) / np.max(np.abs(rl))), "zloo": float(np.max(np.abs(g["zloo"] - rz))),
            "value_vs_reml": abs(float(g["val"]) - float(g["reml"])),
            "grad_rel": float(np.max(np.abs(g["grad"] - orc.reml_value_and_grad(x, z, mean_fn(x, None), 2, th)[1])) / np.linalg.norm(g["grad"]))}
    print(f"grid {pr}x{pc} transport {transport}:", errs)
    ok = errs["info"] == 0 and errs["nll_rel"] < 1e-9 and errs["reml_rel"] < 1e-9 and errs["mean"] < 1e-7 and errs["var"] < 1e-7 \
        and errs["lambda_rel"] < 1e-6 and errs["zloo"] < 1e-6 and errs["value_vs_reml"] < 1e-7 * abs(float(g["reml"])) and errs["grad_rel"] < 1e-7
    print("DEVICE-COMM PROBE", "OK" if ok else "FAILED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
