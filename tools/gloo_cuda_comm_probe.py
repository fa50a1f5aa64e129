#!/usr/bin/env python3
"""Do the DEVICE-RESIDENT communication branches of gpmp_amd/dist (what runs under RCCL: comm tensors stay on the GPU, scalars
and the few-column solve on the device, collectives called with CUDA tensors from the three streams) work end to end?  RCCL
refuses two ranks on one GPU, but gloo moves CUDA tensors too -- so: pr x pc ranks share the one GPU over gloo with
``backend`` forced to the RCCL code path, and factorisation, NLL, prediction with weights, leave-one-out and the REML value +
gradient are compared with the oracle.  (Stream / ordering semantics of RCCL itself are NOT what this checks.)

    python tools/gloo_cuda_comm_probe.py 2 2 [transport]

Round 5: ``python tools/gloo_cuda_comm_probe.py 1 1 bcast nccl`` runs the SAME entry points through ``ProcessGroupNCCL`` itself --
one rank per GPU, so on this pool's one-GPU boxes the 1 x 1 grid: communicator creation with the high-priority options
(gpmp_amd/dist/grid.py), the four communicator kinds, device-resident comm tensors enqueued from the three streams into RCCL's own
streams, the scalar all-reduces; plus the ``DistributedModel`` surface and a short ``fit_covparam`` run.  With N GPUs the same
command covers pr x pc = N ranks under RCCL.
"""
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port, pr, pc, transport, out, backend="gloo"):
    import torch
    import torch.distributed as dist

    local = rank if backend == "nccl" else 0
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(local),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    if backend == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
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
        grid = ProcessGrid(pr, pc)
        if backend == "nccl":
            assert grid.high_priority_comms, "ProcessGroupNCCL.Options not applied"
        ch = BlockCyclicCholesky(grid, n, nb=nb, ops=HipLocalOps(), transport=transport)
        assert ch.backend == backend
        ch.backend = "nccl"                      # <- device-resident communication, as under RCCL (a no-op when it IS RCCL)
        ch.build_local_gram(cov, x, th, 10.0 * math.exp(th[0]) * gnp.eps)
        info = ch.factor()
        nll = ch.negative_log_likelihood(z)
        reml = ch.negative_log_restricted_likelihood(z, P)
        mean, var, (j0, j1), lam = ch.predict(cov, x, z, xt, th, P=P, Pt=Pt, return_lambdas=True)
        zloo, s2, eloo, idx = ch.loo(z, P)
        val, grad = ch.value_and_grad(x, z, th, 2, P=P)
        torch.cuda.synchronize()
        extra = {}
        if backend == "nccl":
            # the Model surface and a short parameter fit through the same process group (every rank makes the same calls)
            from gpmp_amd.dist import DistributedModel, fit_covparam

            mean_fn = lambda a, p: np.hstack((np.ones((len(a), 1)), np.asarray(a)[:, :1]))        # noqa: E731
            model = DistributedModel(grid, mean_fn, cov, None, th, "linear_predictor", nb=nb, transport=transport)
            m_zpm, m_zpv, m_lam = model.predict(x, z, xt, return_lambdas=True)
            m_zloo, _, _ = model.loo(x, z)
            m_reml = model.negative_log_restricted_likelihood(th, x, z)
            noisy = MaternCovariance(2, noise=True)
            th_n = np.concatenate(([th[0], math.log(1e-3)], th[1:]))
            chn = BlockCyclicCholesky(grid, n, nb=nb, ops=HipLocalOps(), transport=transport)
            chn.build_local_gram(noisy, x, th_n, 1e-3)
            assert chn.factor() == 0
            v_n, g_n = chn.value_and_grad(x, z, th_n, 2, noise=True)
            th_fit, fit = fit_covparam(grid, cov, x, z, th + 0.3, P=P, options={"maxiter": 3}, nb=nb, ops=HipLocalOps(), transport=transport)
            torch.cuda.synchronize()
            extra = dict(m_zpm=m_zpm, m_zpv=m_zpv, m_lam=m_lam, m_zloo=m_zloo, m_reml=m_reml, v_n=v_n, g_n=g_n,
                         fit_first=fit["history"][0][1], fit_last=fit["fun"], fit_nfev=fit["nfev"])
        parts = [None] * world
        dist.all_gather_object(parts, (ch.grid.r, j0, j1, mean, var, ch.global_row_index(), lam.cpu().numpy(), idx, zloo))
        if rank == 0:
            zpm, zpv, L, zl = np.full(m, np.nan), np.full(m, np.nan), np.full((n, m), np.nan), np.full(n, np.nan)
            for (r, a, b, mu, v, rows, blk, ix, zz) in parts:
                zpm[a:b], zpv[a:b] = mu, v
                if len(rows) and b > a:
                    L[np.ix_(rows, np.arange(a, b))] = blk
                zl[ix] = zz
            np.savez(out, info=info, nll=nll, reml=reml, zpm=zpm, zpv=zpv, lam=L, zloo=zl, val=val, grad=grad, **extra)
    finally:
        dist.destroy_process_group()


def main():
    import socket

    import torch.multiprocessing as mp

    from oracle import gp_oracle as orc
    from tests.helpers import make_xz, theta_aniso

    pr, pc = int(sys.argv[1]), int(sys.argv[2])
    transport = sys.argv[3] if len(sys.argv) > 3 else "bcast"
    backend = sys.argv[4] if len(sys.argv) > 4 else "gloo"
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = "/tmp/gloo_cuda_probe.npz"
    ctx = mp.spawn(worker, args=(pr * pc, port, pr, pc, transport, out, backend), nprocs=pr * pc, join=False)
    deadline = time.monotonic() + 300.0
    while not ctx.join(timeout=2.0):          # bounded: a collective that never completes must not hang the box
        if time.monotonic() > deadline:
            for p in ctx.processes:
                if p.is_alive():
                    p.kill()
            print("DEVICE-COMM PROBE FAILED: workers still running after 300 s (killed)")
            sys.exit(3)
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
    if backend == "nccl":
        errs.update({"model_mean": float(np.max(np.abs(g["m_zpm"] - rm))), "model_var": float(np.max(np.abs(g["m_zpv"] - rv))),
                     "model_lambda_rel": float(np.max(np.abs(g["m_lam"] - rl)) / np.max(np.abs(rl))),
                     "model_zloo": float(np.max(np.abs(g["m_zloo"] - rz))),
                     "model_reml_rel": abs(float(g["m_reml"]) - float(g["reml"])) / abs(float(g["reml"])),
                     "fit_decreased": bool(float(g["fit_last"]) < float(g["fit_first"])), "fit_nfev": int(g["fit_nfev"])})
        th_n = np.concatenate(([th[0], math.log(1e-3)], th[1:]))
        vn, gn = orc.nll_zero_mean_value_and_grad(x, z, 2, th_n, noise_index=1)
        errs["noisy_ml_value_rel"] = abs(float(g["v_n"]) - vn) / abs(vn)
        errs["noisy_ml_grad_rel"] = float(np.max(np.abs(g["g_n"] - gn)) / np.max(np.abs(gn)))
    print(f"grid {pr}x{pc} transport {transport} backend {backend}:", errs)
    if backend == "nccl" and not (errs["model_mean"] < 1e-7 and errs["model_var"] < 1e-7 and errs["model_lambda_rel"] < 1e-6 and errs["model_zloo"] < 1e-6
                                  and errs["model_reml_rel"] < 1e-9 and errs["fit_decreased"] and errs["noisy_ml_value_rel"] < 1e-9
                                  and errs["noisy_ml_grad_rel"] < 1e-7):
        print("DEVICE-COMM PROBE FAILED (model surface / fit / noisy gradient)")
        sys.exit(1)
    ok = errs["info"] == 0 and errs["nll_rel"] < 1e-9 and errs["reml_rel"] < 1e-9 and errs["mean"] < 1e-7 and errs["var"] < 1e-7 \
        and errs["lambda_rel"] < 1e-6 and errs["zloo"] < 1e-6 and errs["value_vs_reml"] < 1e-7 * abs(float(g["reml"])) and errs["grad_rel"] < 1e-7
    print("DEVICE-COMM PROBE", "OK" if ok else "FAILED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
