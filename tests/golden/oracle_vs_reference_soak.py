#!/usr/bin/env python3
"""Random soak of the ORACLE against the LIVE REFERENCE (gpmp v0.9.37), beyond the committed fixtures.  Build container only (the
reference does not travel); nothing here is imported by a test.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg PYTHONPATH=/root/reference:/root/repo \
        GPMP_BACKEND=numpy python3 /root/repo/tests/golden/oracle_vs_reference_soak.py numpy 300 5
    ... GPMP_BACKEND=torch python3 /root/repo/tests/golden/oracle_vs_reference_soak.py torch 150 5       # autograd gradients

numpy pass: covariance (ii / it / pairwise, plain and noisy kernel), prediction (mean, variance, weights) for the three mean types,
NLL, REML, leave-one-out -- oracle/gp_oracle.py against the reference's NumPy backend on random (n, m, d, p, parameters).
torch pass: ML / REML values and gradients -- the oracle's analytic gradient against the reference's autograd (every p from 0).
Deviations are reported relative to the SURVEY 8(c) tolerances scaled by cond(K) / 1e6; the script fails on a draw beyond them.
"""
import math
import os
import sys
import warnings

import numpy as np

backend = sys.argv[1] if len(sys.argv) > 1 else "numpy"
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 5
os.environ["GPMP_BACKEND"] = backend
os.environ.setdefault("GPMP_LOG_LEVEL", "WARNING")

import gpmp as gp  # noqa: E402  (the reference)
import gpmp.num as gnp  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import gp_oracle as orc  # noqa: E402

assert gnp._gpmp_backend_ == backend
EPS = np.finfo(float).eps


def tonp(a):
    return np.asarray(gnp.to_np(a) if backend == "numpy" else a.detach().cpu().numpy(), dtype=np.float64)


def ref_kernel(p, noisy):
    if not noisy:
        return lambda x, y, prm, pairwise=False: gp.kernel.maternp_covariance(x, y, p, prm, pairwise)

    def kernel(x, y, prm, pairwise=False):          # examples/gpmp_example07_nd_regression.py:95-131
        s2, nv, lir = gnp.exp(prm[0]), gnp.exp(prm[1]), prm[2:]
        if y is x or y is None:
            if pairwise:
                return s2 * gnp.ones((x.shape[0],))
            return s2 * gp.kernel.maternp_kernel(p, gnp.scaled_distance(lir, x, x)) + nv * gnp.eye(x.shape[0])
        D = gnp.scaled_distance_elementwise(lir, x, y) if pairwise else gnp.scaled_distance(lir, x, y)
        return s2 * gp.kernel.maternp_kernel(p, D)

    return kernel


def orc_kernel(p, noisy):
    f = orc.noisy_maternp_covariance if noisy else orc.maternp_covariance
    return lambda x, y, prm, pairwise=False: f(x, y, p, prm, pairwise)


def means(kind):
    if kind == "const":
        return (lambda x, prm: gnp.ones((x.shape[0], 1))), (lambda x, prm: np.ones((x.shape[0], 1)))
    if kind == "linear":
        return (lambda x, prm: gnp.hstack((gnp.ones((x.shape[0], 1)), gnp.asarray(x)))), (lambda x, prm: np.hstack((np.ones((x.shape[0], 1)), np.asarray(x))))
    if kind == "param":
        return (lambda x, prm: (prm[0] + prm[1] * x[:, 0]).reshape(-1, 1)), (lambda x, prm: (prm[0] + prm[1] * np.asarray(x)[:, 0]).reshape(-1, 1))
    return None, None


def main():
    rng = np.random.default_rng(seed)
    worst, bad = {}, []
    for i in range(ncases):
        d, p = int(rng.integers(1, 11)), int(rng.integers(0, 11))
        kind = str(rng.choice(["zero", "const", "linear", "param"]))
        q = {"zero": 0, "const": 1, "linear": d + 1, "param": 0}[kind]
        n, m = int(rng.integers(q + 2, 400)), int(rng.integers(1, 200))
        noisy = bool(rng.integers(0, 2))
        x, xt = rng.random((n, d)), rng.random((m, d))
        z = np.sin(3 * x[:, 0]) + x.sum(axis=1) + 0.05 * rng.standard_normal(n)
        th = np.concatenate(([0.3 * rng.standard_normal()], [math.log(10.0 ** rng.uniform(-4, -1))] if noisy else [], -np.log((0.2 + rng.random(d)) * math.sqrt(d))))
        mparam = np.array([0.3, -0.7]) if kind == "param" else None
        meantype = {"zero": "zero", "const": "linear_predictor", "linear": "linear_predictor", "param": "parameterized"}[kind]
        rmean, omean = means(kind)
        rk, ok = ref_kernel(p, noisy), orc_kernel(p, noisy)
        rmodel = gp.core.Model(rmean, rk, mparam, th, meantype)
        omodel = orc.OracleModel(omean, ok, mparam, th, meantype)
        K = ok(x, x, th)
        ev = np.linalg.eigvalsh(K)
        cond = float(ev[-1] / max(ev[0], 1e-300))
        if cond > 1e12:
            print(f"[oracle soak {i:3d}] skipped: cond(K) = {cond:.1e}", flush=True)
            continue
        cs = max(1.0, cond / 1e6)
        errs = {}
        zs, s2 = float(np.abs(z).max()), math.exp(th[0])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if backend == "numpy":
                errs["K_ii"] = float(np.max(np.abs(tonp(rk(x, x, th)) - K) / np.abs(K))) / 1e-14
                Kit = ok(x, xt, th)
                errs["K_it"] = float(np.max(np.abs(tonp(rk(x, xt, th)) - Kit) / np.maximum(np.abs(Kit), 1e-300))) / 1e-14
                kp = min(n, m)
                Kpw = ok(x[:kp], xt[:kp], th, True)
                errs["K_pairwise"] = float(np.max(np.abs(tonp(rk(x[:kp], xt[:kp], th, True)) - Kpw) / np.maximum(np.abs(Kpw), 1e-300))) / 1e-14
                rm, rv, rl = rmodel.predict(x, z, xt, return_lambdas=True, zero_neg_variances=False)
                om, ov, ol = orc.predict(omodel, x, z, xt, return_lambdas=True, zero_neg_variances=False)
                errs["mean"] = float(np.max(np.abs(tonp(rm) - om))) / (1e-10 * cs * zs)
                errs["var"] = float(np.max(np.abs(tonp(rv) - ov))) / (1e-10 * cs * s2)
                errs["lambda"] = float(np.max(np.abs(tonp(rl) - ol))) / (1e-7 * max(1.0, float(np.abs(ol).max())))
                if kind in ("zero", "param"):
                    rn = float(rmodel.negative_log_likelihood(mparam, th, x, z)) if kind == "param" else float(rmodel.negative_log_likelihood_zero_mean(th, x, z))
                    on = float(orc.negative_log_likelihood(omodel, mparam, th, x, z)) if kind == "param" else float(orc.negative_log_likelihood_zero_mean(omodel, th, x, z))
                else:
                    rn, on = float(rmodel.negative_log_restricted_likelihood(th, x, z)), float(orc.negative_log_restricted_likelihood(omodel, th, x, z))
                errs["criterion"] = abs(rn - on) / (1e-12 * cs * max(1.0, abs(on), n))
                rz, rs, re_ = (tonp(a) for a in rmodel.loo(x, z))
                oz, os_, oe = orc.loo(omodel, x, z)
                errs["loo"] = max(float(np.max(np.abs(rz - oz))) / max(zs, 1.0), float(np.max(np.abs(rs - os_) / os_)), float(np.max(np.abs(re_ - oe))) / max(float(np.abs(oe).max()), 1e-300)) / (1e-8 * cs)
            else:
                import torch

                if kind == "param":
                    continue
                crit = gp.kernel.negative_log_likelihood_zero_mean if kind == "zero" else gp.kernel.negative_log_restricted_likelihood
                _, pre, _, grad = gp.kernel.make_selection_criterion_with_gradient(rmodel, crit, x, z)
                tt = torch.as_tensor(th, dtype=torch.float64)
                rv_, rg = float(pre(tt)), tonp(grad(tt))
                ni = 1 if noisy else None
                if kind == "zero":
                    ov_, og = orc.nll_zero_mean_value_and_grad(x, z, p, th, ni)
                else:
                    ov_, og = orc.reml_value_and_grad(x, z, omean(x, None), p, th, ni)
                # the VALUE is compared across the reference's two backends (the oracle is the NumPy backend, bit for bit): the torch
                # backend's cdist expands norms, a distance error e ~ 1e-8 for near points that enters K as c e h for p >= 1 (1e-9
                # here) but as c e for the exponential kernel p = 0, which is only Lipschitz at h = 0 (1e-7)
                errs["value"] = abs(rv_ - ov_) / ((1e-7 if p == 0 else 1e-9) * cs * max(1.0, abs(ov_), n))
                errs["grad"] = float(np.max(np.abs(rg - og))) / (1e-7 * cs * max(1.0, float(np.abs(og).max())))
        over = {k: v for k, v in errs.items() if not v <= 1.0}
        for k, v in errs.items():
            worst[k] = max(worst.get(k, 0.0), v)
        if over:
            bad.append((i, n, m, d, p, kind, noisy, cond, over))
        print(f"[oracle soak {i:3d}] n={n} m={m} d={d} p={p} {kind}{' noisy' if noisy else ''} cond {cond:.1e}: "
              + " ".join(f"{k} {v:.2g}" for k, v in errs.items()) + (" FAILED" if over else ""), flush=True)
    print("worst deviation / tolerance per quantity:", {k: float(f"{v:.3g}") for k, v in worst.items()})
    print(f"ORACLE VS REFERENCE SOAK {'OK' if not bad else 'FAILED'} ({backend} backend, {ncases} draws, seed {seed})")
    if bad:
        print(bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
