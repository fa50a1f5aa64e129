#!/usr/bin/env python3
"""Random soak of the PRODUCT's host-side REMAP pieces (gpmp_amd/kernel/priors.py, prior_helpers.py: pure host arithmetic, no GPU)
against the LIVE REFERENCE (gpmp/kernel/priors.py:43-312, prior_helpers.py:22-95, 220-290).  Build container only.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg PYTHONPATH=/root/reference:/root/repo GPMP_BACKEND=numpy \
        python3 /root/repo/tests/golden/priors_vs_reference_soak.py 500 5
"""
import os
import sys

import numpy as np

os.environ["GPMP_BACKEND"] = "numpy"
os.environ.setdefault("GPMP_LOG_LEVEL", "WARNING")
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5

import gpmp.kernel.prior_helpers as rh  # noqa: E402  (the reference)
import gpmp.kernel.priors as rp  # noqa: E402
import gpmp.num as rgnp  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpmp_amd.kernel import prior_helpers as ph  # noqa: E402
from gpmp_amd.kernel import priors as pp  # noqa: E402


def f(v):
    return np.asarray(rgnp.to_np(v), dtype=np.float64)


def same(a, b, rtol=1e-13):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        return False
    fin = np.isfinite(a) & np.isfinite(b)
    return bool(np.array_equal(np.isfinite(a), np.isfinite(b)) and np.array_equal(a[~fin], b[~fin]) and np.allclose(a[fin], b[fin], rtol=rtol, atol=1e-300))


def agree(ref_call, prod_call):
    """same value -- or the same exception type (argument errors are part of the behaviour)"""
    out = []
    for c in (ref_call, prod_call):
        try:
            out.append(("ok", f(c())))
        except Exception as e:  # noqa: BLE001
            out.append(("exc", type(e).__name__))
    (ka, va), (kb, vb) = out
    return ka == kb and (va == vb if ka == "exc" else same(va, vb))


def main():
    rng = np.random.default_rng(seed)
    bad = []
    for i in range(ncases):
        n, d = int(rng.integers(1, 200)), int(rng.integers(1, 12))
        xi = rng.random((n, d)) * 10.0 ** rng.uniform(-2, 2)
        if rng.random() < 0.3:                       # a column on a grid (repeated values: zero gaps are skipped)
            xi[:, int(rng.integers(d))] = rng.integers(0, 4, n) * 0.25
        if rng.random() < 0.15:                      # a constant column: no finite gap
            xi[:, int(rng.integers(d))] = 0.7
        theta = np.concatenate(([rng.normal(0, 2)], rng.normal(0, 2, d)))
        errs = {}
        kw = {} if rng.random() < 0.5 else {"prior_rho_min_range_factor": float(10.0 ** rng.uniform(-3, -0.5))}
        a, b = f(rh.compute_logrho_min_from_xi(xi, **kw)), ph.compute_logrho_min_from_xi(xi, **kw)
        errs["logrho_min"] = same(a, b)
        ra = rh.resolve_logsigma2_logrho_prior_args(covparam0_prior=theta, xi=xi)
        pa = ph.resolve_logsigma2_logrho_prior_args(covparam0_prior=theta, xi=xi)
        errs["resolve_args"] = len(ra) == len(pa) and all(same(f(u), np.asarray(v, dtype=np.float64) if not np.isscalar(v) else v) for u, v in zip(ra, pa))
        ls20, lr0, lrmin = float(f(ra[4])), f(ra[5]), f(ra[6])
        t2 = theta + rng.normal(0, 1.0, d + 1)
        errs["jeffreys"] = same(f(rp.log_prior_jeffreys_variance(t2, 1.3)), pp.log_prior_jeffreys_variance(t2, 1.3))
        errs["power_law"] = same(f(rp.log_prior_power_law(t2)), pp.log_prior_power_law(t2))
        errs["gaussian_logsigma2"] = same(f(rp.log_prior_gaussian_logsigma2(t2, ls20)), pp.log_prior_gaussian_logsigma2(t2, ls20))
        for g, cov in ((1.2, 0.9), (2.0, 0.99)):
            errs[f"gaussian_logsigma2[{g},{cov}]"] = same(f(rp.log_prior_gaussian_logsigma2(t2, ls20, gamma=g, sigma2_coverage=cov)),
                                                          pp.log_prior_gaussian_logsigma2(t2, ls20, gamma=g, sigma2_coverage=cov))
        if np.all(np.isfinite(lrmin)):
            # below, at and above the lower bound of a length scale
            for shift in (-0.5, 0.0, 0.3, 3.0):
                t3 = t2.copy()
                t3[1:] = -(lrmin + shift + 0.1 * rng.random(d))
                # (logrho_0 <= logrho_min somewhere -- the prior's centre below the data-driven bound -- is a ValueError in both)
                errs[f"logrho_barrier[{shift}]"] = agree(lambda: rp.log_prior_logrho_barrier_linear(t3, lrmin, lr0), lambda: pp.log_prior_logrho_barrier_linear(t3, lrmin, lr0))
                errs[f"neglog_f[{shift}]"] = agree(lambda: rp.neglog_f_logrho(-t3[1:], lrmin, lr0, alpha=0.7), lambda: pp.neglog_f_logrho(-t3[1:], lrmin, lr0, alpha=0.7))
        wrong = [k for k, v in errs.items() if not v]
        if wrong:
            bad.append((i, n, d, wrong))
        print(f"[priors soak {i:3d}] n={n} d={d}: {len(errs)} comparisons" + (f" FAILED {wrong}" if wrong else ""), flush=True)
    print(f"PRIORS VS REFERENCE SOAK {'OK' if not bad else 'FAILED'} ({ncases} draws, seed {seed})")
    if bad:
        print(bad[:10])
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
