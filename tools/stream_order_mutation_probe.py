#!/usr/bin/env python3
"""Can the stream-ordered in-process fabric (tools/thread_ranks.py) SEE a missing stream dependency of the distributed schedule?
Mutation test: the 2 x 4 grid of thread-ranks runs factorisation + NLL + prediction with weights three times --

  intact          the schedule as shipped, messages held back by up to ~1 ms on the receiving streams
  mutated_main    ``_Streams.wait`` ignores every wait of the CALLER'S (bulk) stream: the trailing updates and bulk solve updates no
                  longer wait for the panel chain's messages
  mutated_side    ... ignores every wait of the SIDE stream instead: the panel chain no longer waits for the bulk updates

and compares each with the single-GPU result.  Expected: intact agrees to rounding, both mutations do not (a fabric that
synchronised the host around every message, as gloo does, would hide part of this).  Prints one line per case.

    python tools/stream_order_mutation_probe.py
"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    import gpmp_amd as gp
    import gpmp_amd.num as gnp
    from gpmp_amd.dist import HipLocalOps, ProcessGrid
    from gpmp_amd.dist import cholesky as chmod
    from gpmp_amd.kernel import MaternCovariance
    from tests.helpers import make_xz, theta_aniso
    from tools import thread_ranks

    pr, pc, n, m, d, nb = 2, 4, 8192, 1500, 4, 512
    x, z = make_xz(n, d, 11)
    xt, _ = make_xz(m, d, 12)
    th = theta_aniso(d, scale=0.5)
    cov = MaternCovariance(2)
    ref_model = gp.Model(None, cov, None, th, "zero")
    rm, rv = ref_model.predict(x, z, xt)
    rnll = float(ref_model.negative_log_likelihood_zero_mean(th, x, z))
    orig_wait = chmod._Streams.wait

    def run_case(mutation, host_staged=False):
        out = {}

        def wait(self, side, ev):
            if mutation == "mutated_main" and not side:
                return
            if mutation == "mutated_side" and side:
                return
            return orig_wait(self, side, ev)

        chmod._Streams.wait = wait
        try:
            def body(rank, world, fabric, classes):
                Ch = classes[0] if host_staged else classes[1]
                Ch.max_delay_cycles = 3_000_000
                grid = ProcessGrid(pr, pc)
                ch = Ch(grid, n, nb=nb, ops=HipLocalOps())
                ch.build_local_gram(cov, x, th, 10.0 * math.exp(th[0]) * gnp.eps)
                info = ch.factor()
                nll = ch.negative_log_likelihood(z) if info == 0 else math.nan
                mean = var = None
                j0 = j1 = 0
                if info == 0:
                    mean, var, (j0, j1) = ch.predict_zero_mean(cov, x, z, xt, th)
                torch.cuda.synchronize()
                out[rank] = (grid.r, j0, j1, mean, var, info, nll)

            errors = thread_ranks.run(pr * pc, body, limit_s=240.0)
        finally:
            chmod._Streams.wait = orig_wait
        if errors:
            return {"case": mutation, "error": errors[0].strip().splitlines()[-1][:200]}
        zpm = np.full(m, np.nan)
        for (r, a, b, mu, v, info, nll) in out.values():
            if mu is not None and r == 0:
                zpm[a:b] = mu
        o = out[0]
        return {"case": mutation, "info": int(o[5]), "nll_rel_diff": abs(o[6] - rnll) / abs(rnll) if o[5] == 0 else None,
                "max_abs_dmean": float(np.nanmax(np.abs(zpm - rm))) if np.isfinite(zpm).any() else None}

    def agrees(res):
        return bool(res.get("info") == 0 and res.get("nll_rel_diff") is not None and res["nll_rel_diff"] < 1e-9 and res["max_abs_dmean"] < 1e-7)

    ok = True
    for case in ("intact", "mutated_main", "mutated_side", "intact"):
        res = run_case(case)
        res["fabric"] = "stream-ordered, device-resident"
        res["agrees_with_single_gpu"] = agrees(res)
        print(res, flush=True)
        ok = ok and (res["agrees_with_single_gpu"] == (case == "intact"))
    # the same mutations on the host-staged fabric (what a gloo run is): informational -- does a host synchronisation around
    # every message hide them?
    for case in ("mutated_main", "mutated_side"):
        res = run_case(case, host_staged=True)
        res["fabric"] = "host-staged (as over gloo)"
        res["agrees_with_single_gpu"] = agrees(res)
        print(res, flush=True)
    print("MUTATION PROBE", "OK: the stream-ordered fabric sees both missing dependencies and passes the intact schedule" if ok else "INCONCLUSIVE")


if __name__ == "__main__":
    main()
