"""DifferentiableSelectionCriterion for the hip backend.

Same method set as the reference wrappers -- NumPy backend gpmp/num/numpy_backend.py:329-350
(``gradient = None`` -> SciPy finite differences) and torch backend gpmp/num/torch_backend.py:547-604
(``evaluate_pre_grad(p)`` followed by ``gradient(p)`` at the SAME p).  Here the gradient is analytic:
when the wrapped criterion is one of the library's ML / REML criteria on a Matern covariance, the
value pass keeps its Cholesky state and ``gradient`` finishes it with potri + the fused trace
kernel (gpmp_amd/core/gradients.py).  Otherwise ``gradient`` is None exactly as in the NumPy backend.
"""
import math

import numpy


def _is_linalg_exception(exc):
    from . import _is_linalg_exception as f

    return f(exc)


class DifferentiableSelectionCriterion:
    def __init__(self, crit, x, z, analytic=None):
        self.crit = crit
        self.x, self.z = x, z
        self._analytic = analytic  # object with value_and_state(p) / gradient_from_state(state)
        self._p_value = None
        self._state = None
        self.gradient = None if analytic is None else self._gradient

    def __call__(self, p):
        return self.evaluate(p)

    def evaluate(self, p):
        return self.crit(p, self.x, self.z)

    def evaluate_no_grad(self, p):
        try:
            return self.crit(p, self.x, self.z)
        except Exception as exc:  # linear-algebra failure -> +inf (numpy_backend.py:344-350)
            if _is_linalg_exception(exc):
                return math.inf
            raise

    def evaluate_pre_grad(self, p):
        p_arr = numpy.array(numpy.asarray(p, dtype=numpy.float64), copy=True)
        if self._analytic is None:
            try:
                return float(self.crit(p_arr, self.x, self.z))
            except Exception as exc:
                if _is_linalg_exception(exc):
                    return math.inf
                raise
        self._p_value = p_arr
        try:
            value, self._state = self._analytic.value_and_state(p_arr, self.x, self.z)
            return float(value)
        except Exception as exc:
            if _is_linalg_exception(exc):
                self._state = None
                return math.inf
            raise

    def evaluate_many(self, P, want_grad=False):
        """Criterion at MANY parameter vectors (rows of ``P``) on the same data -- what a multi-chain sampler asks for at
        every step (the reference evaluates ``selection_criterion(p)`` chain after chain, gpmp/mcmc/param_posterior.py:229-278,
        mcmc/metropolis_hastings.py).  With an analytic ML / REML criterion and at most 4096 observations all rows go through
        ONE batched library call (gpmp_nll_grad_batch with per-problem parameters); otherwise they are evaluated one after the
        other.  A row whose factorisation fails gets +inf (and a zero gradient), as ``evaluate_no_grad`` does.
        Returns ``values`` (C,) or ``(values, grads)`` with ``grads`` (C, len(p))."""
        P = numpy.atleast_2d(numpy.asarray(P, dtype=numpy.float64))
        C = P.shape[0]
        values = numpy.full(C, math.inf)
        grads = numpy.zeros_like(P) if want_grad else None
        fast = getattr(self._analytic, "many_values_and_gradients", None)
        done = False
        if fast is not None:
            try:
                out = fast(P, self.x, self.z, want_grad)
            except Exception as exc:          # one bad row poisons a batched call: fall back to row-by-row below
                if not _is_linalg_exception(exc):
                    raise
                out = None
            if out is not None:
                values, g = out
                if want_grad:
                    grads = g
                done = True
        if not done:
            for c in range(C):
                if want_grad and self._analytic is not None:
                    values[c] = self.evaluate_pre_grad(P[c])
                    grads[c] = self._gradient(P[c])
                else:
                    values[c] = self.evaluate_no_grad(P[c])
        return (values, grads) if want_grad else values

    def _gradient(self, p):
        p_arr = numpy.asarray(p, dtype=numpy.float64)
        if self._p_value is None or not numpy.array_equal(p_arr, self._p_value):
            # torch backend raises here (torch_backend.py:588-594); recomputing is always correct
            self.evaluate_pre_grad(p_arr)
        if self._state is None:
            return numpy.zeros_like(p_arr)  # criterion was +inf at p
        return self._analytic.gradient_from_state(self._state)


class BatchDifferentiableSelectionCriterion:
    """Scalar criterion evaluated on mini-batches: gpmp/num/torch_backend.py:607-718, same constructor
    arguments, same weighting (each batch value times its size, ``reduction='mean'`` divides by the number of
    points seen) and the same cycling rule for ``batches_per_eval > 0``.

    ``loader`` is any sized iterable of ``(x_batch, z_batch)``.  The reference differentiates every batch with
    autograd inside ``evaluate_pre_grad``; here each batch contributes its analytic gradient
    (``analytic.value_and_state`` + ``gradient_from_state``), and without an analytic form ``gradient`` is None
    exactly as for the array criterion.
    """

    def __init__(self, crit, loader, reduction="mean", batches_per_eval=0, analytic=None):
        if reduction not in ("mean", "sum"):
            raise ValueError("reduction must be 'mean' or 'sum'")
        if batches_per_eval < 0:
            raise ValueError("batches_per_eval must be >= 0")
        if len(loader) == 0:
            raise ValueError("DataLoader is empty.")
        self.crit, self.loader, self.reduction, self.bpe = crit, loader, reduction, int(batches_per_eval)
        self._batch_iter = iter(loader) if self.bpe > 0 else None
        self._analytic = analytic
        self._gradient = None
        self.gradient = None if analytic is None else self._cached_gradient

    def __call__(self, p):
        return self.evaluate(p)

    def _batches(self):
        if self.bpe == 0:
            yield from self.loader
            return
        for _ in range(self.bpe):
            try:
                yield next(self._batch_iter)
            except StopIteration:
                self._batch_iter = iter(self.loader)
                yield next(self._batch_iter)

    @staticmethod
    def _prepare(xb, zb):
        from . import asarray

        return asarray(xb), asarray(zb)

    def _reduce(self, total, n):
        if n == 0:
            raise ValueError("Loader is empty.")
        return total / n if self.reduction == "mean" else total

    use_batched_kernel = True      # set False to force the one-batch-at-a-time route (tests compare both)

    def _accumulate(self, p_arr, want_grad, p_call=None):
        """Sum of (batch value x batch size), points seen and -- with ``want_grad`` -- the sum of (batch gradient x batch size)
        over the batches of this evaluation, STREAMING the loader: whether the batched kernel applies is decided first (a
        declared Matern covariance: model level, no data touched) and per batch from its size on the host (<= 4096 points);
        qualifying batches are collected into pieces of at most one library call's workspace budget
        (``batch_piece_limit``) and go through gpmp_nll_grad_batch piece by piece, every other batch is evaluated on its own
        and dropped -- at no time more than one piece of the loader is resident on the device.  ``p_call`` is what a
        non-analytic criterion callable receives (the caller's own object, as in the reference: torch_backend.py:664-676)."""
        an = self._analytic
        p_call = p_arr if p_call is None or an is not None else p_call
        fn = getattr(an, "batch_values_and_gradients", None)
        can_batch = fn is not None and self.use_batched_kernel and (getattr(an, "batch_qualifies", None) is None or an.batch_qualifies())
        max_pts = getattr(an, "batch_max_points", 0) if can_batch else 0
        limit_fn = getattr(an, "batch_piece_limit", None)
        total, n = 0.0, 0
        grad = numpy.zeros_like(p_arr) if want_grad else None
        pending, nmax = [], 0

        def one(xb, zb):
            nonlocal total, n, grad
            bs = xb.shape[0]
            if want_grad:
                value, state = an.value_and_state(p_arr, xb, zb)
                grad += bs * numpy.asarray(an.gradient_from_state(state), dtype=numpy.float64)
            else:
                value = self.crit(p_call, xb, zb)
            total += float(value) * bs
            n += bs

        def flush():
            nonlocal total, n, grad, pending, nmax, can_batch
            if not pending:
                return
            out = fn(p_arr, pending, want_grad)
            if out is None:                      # (e.g. more mean columns than the batched kernel carries): one by one,
                can_batch = False                # and for the REST of this evaluation -- the design does not change between pieces
                for xb, zb in pending:
                    one(xb, zb)
            else:
                sizes = numpy.array([int(xb.shape[0]) for xb, _ in pending], dtype=numpy.float64)
                total += float(numpy.dot(out[0], sizes))
                n += int(sizes.sum())
                if want_grad:
                    grad += sizes @ out[1]
            pending, nmax = [], 0

        for xb_h, zb_h in self._batches():
            bs = int(xb_h.shape[0])
            if can_batch and 0 < bs <= max_pts:
                xb, zb = self._prepare(xb_h, zb_h)
                pending.append((xb, zb))
                nmax = max(nmax, bs)
                # q is not known here (the mean is a user callable): size the piece for the widest design the kernel takes
                lim = limit_fn(nmax, int(xb.shape[1]), 7, want_grad, xb.device) if limit_fn is not None else 64
                if len(pending) >= max(1, lim):
                    flush()
            else:
                one(*self._prepare(xb_h, zb_h))
        flush()
        return total, n, grad

    def evaluate(self, p):
        p_arr = numpy.array(numpy.asarray(p, dtype=numpy.float64), copy=True)
        total, n, _ = self._accumulate(p_arr, False, p_call=p)
        return self._reduce(total, n)

    def evaluate_no_grad(self, p):
        try:
            return self.evaluate(p)
        except Exception as exc:
            if _is_linalg_exception(exc):
                return math.inf
            raise

    def evaluate_pre_grad(self, p):
        p_arr = numpy.array(numpy.asarray(p, dtype=numpy.float64), copy=True)
        if self._analytic is None:
            return self.evaluate_no_grad(p)
        try:
            total, n, grad = self._accumulate(p_arr, True)
        except Exception as exc:
            if _is_linalg_exception(exc):
                self._gradient = numpy.zeros_like(p_arr)
                return math.inf
            raise
        if n == 0:
            raise ValueError("Loader is empty.")
        if self.reduction == "mean":
            total, grad = total / n, grad / n
        self._gradient = grad
        return total

    def _cached_gradient(self, _p):
        if self._gradient is None:
            raise RuntimeError("Call `evaluate_pre_grad` first.")
        return self._gradient


class SecondOrderDifferentiableFunction:
    """Gradient and Hessian of a scalar function of a small parameter vector -- same method set as
    gpmp/num/torch_backend.py:721-779 (``evaluate(x)``, ``gradient()``, ``hessian()``), which uses second-order
    autograd.  The hip backend records no autograd graph through its kernels, so the derivatives are central
    finite differences of ``f`` (5-point first derivative as gpmp/num/shared.py:44-55, 4-point mixed second
    derivatives), i.e. O(p^2) evaluations of ``f`` for p parameters; intended for the p ~ d + 1 covariance
    parameters of gpmp/core/fisher.py:158-191.
    """

    def __init__(self, f, step=1e-3):
        self.f, self.h = f, float(step)
        self._x = self._y = None

    def _val(self, x):
        v = self.f(numpy.array(x, dtype=numpy.float64))
        v = v.item() if hasattr(v, "item") else float(v)
        return float(v)

    def evaluate(self, x):
        self._x = numpy.array(numpy.asarray(x.detach().cpu() if hasattr(x, "detach") else x, dtype=numpy.float64), copy=True).reshape(-1)
        self._y = self._val(self._x)
        return self._y

    def gradient(self, retain=True):
        if self._x is None:
            raise RuntimeError("Call evaluate(x) before calling gradient().")
        x, h = self._x, self.h
        g = numpy.zeros_like(x)
        for i in range(x.size):
            e = numpy.zeros_like(x)
            e[i] = h
            g[i] = (-self._val(x + 2 * e) + 8 * self._val(x + e) - 8 * self._val(x - e) + self._val(x - 2 * e)) / (12 * h)
        self._grad = g
        return g

    def hessian(self):
        if self._x is None:
            raise RuntimeError("Call evaluate(x) before calling hessian().")
        x, h, p = self._x, self.h, self._x.size
        H = numpy.zeros((p, p))
        f0 = self._y
        for i in range(p):
            ei = numpy.zeros_like(x)
            ei[i] = h
            H[i, i] = (-self._val(x + 2 * ei) + 16 * self._val(x + ei) - 30 * f0 + 16 * self._val(x - ei) - self._val(x - 2 * ei)) / (12 * h * h)
            for j in range(i):
                ej = numpy.zeros_like(x)
                ej[j] = h
                H[i, j] = H[j, i] = (self._val(x + ei + ej) - self._val(x + ei - ej) - self._val(x - ei + ej) + self._val(x - ei - ej)) / (4 * h * h)
        return H
