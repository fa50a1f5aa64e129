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

    def _gradient(self, p):
        p_arr = numpy.asarray(p, dtype=numpy.float64)
        if self._p_value is None or not numpy.array_equal(p_arr, self._p_value):
            # torch backend raises here (torch_backend.py:588-594); recomputing is always correct
            self.evaluate_pre_grad(p_arr)
        if self._state is None:
            return numpy.zeros_like(p_arr)  # criterion was +inf at p
        return self._analytic.gradient_from_state(self._state)
