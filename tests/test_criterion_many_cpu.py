"""Host logic of ``criterion.evaluate_many`` (gpmp_amd/num/criterion.py): which route is taken and what a failing row does.
No GPU: the criterion and the analytic object are stand-ins (the batched device route is tested in
tests/test_batch_driver_gpu.py::test_criterion_at_many_parameter_vectors_sampler_pattern)."""
import math

import numpy as np

from gpmp_amd.num.criterion import DifferentiableSelectionCriterion


def _crit(p, x, z):
    p = np.asarray(p, dtype=float)
    if p[0] < 0:
        raise np.linalg.LinAlgError("Matrix is not positive definite")
    return float(np.sum((p - 1.0) ** 2) + x + z)


class _Analytic:
    def __init__(self, batched):
        self.batched, self.calls = batched, 0

    def value_and_state(self, p, x, z):
        if p[0] < 0:
            raise np.linalg.LinAlgError("Matrix is not positive definite")
        return float(np.sum((p - 1.0) ** 2) + x + z), np.array(p, dtype=float)

    def gradient_from_state(self, st):
        return 2.0 * (st - 1.0)

    def many_values_and_gradients(self, P, x, z, want_grad):
        self.calls += 1
        if self.batched == "none":
            return None                              # the batched driver does not apply (e.g. n > 2048)
        if np.any(P[:, 0] < 0):
            raise np.linalg.LinAlgError("Matrix is not positive definite: batched problem failed")
        v = np.sum((P - 1.0) ** 2, axis=1) + x + z
        return v, (2.0 * (P - 1.0) if want_grad else None)


P = np.array([[1.0, 1.0], [0.0, 2.0], [3.0, -1.0]])


def test_without_analytic_form_rows_are_evaluated_one_by_one():
    c = DifferentiableSelectionCriterion(_crit, 2.0, 3.0)
    np.testing.assert_allclose(c.evaluate_many(P), [5.0, 7.0, 13.0])
    bad = P.copy(); bad[1, 0] = -1.0
    v = c.evaluate_many(bad)
    assert v[0] == 5.0 and math.isinf(v[1]) and v[2] == 13.0


def test_batched_route_is_used_when_it_applies():
    a = _Analytic("yes")
    c = DifferentiableSelectionCriterion(_crit, 2.0, 3.0, analytic=a)
    v, g = c.evaluate_many(P, want_grad=True)
    assert a.calls == 1
    np.testing.assert_allclose(v, [5.0, 7.0, 13.0])
    np.testing.assert_allclose(g, 2.0 * (P - 1.0))
    np.testing.assert_allclose(c.evaluate_many(P[0]), [5.0])          # a single vector is one row


def test_fallback_when_the_batched_route_declines_or_a_row_fails():
    a = _Analytic("none")
    c = DifferentiableSelectionCriterion(_crit, 2.0, 3.0, analytic=a)
    v, g = c.evaluate_many(P, want_grad=True)
    assert a.calls == 1
    np.testing.assert_allclose(v, [5.0, 7.0, 13.0]); np.testing.assert_allclose(g, 2.0 * (P - 1.0))
    a2 = _Analytic("yes")
    c2 = DifferentiableSelectionCriterion(_crit, 2.0, 3.0, analytic=a2)
    bad = P.copy(); bad[2, 0] = -3.0
    v, g = c2.evaluate_many(bad, want_grad=True)                      # the batched call raises: row by row, +inf for the bad row
    assert math.isinf(v[2]) and np.all(g[2] == 0.0)
    np.testing.assert_allclose(v[:2], [5.0, 7.0]); np.testing.assert_allclose(g[:2], 2.0 * (P[:2] - 1.0))
