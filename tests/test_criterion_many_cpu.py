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


# ---- BatchDifferentiableSelectionCriterion: the loader is STREAMED (round 2 materialised every batch on the device before
#      it knew whether the batched kernel applied)
class _Loader:
    """sized iterable of (x, z) host batches that counts how many batches have been drawn"""

    def __init__(self, sizes):
        self.sizes, self.drawn = sizes, 0

    def __len__(self):
        return len(self.sizes)

    def __iter__(self):
        for k, n in enumerate(self.sizes):
            self.drawn += 1
            yield np.full((n, 2), float(k)), np.full(n, 1.0)


class _BatchAnalytic:
    batch_max_points = 100

    def __init__(self, loader, qualifies, limit):
        self.loader, self.qualifies, self.limit = loader, qualifies, limit
        self.single_at, self.pieces = [], []

    def batch_qualifies(self):
        return self.qualifies

    def batch_piece_limit(self, nmax, d, q, want_grad, device=None):
        return self.limit

    def value_and_state(self, p, x, z):
        self.single_at.append((self.loader.drawn, int(x.shape[0])))
        return float(np.sum(p) + x[0, 0]), np.asarray(p, dtype=float)

    def gradient_from_state(self, st):
        return np.ones_like(st)

    def batch_values_and_gradients(self, p, batches, want_grad):
        self.pieces.append((self.loader.drawn, [int(x.shape[0]) for x, _ in batches]))
        v = np.array([float(np.sum(p) + float(x[0, 0])) for x, _ in batches])
        return v, (np.ones((len(batches), len(p))) if want_grad else None)


def _make(sizes, qualifies, limit, monkeypatch):
    from gpmp_amd.num.criterion import BatchDifferentiableSelectionCriterion

    loader = _Loader(sizes)
    an = _BatchAnalytic(loader, qualifies, limit)
    c = BatchDifferentiableSelectionCriterion(lambda p, x, z: float(np.sum(p) + x[0, 0]), loader, reduction="mean", analytic=an)
    monkeypatch.setattr(c, "_prepare", lambda xb, zb: (xb, zb))          # (no device here)
    return c, an, loader


def test_batch_criterion_streams_when_the_batched_kernel_does_not_apply(monkeypatch):
    sizes = [10, 20, 30, 40]
    c, an, loader = _make(sizes, qualifies=False, limit=8, monkeypatch=monkeypatch)
    p = np.array([0.5, 0.25])
    v = c.evaluate_pre_grad(p)
    # every batch is evaluated as soon as it is drawn: batch k is handled when exactly k + 1 batches have left the loader
    assert an.single_at == [(k + 1, n) for k, n in enumerate(sizes)] and an.pieces == []
    want = sum((0.75 + k) * n for k, n in enumerate(sizes)) / sum(sizes)
    assert abs(v - want) < 1e-14
    np.testing.assert_allclose(c._cached_gradient(p), [1.0, 1.0])


def test_batch_criterion_collects_bounded_pieces_and_isolates_oversized_batches(monkeypatch):
    sizes = [10, 20, 500, 30, 40, 50, 60]                  # 500 > batch_max_points: never enters a piece
    c, an, loader = _make(sizes, qualifies=True, limit=2, monkeypatch=monkeypatch)
    p = np.array([0.5, 0.25])
    v = c.evaluate_pre_grad(p)
    assert [sz for _, sz in an.pieces] == [[10, 20], [30, 40], [50, 60]]      # pieces of at most `limit` problems
    assert an.pieces[0][0] == 2 and an.pieces[1][0] == 5                         # ... flushed as soon as they are full
    assert an.single_at == [(3, 500)]                                           # the oversized batch alone, when drawn
    want = sum((0.75 + k) * n for k, n in enumerate(sizes)) / sum(sizes)
    assert abs(v - want) < 1e-14 and abs(c.evaluate(p) - want) < 1e-14
