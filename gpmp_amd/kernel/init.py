"""Initial guess of the covariance parameters -- counterpart of gpmp/kernel/init.py (arrays path)."""
import math

import numpy

from .. import num as gnp


def _rho_from_range(xi):
    d = xi.shape[1]
    delta = gnp.to_np(gnp.max(xi, axis=0) - gnp.min(xi, axis=0))
    return math.exp(math.lgamma(d / 2 + 1) / d) / (math.pi ** 0.5) * delta


def anisotropic_parameters_initial_guess_zero_mean(model, xi=None, zi=None, dataloader=None):
    """gpmp/kernel/init.py:27-36."""
    if dataloader is not None:
        raise NotImplementedError("dataloaders are outside the hot path (SURVEY.md section 2, row 23)")
    xi, zi = gnp.asarray(xi), gnp.asarray(zi).reshape(-1)
    rho = _rho_from_range(xi)
    covparam = numpy.concatenate(([math.log(1.0)], -numpy.log(rho)))
    sigma2 = float(model.norm_k_sqrd_with_zero_mean(xi, zi, covparam)) / xi.shape[0]
    return numpy.concatenate(([math.log(sigma2)], -numpy.log(rho)))


def anisotropic_parameters_initial_guess(model, xi=None, zi=None, dataloader=None):
    """gpmp/kernel/init.py:54-66: rho from the data range, sigma^2 = (Wz)^T (WKW)^-1 (Wz) / n."""
    if dataloader is not None:
        raise NotImplementedError("dataloaders are outside the hot path (SURVEY.md section 2, row 23)")
    xi, zi = gnp.asarray(xi), gnp.asarray(zi).reshape(-1)
    rho = _rho_from_range(xi)
    covparam = numpy.concatenate(([math.log(1.0)], -numpy.log(rho)))
    sigma2 = float(model.norm_k_sqrd(xi, zi, covparam)) / xi.shape[0]
    return numpy.concatenate(([math.log(sigma2)], -numpy.log(rho)))


def anisotropic_parameters_initial_guess_constant_mean(model, xi=None, zi=None, dataloader=None):
    """gpmp/kernel/init.py:38-52: (GLS constant mean, covparam) from K^-1 1 and K^-1 z at unit variance."""
    if dataloader is not None:
        raise NotImplementedError("dataloaders are outside the hot path (SURVEY.md section 2, row 23)")
    xi, zi = gnp.asarray(xi), gnp.asarray(zi).reshape(-1, 1)
    n = xi.shape[0]
    rho = _rho_from_range(xi)
    covparam = numpy.concatenate(([math.log(1.0)], -numpy.log(rho)))
    zTKinvz, Kinv1, Kinvz = model.k_inverses(xi, zi, covparam)
    mean_gls = float(gnp.sum(Kinvz)) / float(gnp.sum(Kinv1))
    sigma2 = float(zTKinvz) / n
    return numpy.array([mean_gls]), numpy.concatenate(([math.log(sigma2)], -numpy.log(rho)))
