"""Initial guess of the covariance parameters -- counterpart of gpmp/kernel/init.py (arrays or a DataLoader)."""
import math

import numpy

from .. import num as gnp


def _rho_from_range(xi, dataloader=None):
    """init.py:15-17,27-30: per-dimension range of the inputs (from the shards' extrema for a loader)."""
    if dataloader is not None:
        ds = dataloader.dataset
        delta = gnp.to_np(ds._reduce_max("x") - ds._reduce_min("x"))
    else:
        delta = gnp.to_np(gnp.max(xi, axis=0) - gnp.min(xi, axis=0))
    d = delta.shape[0]
    return math.exp(math.lgamma(d / 2 + 1) / d) / (math.pi ** 0.5) * delta


def _source(xi, zi, dataloader):
    arrays = xi is not None and zi is not None
    if arrays and dataloader is not None:
        raise ValueError("Provide either (xi, zi) or loader, not both.")
    if not arrays and dataloader is None:
        raise ValueError("Provide either (xi, zi) or loader.")
    return arrays


def anisotropic_parameters_initial_guess_zero_mean(model, xi=None, zi=None, dataloader=None):
    """gpmp/kernel/init.py:27-36."""
    if _source(xi, zi, dataloader):
        xi, zi = gnp.asarray(xi), gnp.asarray(zi).reshape(-1)
        rho = _rho_from_range(xi)
        covparam = numpy.concatenate(([math.log(1.0)], -numpy.log(rho)))
        sigma2 = float(model.norm_k_sqrd_with_zero_mean(xi, zi, covparam)) / xi.shape[0]
    else:   # batch-size weighted mean of the per-batch estimates (init.py:31-32)
        rho = _rho_from_range(None, dataloader)
        covparam = numpy.concatenate(([math.log(1.0)], -numpy.log(rho)))
        sigma2 = float(dataloader.reduce_mean(
            lambda x, z: float(model.norm_k_sqrd_with_zero_mean(x, gnp.asarray(z).reshape(-1), covparam)) / x.shape[0]))
    return numpy.concatenate(([math.log(sigma2)], -numpy.log(rho)))


def anisotropic_parameters_initial_guess(model, xi=None, zi=None, dataloader=None):
    """gpmp/kernel/init.py:54-66: rho from the data range, sigma^2 = (Wz)^T (WKW)^-1 (Wz) / n."""
    if _source(xi, zi, dataloader):
        xi, zi = gnp.asarray(xi), gnp.asarray(zi).reshape(-1)
        rho = _rho_from_range(xi)
        covparam = numpy.concatenate(([math.log(1.0)], -numpy.log(rho)))
        sigma2 = float(model.norm_k_sqrd(xi, zi, covparam)) / xi.shape[0]
    else:   # init.py:63-65
        rho = _rho_from_range(None, dataloader)
        covparam = numpy.concatenate(([math.log(1.0)], -numpy.log(rho)))
        sigma2 = float(dataloader.reduce_mean(
            lambda x, z: float(model.norm_k_sqrd(x, gnp.asarray(z).reshape(-1), covparam)) / x.shape[0]))
    return numpy.concatenate(([math.log(sigma2)], -numpy.log(rho)))


def anisotropic_parameters_initial_guess_constant_mean(model, xi=None, zi=None, dataloader=None):
    """gpmp/kernel/init.py:38-52: (GLS constant mean, covparam) from K^-1 1 and K^-1 z at unit variance."""
    def gls(x, z, covparam):
        zTKinvz, Kinv1, Kinvz = model.k_inverses(gnp.asarray(x), gnp.asarray(z).reshape(-1, 1), covparam)
        return float(gnp.sum(Kinvz)) / float(gnp.sum(Kinv1)), float(zTKinvz) / x.shape[0]

    if _source(xi, zi, dataloader):
        rho = _rho_from_range(gnp.asarray(xi))
        covparam = numpy.concatenate(([math.log(1.0)], -numpy.log(rho)))
        mean_gls, sigma2 = gls(xi, zi, covparam)
    else:   # init.py:46-51
        rho = _rho_from_range(None, dataloader)
        covparam = numpy.concatenate(([math.log(1.0)], -numpy.log(rho)))
        both = gnp.to_np(dataloader.reduce_mean(lambda x, z: numpy.array(gls(x, z, covparam))))
        mean_gls, sigma2 = float(both[0]), float(both[1])
    return numpy.array([mean_gls]), numpy.concatenate(([math.log(sigma2)], -numpy.log(rho)))
