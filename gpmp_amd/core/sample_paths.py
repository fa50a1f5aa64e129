"""Sample paths -- counterpart of gpmp/core/sample_paths.py (Cholesky / SVD routes + conditioning by kriging)."""
from .. import num as gnp
from .utils import mean_values as _mean_values


def sample_paths(model, xt, nb_paths, method: str = "chol", check_result: bool = True):
    """gpmp/core/sample_paths.py:18-60: C W with K(xt, xt) = C C^T and W ~ N(0, I) (backend RNG)."""
    xt_ = gnp.asarray(xt)
    K = gnp.asarray(model.covariance(xt_, xt_, model.covparam))
    if method == "chol":
        C = gnp.cholesky(K)        # raises LinAlgError (not NaNs) when K is not positive definite
    elif method == "svd":
        # symmetric square root U sqrt(s) U^T for covariances that are only positive SEMI-definite (repeated points:
        # observation and prediction sets stacked).  Dense symmetric decomposition from torch (not on the hot path).
        U, s, Vt = gnp.svd(K, full_matrices=True, hermitian=True)
        C = gnp.matmul(U * gnp.sqrt(s), Vt)
    else:
        raise ValueError("method must be 'chol' or 'svd'")
    return gnp.matmul(C, gnp.randn(K.shape[0], nb_paths))


def conditional_sample_paths(model, ztsim, xi_ind, zi, xt_ind, lambda_t, convert_out: bool = True):
    """gpmp/core/sample_paths.py:63-119: ztsim[xt_ind] + lambda_t^T (zi - ztsim[xi_ind])."""
    import torch

    zi_ = gnp.asarray(zi).reshape(-1, 1)
    ztsim_ = gnp.asarray(ztsim)
    xi_ind = torch.as_tensor(gnp.to_np(xi_ind), device=ztsim_.device).long().reshape(-1)
    xt_ind = torch.as_tensor(gnp.to_np(xt_ind), device=ztsim_.device).long().reshape(-1)
    delta = zi_ - ztsim_[xi_ind, :]
    lam = gnp.asarray(lambda_t)
    out = ztsim_[xt_ind, :] + gnp.matmul(lam.T.contiguous(), delta)
    return gnp.to_np(out) if convert_out else out


def conditional_sample_paths_parameterized_mean(model, ztsim, xi, xi_ind, zi, xt, xt_ind, lambda_t, convert_out: bool = True):
    """gpmp/core/sample_paths.py:122-182."""
    import torch

    xi_, zi_, xt_ = gnp.asarray(xi), gnp.asarray(zi).reshape(-1), gnp.asarray(xt)
    ztsim_ = gnp.asarray(ztsim)
    xi_ind = torch.as_tensor(gnp.to_np(xi_ind), device=ztsim_.device).long().reshape(-1)
    xt_ind = torch.as_tensor(gnp.to_np(xt_ind), device=ztsim_.device).long().reshape(-1)
    zi_centered = zi_ - _mean_values(model, xi_, model.meanparam).reshape(-1)
    zt_prior_mean = _mean_values(model, xt_, model.meanparam).reshape(-1, 1)
    delta = zi_centered.reshape(-1, 1) - ztsim_[xi_ind, :]
    out = ztsim_[xt_ind, :] + gnp.matmul(gnp.asarray(lambda_t).T.contiguous(), delta) + zt_prior_mean
    return gnp.to_np(out) if convert_out else out
