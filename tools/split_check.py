import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import torch, gpmp_amd.num as gnp
from oracle import gp_oracle as orc
import scipy.linalg as sla
for n in (2048, 2100, 3333):
    rng = np.random.default_rng(n)
    x = rng.random((n, 3)); th = np.array([0.2, 0.5, 0.3, 0.1])
    K = orc.maternp_covariance(x, None, 2, th) + 1e-6 * np.eye(n)
    L = gnp.to_np(gnp.cholesky(gnp.asarray(K)))
    Lr = sla.cholesky(K, lower=True)
    print(n, "max rel err", np.abs(L - Lr).max() / np.abs(Lr).max())
