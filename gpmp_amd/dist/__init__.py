"""gpmp_amd.dist -- one process per GPU over torch.distributed (backend "nccl" = RCCL over xGMI).

The reference has no distributed code.  Two places of the hot path shard:

* ``sharded_predict``: the prediction set is split over ranks, the observations are replicated and
  every rank factors K itself -- no collective on the data path (section 8e.2).
* ``BlockCyclicCholesky``: for n beyond one GPU's HBM, K is 2-D block-cyclic over a Pr x Pc process
  grid; per block column the diagonal factor and the panel are broadcast with RCCL and each rank updates
  its own blocks with the fp64 MFMA GEMM (section 8e.3).  Scalars (log-det, quadratic form) are reduced
  with one small all-reduce.  On that factor: NLL / REML and their analytic gradient, leave-one-out, zero-mean and universal
  kriging, and ``fit_covparam`` (SciPy over the distributed criterion).

The host logic is backend independent (``LocalOps``): the product path uses ``HipLocalOps`` (C ABI of
libgpmp_hip.so); the CPU tests inject a NumPy implementation and run the same schedule over gloo.
"""
from .grid import ProcessGrid
from .cholesky import BlockCyclicCholesky
from .local_ops import HipLocalOps
from .predict import sharded_predict, shard_bounds
from .fit import distributed_criterion, fit_covparam
from .model import DistributedModel

__all__ = ["ProcessGrid", "BlockCyclicCholesky", "HipLocalOps", "sharded_predict", "shard_bounds", "distributed_criterion", "fit_covparam", "DistributedModel"]
