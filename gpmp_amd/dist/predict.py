"""xt-sharded prediction: each rank predicts its own slice of the target points (no data-path collective)."""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(m: int, world: int, rank: int):
    """Contiguous, balanced split of m items: the first (m % world) ranks get one extra."""
    base, extra = divmod(m, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def sharded_predict(model, xi, zi, xt, gather: bool = True, group=None):
    """Posterior mean / variance at xt with the target points split over the ranks of ``group``.

    Every rank holds (xi, zi) and the full xt (or just needs its own slice to be valid); it builds and
    factors K itself and solves for its slice.  With ``gather`` the (m,) results are assembled on every
    rank with one all-gather of 2m doubles -- results only, nothing on the O(n^2 m) path.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    m = xt.shape[0]
    lo, hi = shard_bounds(m, world, rank)
    zpm, zpv = model.predict(xi, zi, xt[lo:hi])
    if not gather or world == 1:
        return zpm, zpv, (lo, hi)
    counts = [shard_bounds(m, world, r)[1] - shard_bounds(m, world, r)[0] for r in range(world)]
    width = max(counts)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    mine = torch.zeros((2, width), dtype=torch.float64, device=dev)
    mine[0, : hi - lo] = torch.as_tensor(np.asarray(zpm), device=dev)
    mine[1, : hi - lo] = torch.as_tensor(np.asarray(zpv), device=dev)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    zpm_all = np.concatenate([p[0, :c].cpu().numpy() for p, c in zip(parts, counts)])
    zpv_all = np.concatenate([p[1, :c].cpu().numpy() for p, c in zip(parts, counts)])
    return zpm_all, zpv_all, (lo, hi)
