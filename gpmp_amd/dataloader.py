"""Sharded dataset + mini-batch loader for the batch selection criteria -- counterpart of the parts of
gpmp/dataloader.py the parameter-selection procedures use (``Dataset``: x_list / z_list shards, reductions over shards;
``DataLoader``: sized iterable of ``(x_batch, z_batch)``, ``reduce_mean``).  Host-side bookkeeping only: the shards are
device tensors, a batch is a gather of rows.  The scalers / k-fold helpers of the reference module are not carried.
"""
import bisect

import numpy as np

from . import num as gnp


class Dataset:
    """gpmp/dataloader.py:55-120: observations held as one or several shards ``(x_k, z_k)``."""

    def __init__(self, x, z):
        xs = x if isinstance(x, list) else [x]
        zs = z if isinstance(z, list) else [z]
        if len(xs) != len(zs):
            raise ValueError("x and z shard counts differ")
        self.x_list = [gnp.asarray(v) for v in xs]
        self.z_list = [gnp.asarray(v) for v in zs]
        for a, b in zip(self.x_list, self.z_list):
            if a.shape[0] != b.shape[0]:
                raise ValueError("shard length mismatch")
        self.size = sum(a.shape[0] for a in self.x_list)
        self._shard_bounds = list(np.cumsum([a.shape[0] for a in self.x_list]))

    def __len__(self):
        return self.size

    def __getitem__(self, idx):
        k = bisect.bisect_right(self._shard_bounds, idx)
        start = 0 if k == 0 else self._shard_bounds[k - 1]
        return self.x_list[k][idx - start], self.z_list[k][idx - start]

    def __repr__(self):
        return f"Dataset(size={self.size}, shards={len(self.x_list)})"

    def _shards(self, which):
        return self.x_list if which == "x" else self.z_list

    def _reduce_min(self, which):
        out = None
        for s in self._shards(which):
            m = gnp.min(s, axis=0)
            out = m if out is None else gnp.minimum(out, m)
        return out

    def _reduce_max(self, which):
        out = None
        for s in self._shards(which):
            m = gnp.max(s, axis=0)
            out = m if out is None else gnp.maximum(out, m)
        return out

    def _reduce_mean(self, which):
        return sum(gnp.sum(s, axis=0) for s in self._shards(which)) / self.size

    def _reduce_var(self, which):
        mu = self._reduce_mean(which)
        return sum(gnp.sum((s - mu) ** 2, axis=0) for s in self._shards(which)) / (self.size - 1)

    def _reduce_std(self, which):
        return gnp.sqrt(self._reduce_var(which))


class DataLoader:
    """gpmp/dataloader.py:322-536: batches of ``batch_size`` rows in dataset order or (``shuffle``) in a fresh
    permutation per epoch (seeded by ``seed + epoch`` when a seed is given); ``len()`` = batches per epoch."""

    def __init__(self, dataset, batch_size=None, shuffle=True, drop_last=False, seed=None, infinite=False):
        self.dataset = dataset
        batch_size = len(dataset) if batch_size is None else int(batch_size)
        if batch_size <= 0:
            raise ValueError("batch_size must be a positive integer.")
        self.batch_size, self.shuffle, self.drop_last = batch_size, shuffle, drop_last
        self._base_seed, self._epoch, self._infinite = seed, 0, infinite

    def set_epoch(self, epoch):
        self._epoch = epoch

    def __len__(self):
        n = len(self.dataset)
        full = n // self.batch_size
        return full if (self.drop_last or n % self.batch_size == 0) else full + 1

    def _fetch_batch(self, idx):
        idx = np.asarray(idx)
        xs, zs, start = [], [], 0
        for k, end in enumerate(self.dataset._shard_bounds):
            sel = idx[(idx >= start) & (idx < end)] - start
            if sel.size:
                rows = gnp.asarray(sel.astype(np.int64))
                xs.append(self.dataset.x_list[k][rows])
                zs.append(self.dataset.z_list[k][rows])
            start = end
        return gnp.concatenate(xs, 0), gnp.concatenate(zs, 0)

    def __iter__(self):
        while True:
            n = len(self.dataset)
            if self.shuffle:
                rng = np.random.default_rng(None if self._base_seed is None else self._base_seed + self._epoch)
                order = rng.permutation(n)
            else:
                order = np.arange(n)
            for start in range(0, n, self.batch_size):
                if start + self.batch_size > n and self.drop_last:
                    break
                yield self._fetch_batch(order[start:start + self.batch_size])
            self._epoch += 1
            if not self._infinite:
                break

    def reduce_mean(self, func):
        """gpmp/dataloader.py:484-513: batch-size weighted mean of ``func(x_batch, z_batch)`` over one epoch."""
        total, weight = None, 0
        for xb, zb in self:
            v = gnp.asarray(func(xb, zb)) * xb.shape[0]
            total = v if total is None else total + v
            weight += xb.shape[0]
        return total / weight
