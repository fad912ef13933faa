"""Per-rank work split for batches of independent molecules.

Molecules never exchange data, so multi-GPU voxelization is a partition of the batch: one process
per GPU, each voxelizing its own contiguous chunk into its own device-resident (B_r, C, D, D, D)
tensor. There is no data-path collective (no RCCL traffic over xGMI); torch.distributed is only used
by callers that want a barrier or to collect counts.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(num_items: int, world_size: int) -> np.ndarray:
    """Boundaries (world_size + 1,) of contiguous, near-equal chunks; the first `rem` ranks get one more."""
    assert num_items >= 0 and world_size >= 1
    base, rem = divmod(num_items, world_size)
    sizes = np.full(world_size, base, dtype=np.int64)
    sizes[:rem] += 1
    return np.concatenate([[0], np.cumsum(sizes)])


def shard_range(num_items: int, rank: int, world_size: int) -> tuple[int, int]:
    b = shard_bounds(num_items, world_size)
    return int(b[rank]), int(b[rank + 1])


def balanced_shard_bounds(weights, world_size: int) -> np.ndarray:
    """Contiguous chunks balanced by a per-molecule cost (e.g. atom counts): boundaries where the
    cumulative weight crosses k/world_size of the total. Order is preserved, every item is owned once."""
    w = np.asarray(weights, dtype=np.float64)
    n = w.shape[0]
    if n == 0:
        return np.zeros(world_size + 1, dtype=np.int64)
    cum = np.concatenate([[0.0], np.cumsum(w)])
    targets = cum[-1] * np.arange(1, world_size) / world_size
    cuts = np.searchsorted(cum, targets, side="left")
    b = np.concatenate([[0], cuts, [n]]).astype(np.int64)
    return np.maximum.accumulate(b)


def local_offsets(offsets, lo: int, hi: int) -> np.ndarray:
    """Atom offsets of molecules [lo, hi) rebased to start at 0 (offsets: global (B+1,) array)."""
    off = np.asarray(offsets, dtype=np.int64)
    return off[lo : hi + 1] - off[lo]
