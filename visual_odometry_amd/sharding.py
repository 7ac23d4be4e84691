"""Pair sharding across the GPUs of one node: one process per GPU, contiguous blocks of independent frame
pairs, one halo frame per rank for consecutive-pair sequences, and a single gather of the per-pair records
at the end (RCCL on the GPU box; gloo in the CPU tests).  No data-path collective."""
from __future__ import annotations

import numpy as np

RECORD_WIDTH = 16      # R (9) + t (3) + n_kp1, n_match, n_inl, n_good as float64: 128 B per pair


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous block partition: the first (n % world) ranks get one extra item."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def sequence_shard(n_frames: int, rank: int, world: int):
    """For a sequence of n_frames (pairs k -> k+1): (pair_lo, pair_hi, frame_lo, frame_hi_exclusive).
    Each rank needs its pairs' frames plus one halo frame."""
    lo, hi = shard_range(max(n_frames - 1, 0), rank, world)
    return lo, hi, lo, (hi + 1 if hi > lo else lo)


def pack_records(results) -> np.ndarray:
    rec = np.zeros((len(results), RECORD_WIDTH), np.float64)
    rec[:, :9] = results["R"]; rec[:, 9:12] = results["t"]
    rec[:, 12] = results["n_kp1"]; rec[:, 13] = results["n_match"]
    rec[:, 14] = results["n_inl"]; rec[:, 15] = results["n_good"]
    return rec


def gather_records(rec: np.ndarray, counts, dist=None, device=None):
    """All-gather variable-length per-rank record blocks (counts[r] rows from rank r) into one array in
    pair order.  `dist` is torch.distributed (None = single process)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return rec.copy()
    import torch
    world = dist.get_world_size()
    m = int(max(counts))
    pad = np.zeros((m, RECORD_WIDTH), np.float64)
    pad[:len(rec)] = rec
    mine = torch.from_numpy(pad)
    if device is not None:
        mine = mine.to(device)
    out = torch.empty((world * m, RECORD_WIDTH), dtype=torch.float64, device=mine.device)
    dist.all_gather_into_tensor(out, mine)
    out = out.cpu().numpy().reshape(world, m, RECORD_WIDTH)
    return np.concatenate([out[r, :counts[r]] for r in range(world)], axis=0)
