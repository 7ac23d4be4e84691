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


def sharded_plan(n_items: int, world: int, chunk: int):
    """(items per rank, number of chunk rounds).  Every rank runs the same number of rounds because each round ends in
    a collective; a rank whose block is exhausted contributes an empty (padded) chunk."""
    counts = [shard_range(n_items, r, world)[1] - shard_range(n_items, r, world)[0] for r in range(world)]
    return counts, (max(counts) + chunk - 1) // chunk if counts and max(counts) > 0 else 0


def run_sharded(n_items: int, rank: int, world: int, chunk: int, process_chunk, gather_chunk) -> np.ndarray:
    """The N-GPU driver loop of BASELINE configs 4 and 5: this rank's contiguous block of `n_items` independent items
    (frame pairs) is cut into chunks of at most `chunk`; process_chunk(a, b) -> [b - a, RECORD_WIDTH] float64 runs items
    [a, b) (for a consecutive-pair sequence that needs frames a .. b inclusive: b is the halo frame);
    gather_chunk(padded [chunk, RECORD_WIDTH]) -> [world, chunk, RECORD_WIDTH] is the one collective (RCCL through
    FrontEnd.gather_records on the GPUs, gloo in the CPU test).  Returns the records of ALL items in global order,
    identical on every rank."""
    lo, hi = shard_range(n_items, rank, world)
    _, rounds = sharded_plan(n_items, world, chunk)
    out = np.full((n_items, RECORD_WIDTH), np.nan)
    for c in range(rounds):
        a = min(lo + c * chunk, hi); b = min(a + chunk, hi)
        rec = np.zeros((chunk, RECORD_WIDTH), np.float64)
        if b > a:
            mine = np.asarray(process_chunk(a, b), np.float64)
            if mine.shape != (b - a, RECORD_WIDTH):
                raise ValueError(f"process_chunk({a}, {b}) returned {mine.shape}")
            rec[:b - a] = mine
        everyone = np.asarray(gather_chunk(rec)).reshape(world, chunk, RECORD_WIDTH)
        for r in range(world):
            rlo, rhi = shard_range(n_items, r, world)
            ra = min(rlo + c * chunk, rhi); rb = min(ra + chunk, rhi)
            out[ra:rb] = everyone[r, :rb - ra]
    return out


def run_sharded_pipelined(n_items: int, rank: int, world: int, chunk: int, pipeline, plan_chunk) -> np.ndarray:
    """run_sharded on the chunk pipeline (pipeline.ChunkPipeline: several contexts, page-locked uploads, detection and
    geometry of different chunks overlapping, the record gather issued behind each chunk): the loop that runs BASELINE
    configs 4 and 5 at the GPU's streamed rate.  plan_chunk(a, b) -> dict(pairs=[b - a, 2] slot indices, n_frames=slots to
    detect, uploads=[(page-locked frames, first_slot), ...]) describes items [a, b) of this rank's block; a rank whose block
    is exhausted submits empty chunks so that every rank issues the same collectives.  Returns the records of ALL items in
    global order, identical on every rank."""
    lo, hi = shard_range(n_items, rank, world)
    _, rounds = sharded_plan(n_items, world, chunk)
    out = np.full((n_items, RECORD_WIDTH), np.nan)

    def place(ret):
        c = ret.tag
        everyone = ret.gathered if ret.gathered is not None else ret.records()[None]
        for r in range(everyone.shape[0]):
            rlo, rhi = shard_range(n_items, r if everyone.shape[0] > 1 else rank, world)
            ra = min(rlo + c * chunk, rhi); rb = min(ra + chunk, rhi)
            out[ra:rb] = everyone[r, :rb - ra]

    for c in range(rounds):
        a = min(lo + c * chunk, hi); b = min(a + chunk, hi)
        spec = plan_chunk(a, b) if b > a else dict(pairs=np.zeros((0, 2), np.int32), n_frames=0, uploads=None)
        ret = pipeline.submit(spec["pairs"], spec["n_frames"], spec.get("uploads"), tag=c)
        if ret is not None:
            place(ret)
    for ret in pipeline.drain():
        place(ret)
    return out


def ring_uploads(ring, first_view: int, count: int):
    """`count` consecutive views of a closed flight starting at `first_view`, as slices of the page-locked array `ring`
    ([D, H, W]: every rendered view once) -> [(slice, first_slot), ...] for ChunkPipeline.submit: no host copy, at most two
    DMA transfers per wrap of the ring."""
    D = len(ring)
    out, slot, v = [], 0, first_view % D
    while count > 0:
        n = min(count, D - v)
        out.append((ring[v:v + n], slot))
        slot += n; count -= n; v = 0
    return out


def records_to_trajectory(rec: np.ndarray):
    """Chain the gathered relative poses x_{k+1} ~ R_k x_k + t_k (unit-norm t) into camera centres; pairs whose
    n_inl column is negative (failed) repeat the previous step.  Returns (centres [n+1, 3], number of failed pairs)."""
    T = np.eye(4); centres = [T[:3, 3].copy()]; bad = 0
    step = np.eye(4)
    for row in rec:
        if row[14] >= 0 and np.all(np.isfinite(row[:12])):
            step = np.eye(4); step[:3, :3] = row[:9].reshape(3, 3); step[:3, 3] = row[9:12]
        else:
            bad += 1
        T = T @ np.linalg.inv(step)
        centres.append(T[:3, 3].copy())
    return np.stack(centres), bad


def ate_after_alignment(est: np.ndarray, ref: np.ndarray) -> float:
    """RMS distance between two centre trajectories after the best similarity alignment (Umeyama): monocular scale is
    unobservable, so this is the ATE SURVEY 8(e) prescribes for the chained relative poses."""
    a, b = est - est.mean(0), ref - ref.mean(0)
    u, sv, vt = np.linalg.svd(b.T @ a / len(a))
    d = np.eye(3); d[2, 2] = np.sign(np.linalg.det(u @ vt))
    rot = u @ d @ vt
    scale = np.trace(np.diag(sv) @ d) / max((a ** 2).sum() / len(a), 1e-300)
    return float(np.sqrt((((scale * (rot @ a.T)).T - b) ** 2).sum(1).mean()))


def pack_records(results) -> np.ndarray:
    rec = np.zeros((len(results), RECORD_WIDTH), np.float64)
    rec[:, :9] = results["R"]; rec[:, 9:12] = results["t"]
    rec[:, 12] = results["n_kp1"]; rec[:, 13] = results["n_match"]
    rec[:, 14] = np.where(results["status"] == 0, results["n_inl"], results["status"])   # as k_pack_records does on the device
    rec[:, 15] = results["n_good"]
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
