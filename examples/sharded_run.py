#!/usr/bin/env python3
"""BASELINE configs 4 and 5 as a driver: a batch of N independent frame pairs, or an N-frame sequence, partitioned over
the GPUs of one node.

    python examples/sharded_run.py --workload batch --items 10000                      # config 4 shape (1 GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
        examples/sharded_run.py --workload sequence --items 4541 --width 1241 --height 376   # config 5 shape

One process per GPU.  Rank r owns the contiguous block sharding.shard_range gives it (a sequence block also needs ONE
halo frame: its last pair's second view), walks it in chunks that stay resident in HBM, and after every chunk the
ranks all-gather their 128-byte records over RCCL (FrontEnd.gather_records -> vo_pairs_gather); rank 0 chains the
gathered relative poses and reports the ATE (after similarity alignment) against the synthetic ground truth
(tests/scripts/sharded_vs_oracle.py runs this driver and checks its records against the CPU oracle).  KITTI-00 itself is not in the image:
--width 1241 --height 376 renders a KITTI-shaped synthetic flight instead, and says so."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import sharding, synth  # noqa: E402
from visual_odometry_amd.frontend import FrontEnd  # noqa: E402


def main(argv=None, on_records=None):
    """on_records(rec, seq, view, args, out): optional hook rank 0 calls with every pair's gathered record before it
    prints the summary (the test scripts hang their checker there; the driver itself never loads one)."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["batch", "sequence"], default="sequence")
    ap.add_argument("--items", type=int, default=1024, help="batch: pairs; sequence: frames")
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--nfeatures", type=int, default=2000)
    ap.add_argument("--nlevels", type=int, default=8)
    ap.add_argument("--distinct-frames", type=int, default=128, help="rendered views of the closed flight (tiled if fewer than needed)")
    ap.add_argument("--chunk", type=int, default=256, help="pairs per resident chunk")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl")
    a = ap.parse_args(argv)

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # render before any GPU / process-group initialisation (the renderer forks workers); the other ranks read the cache
    D = a.distinct_frames
    if rank == 0:
        synth.prerender(D, a.width, a.height, "/tmp", "loop")
    seq = None
    dist = torch = None
    if world > 1:
        import torch
        import torch.distributed as dist
        if a.dist_backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    device = local if a.dist_backend == "nccl" else 0

    if dist:
        dist.barrier()
    if seq is None:
        seq = synth.sequence(D, a.width, a.height, cache_dir="/tmp", trajectory="loop", workers=1)
    K = seq["K"]
    n_items = a.items if a.workload == "batch" else max(a.items - 1, 0)         # items = independent units = pairs

    def view(g):                                         # global frame index -> rendered view (the flight is a closed loop)
        return g % D

    C = a.chunk
    fe = FrontEnd(a.height, a.width, max_frames=2 * C if a.workload == "batch" else C + 1, max_pairs=C,
                  nfeatures=a.nfeatures, nlevels=a.nlevels, device=device)
    use_lib = world > 1 and a.dist_backend == "nccl"
    if use_lib:
        ident = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            ident.copy_(torch.frombuffer(bytearray(fe.ctx.comm_unique_id()), dtype=torch.uint8))
        dist.broadcast(ident, 0)
        fe.ctx.comm_init(ident.cpu().numpy().tobytes(), rank, world)

    def process_chunk(lo, hi):
        n = hi - lo
        if a.workload == "sequence":                     # pairs (g, g + 1), g in [lo, hi): frames lo .. hi, hi is the halo
            idx = np.array([view(g) for g in range(lo, hi + 1)])
            pairs = np.stack([np.arange(n), np.arange(n) + 1], 1)
        else:                                            # independent pair p = views (2p, 2p + 1): both frames detected
            idx = np.array([view(2 * p + k) for p in range(lo, hi) for k in (0, 1)])
            pairs = np.stack([2 * np.arange(n), 2 * np.arange(n) + 1], 1)
        fe.upload(seq["frames"][idx])
        fe.detect(0, len(idx))
        res, _ = fe.run_pairs(pairs.astype(np.int32), K)
        return sharding.pack_records(res)

    def gather_chunk(rec):
        if world == 1:
            return rec[None]
        if use_lib:                                      # the device-side records of the chunk just run, over RCCL
            return fe.gather_records(C, world, wait=True).copy()
        mine = torch.from_numpy(rec)
        out = torch.empty((world * C, sharding.RECORD_WIDTH), dtype=torch.float64)
        dist.all_gather_into_tensor(out, mine)
        return out.numpy().reshape(world, C, sharding.RECORD_WIDTH)

    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    rec = sharding.run_sharded(n_items, rank, world, C, process_chunk, gather_chunk)
    if dist:
        dist.barrier()
    dt = time.perf_counter() - t0

    if rank == 0:
        out = {"workload": a.workload, "items": a.items, "pairs": n_items, "world": world, "seconds": round(dt, 3),
               "pairs_per_s": round(n_items / dt, 1) if dt > 0 else None,
               "failed_pairs": int((rec[:, 14] < 0).sum()), "mean_inliers": round(float(rec[rec[:, 14] >= 0, 14].mean()), 1),
               "frames": f"{a.width}x{a.height} synthetic closed flight, {D} distinct views" +
                         (" (KITTI-shaped stand-in: the dataset is not in the image)" if (a.width, a.height) == (1241, 376) else "")}
        if a.workload == "sequence":
            centres, bad = sharding.records_to_trajectory(rec)
            gt_rec = np.zeros((n_items, sharding.RECORD_WIDTH))
            for g in range(n_items):
                R, t = synth.relative_pose(seq["R"][view(g)], seq["C"][view(g)], seq["R"][view(g + 1)], seq["C"][view(g + 1)])
                gt_rec[g, :9] = R.ravel(); gt_rec[g, 9:12] = t
            gt_centres, _ = sharding.records_to_trajectory(gt_rec)
            span = float(np.linalg.norm(gt_centres.max(0) - gt_centres.min(0)))
            out["ate_vs_ground_truth"] = round(sharding.ate_after_alignment(centres, gt_centres), 4)
            out["trajectory_extent_unit_steps"] = round(span, 1)
        if on_records is not None:
            on_records(rec, seq, view, a, out)
        print(json.dumps(out), flush=True)
    if dist:
        if use_lib:
            fe.ctx.comm_destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
