#!/usr/bin/env python3
"""BASELINE configs 4 and 5 as a driver: a batch of N independent frame pairs, or an N-frame sequence, partitioned over
the GPUs of one node.

    python examples/sharded_run.py --workload batch --items 10000                      # config 4 shape (1 GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
        examples/sharded_run.py --workload sequence --items 4541 --width 1241 --height 376   # config 5 shape

One process per GPU, no PyTorch in it (torch.distributed.run is only the launcher: the RCCL id travels through a file).  Rank r
owns the contiguous block sharding.shard_range gives it (a sequence block also needs ONE halo frame: its last pair's second
view) and walks it through the chunk pipeline (pipeline.ChunkPipeline: several contexts, frames DMAed from page-locked host
memory, detection and geometry of different chunks overlapping); behind every chunk the ranks all-gather their 128-byte
records over RCCL on the context's stream (vo_pairs_gather); rank 0 chains the gathered relative poses and reports the ATE
(after similarity alignment) against the synthetic ground truth
(tests/scripts/sharded_vs_oracle.py runs this driver and checks its records against the CPU oracle).  KITTI-00 itself is not in the image:
--width 1241 --height 376 renders a KITTI-shaped synthetic flight instead, and says so."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import _lib, sharding, synth  # noqa: E402
from visual_odometry_amd.frontend import FrontEnd  # noqa: E402
from visual_odometry_amd.pipeline import ChunkPipeline  # noqa: E402
from visual_odometry_amd.rendezvous import FileRendezvous, LibraryCollectives, init_library_comm  # noqa: E402


def main(argv=None, on_records=None):
    """on_records(rec, seq, view, args, out): optional hook rank 0 calls with every pair's gathered record before it
    prints the summary (the test scripts hang their checker there; the driver itself never loads one)."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["batch", "sequence"], default="sequence")
    ap.add_argument("--items", type=int, default=1024, help="batch: pairs; sequence: frames")
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--detector", choices=["orb", "sift"], default="orb")
    ap.add_argument("--nfeatures", type=int, default=2000)
    ap.add_argument("--nlevels", type=int, default=8)
    ap.add_argument("--distinct-frames", type=int, default=128, help="rendered views of the closed flight (walked cyclically)")
    ap.add_argument("--chunk", type=int, default=256, help="pairs per resident chunk")
    ap.add_argument("--contexts", type=int, default=3, help="contexts (streams) alternating over the chunks")
    a = ap.parse_args(argv)

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # render before any GPU initialisation (the renderer forks workers); the other ranks read the cache rank 0 wrote
    D = a.distinct_frames
    if rank == 0:
        synth.prerender(D, a.width, a.height, "/tmp", "loop")
    rdv = FileRendezvous(rank, world) if world > 1 else None      # no PyTorch: the communicator id travels through a file
    if rdv:
        rdv.barrier("frames_rendered")
    seq = synth.sequence(D, a.width, a.height, cache_dir="/tmp", trajectory="loop", workers=1)
    K = seq["K"]
    n_items = a.items if a.workload == "batch" else max(a.items - 1, 0)         # items = independent units = pairs

    def view(g):                                         # global frame index -> rendered view (the flight is a closed loop)
        return g % D

    C = a.chunk
    nf = 2 * C if a.workload == "batch" else C + 1
    kw = dict(detector="sift") if a.detector == "sift" else dict(nfeatures=a.nfeatures, nlevels=a.nlevels)
    fes = [FrontEnd(a.height, a.width, max_frames=nf, max_pairs=C, device=local, **kw) for _ in range(max(1, a.contexts))]
    coll = LibraryCollectives(fes[0].ctx, world)
    if world > 1:
        init_library_comm(fes, rdv, rank, world)         # one RCCL communicator per context
    # every rendered view once in page-locked memory: a chunk of consecutive views is a slice of it, DMAed without a host copy
    ring = _lib.PinnedArray((D, a.height, a.width), np.uint8)
    ring.array[...] = seq["frames"]
    pipe = ChunkPipeline(fes, K, world=world, rank=rank, gather="library" if world > 1 else None, gather_rows=C)

    def plan_chunk(lo, hi):
        n = hi - lo
        if a.workload == "sequence":                     # pairs (g, g + 1), g in [lo, hi): frames lo .. hi, hi is the halo
            return dict(pairs=np.stack([np.arange(n), np.arange(n) + 1], 1), n_frames=n + 1, uploads=sharding.ring_uploads(ring.array, lo, n + 1))
        return dict(pairs=np.stack([2 * np.arange(n), 2 * np.arange(n) + 1], 1), n_frames=2 * n,          # independent pair p = views (2p, 2p + 1)
                    uploads=sharding.ring_uploads(ring.array, 2 * lo, 2 * n))

    coll.barrier()
    t0 = time.perf_counter()
    rec = sharding.run_sharded_pipelined(n_items, rank, world, C, pipe, plan_chunk)
    coll.barrier()
    dt = coll.allreduce_max(time.perf_counter() - t0)

    if rank == 0:
        ok = rec[:, 14] >= 0
        out = {"workload": a.workload, "detector": a.detector, "items": a.items, "pairs": n_items, "world": world, "seconds": round(dt, 3),
               "pairs_per_s": round(n_items / dt, 1) if dt > 0 else None, "contexts": len(fes), "chunk": C,
               "failed_pairs": int((~ok).sum()), "mean_inliers": round(float(rec[ok, 14].mean()), 1) if ok.any() else None,
               "frames": f"{a.width}x{a.height} synthetic closed flight, {D} distinct views, DMAed from page-locked host memory" +
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
    coll.barrier()
    if world > 1:
        for f in fes:
            f.ctx.comm_destroy()
        rdv.close()


if __name__ == "__main__":
    main()
