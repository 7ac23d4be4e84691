#!/usr/bin/env python3
"""bench.py — frame-pairs/sec of the MI355X visual-odometry front end (BASELINE.json metric).

A step = one pass of the whole per-pair hot path (ORB detect+describe of every frame of a chunk, Hamming
matching, 5-point E-RANSAC, recoverPose, DLT triangulation, result download) over one chunk of a seeded
synthetic 1280x720 drone sequence that is already resident in HBM.  N > 1: one process per GPU, every rank
runs its own chunk (weak scaling, no data-path collective); the per-pair [R|t] + counts records are
all-gathered over RCCL each step (the trajectory gather).

Prints ONE JSON line on rank 0 (see the driver contract); adds `roofline` for the dominant kernel (HIP
events on the library's stream, algorithmic bytes from vo_stage_bytes) and `cpu_baseline` (the CPU oracle,
oracle/libvoo.so, timed on the host cores on a bounded sample of the same workload).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def ping_pong(n_distinct, n_total, start=0):
    """Frame indices walking 0..n-1..0.. so consecutive entries are always neighbouring views."""
    period = 2 * (n_distinct - 1)
    idx = []
    for k in range(n_total):
        m = (start + k) % period
        idx.append(m if m < n_distinct else period - m)
    return np.array(idx)


STAGE_KERNELS = {"fast_score_nms": [("k_fast<false>", 1)], "gaussian_blur": [("k_blur", 1)],
                 "pyramid_resize": [("k_resize_tiled", 7)],
                 "select_fast": [("k_sel_threshold", 1), ("k_sel_rows<false>", 1), ("k_sel_rows<true>", 1)],
                 "match_nn": [("k_nn_mfma<false>", 1)], "essential_ransac": [("k_ransac", 1)]}


def pmc_traffic(stage, nframes):
    """HBM bytes per launch of `stage` from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, KiB ->
    bytes, tools/collect_traffic.py); None when no pass exists for this workload size."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        t = json.load(open(path))
        if t.get("_meta", {}).get("frames_per_launch") != nframes:
            return None
        return float(sum(t[k]["hbm_bytes_per_launch"] * n for k, n in STAGE_KERNELS[stage]))
    except (OSError, KeyError, ValueError):
        return None


VALU_PEAK_TLANEOPS = 35.0   # measured on MI355X by tools/ubench/valu_rates.hip (profiles/r01_valu_issue_rates.txt)


def pmc_valu(stage):
    """VALU lane-operations per launch of `stage` (SQ_INSTS_VALU x 64) from the committed PMC pass, or None."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        return float(sum(t[k]["valu_wave_insts_per_launch"] * n for k, n in STAGE_KERNELS[stage])) * 64.0
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(frames, K, nfeatures, nlevels, match_mode, ratio, budget_s=12.0):
    """The CPU oracle ("port") on the host cores over a bounded sample of the same pairs."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    O.lib()
    cores = max(1, min(os.cpu_count() or 1, 16))
    p = O.orb_params(nfeatures=nfeatures, nlevels=nlevels)
    n_pairs = len(frames) - 1
    t0 = time.perf_counter()
    O.pair(frames[0], frames[1], p, K, match_mode=match_mode, ratio=ratio, want_points=True)   # warm + cost probe
    one = time.perf_counter() - t0
    sample = int(max(cores, min(4000, cores * max(1, int(budget_s / max(one, 1e-3))))))
    jobs = [(i % n_pairs) for i in range(sample)]

    def work(i):
        O.pair(frames[i], frames[i + 1], p, K, match_mode=match_mode, ratio=ratio, want_points=True)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(work, jobs))
    dt = time.perf_counter() - t0
    return {"value": round(sample / dt, 3), "unit": "frame-pairs/s", "cores": cores, "kind": "port",
            "sample": f"{sample} pairs of the same 1280x720 sequence via oracle/libvoo.so (scalar C restatement "
                      f"of the cv2 path), {cores} host threads, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--pairs-per-step", type=int, default=256)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--nfeatures", type=int, default=2000)
    ap.add_argument("--nlevels", type=int, default=8)
    ap.add_argument("--distinct-frames", type=int, default=17)
    ap.add_argument("--matcher", choices=["crosscheck", "ratio"], default="crosscheck")
    ap.add_argument("--ratio", type=float, default=0.8)
    ap.add_argument("--matcher-kernel", choices=["mfma", "popcount"], default="mfma",
                    help="Hamming NN kernel: int8 MFMA over +1/-1 bytes (default) or XOR + popcount (same results)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-stream-pass", action="store_true", help="skip the frames-streamed-from-host measurement")
    ap.add_argument("--workload", choices=["sequence", "independent"], default="sequence",
                    help="sequence: C+1 consecutive frames -> C pairs, each frame detected once (BASELINE config 2); "
                         "independent: C pairs with their own two frames each, 2C detections (BASELINE config 4 accounting)")
    ap.add_argument("--contexts", type=int, default=2, help="contexts (streams) per GPU alternating over the chunks")
    ap.add_argument("--chain-detect", type=int, default=1,
                    help="1: a context's detection starts after the previous context's detection (software pipeline)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed and run the trajectory gather even with one rank (exercises RCCL on a 1-GPU box)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearse the multi-process path on a box with fewer GPUs than ranks (all ranks share GPU 0)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    dist = torch = None
    device = local_rank
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            device = 0
            dist.init_process_group("gloo", rank=rank, world_size=world)
    on_gpu = use_dist and args.dist_backend == "nccl"

    from visual_odometry_amd import synth
    from visual_odometry_amd.frontend import FrontEnd, MATCH_CROSSCHECK, MATCH_RATIO, chain_poses

    C = args.pairs_per_step
    seq = synth.sequence(args.distinct_frames, args.width, args.height, cache_dir="/tmp")
    K = seq["K"]
    if args.workload == "sequence":
        order = ping_pong(args.distinct_frames, C + 1, start=rank * 3)
        frames = seq["frames"][order]                     # chunk of C+1 consecutive views -> C pairs
        pairs = np.stack([np.arange(C), np.arange(C) + 1], axis=1).astype(np.int32)
    else:
        a = ping_pong(args.distinct_frames, C, start=rank * 3)
        b = ping_pong(args.distinct_frames, C, start=rank * 3 + 1)
        frames = seq["frames"][np.stack([a, b], axis=1).ravel()]      # 2C frames, pair k = slots (2k, 2k+1)
        pairs = np.stack([2 * np.arange(C), 2 * np.arange(C) + 1], axis=1).astype(np.int32)
    NF = len(frames)
    match_mode = MATCH_CROSSCHECK if args.matcher == "crosscheck" else MATCH_RATIO

    # Two contexts (two HIP streams, two sets of resident buffers) on the GPU: while one chunk is in its
    # latency-bound RANSAC / pose kernels the other chunk's streaming ORB kernels fill the machine.
    n_ctx = max(1, args.contexts)
    fes = []
    for c in range(n_ctx):
        fe_c = FrontEnd(args.height, args.width, max_frames=NF, max_pairs=C, nfeatures=args.nfeatures,
                        nlevels=args.nlevels, device=device)
        t_up = time.perf_counter()
        fe_c.upload(frames)                               # inputs resident in HBM before the timed region
        upload_s = time.perf_counter() - t_up
        fe_c.ctx.set_matcher_kernel(args.matcher_kernel)
        fes.append(fe_c)
    fe = fes[0]
    opts = fe.make_opts(match_mode=match_mode, ratio=args.ratio, want_points=True)

    # per pair: R (9) t (3) n_kp1 n_match n_inl n_good.  Page-locked staging + preallocated device tensors: the gather
    # costs one asynchronous 32 KB copy and one collective per step, nothing that waits for the GPU.
    gathered = None
    if use_dist:
        rec_t = torch.zeros((C, 16), dtype=torch.float64, pin_memory=on_gpu)
        rec = rec_t.numpy()
        mine_d = torch.empty((C, 16), dtype=torch.float64, device="cuda") if on_gpu else None
        out_d = torch.empty((world * C, 16), dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        copied = torch.cuda.Event() if on_gpu else None   # the staging buffer is free again once this has fired
    in_flight = [None] * n_ctx
    counter = [0]

    def consume(res):
        nonlocal gathered
        if use_dist:                                      # trajectory gather over RCCL / xGMI: 128 B per pair
            if on_gpu and gathered is not None:
                copied.synchronize()
            rec[:, :9] = res["R"]; rec[:, 9:12] = res["t"]
            rec[:, 12] = res["n_kp1"]; rec[:, 13] = res["n_match"]; rec[:, 14] = res["n_inl"]; rec[:, 15] = res["n_good"]
            if on_gpu:
                mine_d.copy_(rec_t, non_blocking=True)
                copied.record()
                dist.all_gather_into_tensor(out_d, mine_d)
            else:
                dist.all_gather_into_tensor(out_d, rec_t)
            gathered = out_d

    staged = [None] * n_ctx                               # page-locked copies of the chunk (streamed-from-host pass)

    def step(stream_frames=False):
        """Enqueue one chunk on the next context; first retire (wait + gather) the chunk that context ran before."""
        k = counter[0] % n_ctx
        counter[0] += 1
        f = fes[k]
        if in_flight[k] is not None:
            f.wait()
            consume(in_flight[k])
        if stream_frames:
            f.upload(staged[k].array, wait=False)         # DMA from pinned host memory, beside the other context's kernels
        f.detect(0, NF, wait=False, after=fes[(k - 1) % n_ctx] if args.chain_detect else None)
        in_flight[k], _ = f.run_pairs(pairs, K, opts, wait=False)

    def drain():
        last = None
        for j in range(n_ctx):
            k = (counter[0] + j) % n_ctx                  # oldest first
            if in_flight[k] is not None:
                fes[k].wait()
                consume(in_flight[k])
                last = in_flight[k]
                in_flight[k] = None
        return last

    def sync():
        if use_dist:
            dist.barrier()
            if on_gpu:
                torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    drain()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    res = drain()                                         # every enqueued chunk finished, results on the host
    sync()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    res = res.copy()

    # The same steps with every chunk's frames coming from (page-locked) host memory: the PCIe-inclusive rate.
    # Not `value` (the contract times HBM-resident inputs); reported in config.
    streamed = None
    if not args.no_stream_pass:
        for k in range(n_ctx):
            staged[k] = fes[k].pinned_frames(NF)
            staged[k].array[...] = frames
        for _ in range(args.warmup):
            step(True)
        drain()
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step(True)
        drain()
        sync()
        streamed = C * args.steps / (time.perf_counter() - t1)

    # Per-kernel durations for the roofline: HIP events on the library's stream around every kernel family.
    # With several contexts the timed region overlaps kernels of different streams, which stretches every
    # bracket, so the stage times are taken in a pass of the same step on ONE context right after the timed
    # region (same process, same resident data); with --contexts 1 that pass repeats the timed workload as is.
    prof, prof_steps = {}, max(1, min(args.steps, 5))
    if not args.no_profile:
        fe.profile(True)
        for _ in range(prof_steps):
            fe.detect(0, NF, wait=False)
            fe.run_pairs(pairs, K, opts)
        prof = fe.profile_read()
        fe.profile(False)

    ok = int((res["status"] == 0).sum())
    if rank == 0:
        value = world * C * args.steps / dt
        line = {
            "metric": "frame-pairs/sec (1280x720, 2000 ORB feats)", "value": round(value, 2), "unit": "frame-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8+f32+f64",
            "data": "synthetic",
            "config": {"workload": f"BASELINE config 2: seeded synthetic {args.width}x{args.height} drone sequence, "
                                   f"{args.nfeatures} ORB features/frame, {args.nlevels} levels; " +
                                   (f"chunk of {C + 1} consecutive frames -> {C} pairs per GPU per step, each frame detected once"
                                    if args.workload == "sequence" else
                                    f"{C} independent pairs per GPU per step, both frames of every pair detected ({NF} detections)"),
                       "pairs_per_step_per_gpu": C, "distinct_rendered_frames": args.distinct_frames,
                       "contexts_per_gpu": n_ctx,
                       "matcher": args.matcher, "matcher_kernel": args.matcher_kernel, "ransac": "5-point, conf 0.99, 1 px, seed 2^64-1, <=1000 iters",
                       "parallelism": f"pair-sharded x{world}, RCCL all_gather of 128 B/pair per step" if world > 1 else "single GPU",
                       "streamed_from_host_pairs_per_s_per_gpu": round(streamed, 1) if streamed else None,
                       "unoverlapped_pageable_upload_ms_per_chunk": round(1000 * upload_s, 2),
                       "pairs_ok_last_step": ok,
                       "mean_inliers_last_step": round(float(res["n_inl"].mean()), 1)},
        }
        if prof:
            # per-stage HIP-event times over the timed region; dominant kernel = largest share
            stages = {k: {"ms_per_launch": round(ms / n, 4), "launches": n, "ms_total": round(ms, 3)} for k, (ms, n) in prof.items()}
            dom = max((k for k in prof if k != "misc"), key=lambda k: prof[k][0])
            ms, n = prof[dom]
            nframes = NF
            b = fe.stage_bytes(dom, nframes)
            ach = b / (ms / n * 1e-3) / 1e9 if b > 0 and ms > 0 else 0.0
            line["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                                "traffic": pmc_traffic(dom, nframes),
                                "algorithmic_bytes_per_launch": b, "avg_launch_ms": round(ms / n, 4),
                                "measured": f"HIP events, {prof_steps} single-context steps right after the timed region "
                                            f"(the timed region overlaps {n_ctx} contexts)"}
            hbm_stages = {}
            for k in ("pyramid_resize", "fast_score_nms", "gaussian_blur"):
                if k in prof:
                    bb = fe.stage_bytes(k, nframes)
                    hbm_stages[k] = round(bb / (prof[k][0] / prof[k][1] * 1e-3) / 1e9, 1)
            lane_ops = pmc_valu(dom) if pmc_traffic(dom, nframes) is not None else None
            if lane_ops:
                rate = lane_ops / (ms / n * 1e-3) / 1e12
                line["roofline"]["valu"] = {
                    "lane_ops_per_launch": lane_ops, "achieved_Tlaneops": round(rate, 2),
                    "measured_issue_rate_Tlaneops": {"typical_int_op": VALU_PEAK_TLANEOPS, "packed_16bit_and_sdwa_forms": 65.0},
                    "ratio_to_typical_rate": round(rate / VALU_PEAK_TLANEOPS, 3),
                    "note": "SQ_INSTS_VALU x 64 (committed PMC pass) / event time; issue rates from tools/ubench/valu_rates.hip: "
                            "the kernel is bound by integer VALU issue, not by HBM"}
            line["stages"] = stages
            line["streaming_kernels_GBps"] = hbm_stages
        if not args.no_cpu_baseline and world == 1:
            n_cpu = min(args.distinct_frames, 9)
            line["cpu_baseline"] = cpu_baseline(seq["frames"][:n_cpu], K, args.nfeatures, args.nlevels, match_mode, args.ratio)
        if gathered is not None:
            traj = chain_poses(gathered[:, :9].reshape(-1, 3, 3).cpu().numpy(), gathered[:, 9:12].cpu().numpy())
            line["config"]["trajectory_poses_gathered"] = int(traj.shape[0])
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
