#!/usr/bin/env python3
"""bench.py — frame-pairs/sec of the MI355X visual-odometry front end (BASELINE.json metric).

A step = one pass of the whole per-pair hot path (ORB detect+describe of every frame of a chunk, Hamming
matching, 5-point E-RANSAC, recoverPose, DLT triangulation, result download) over one chunk of a seeded synthetic
drone sequence (a closed flight of --distinct-frames rendered views) that is already resident in HBM.  N > 1: one
process per GPU, every rank runs its own chunk (weak scaling, no data-path collective); the per-pair [R|t] + counts
records (128 B per pair) are all-gathered over RCCL each step by the library itself (vo_pairs_gather: packed on the
device, ncclAllGather on the context stream; torch.distributed only carries the barrier / max-reduce of the timing
contract and the 128-byte communicator id).

Prints ONE JSON line on rank 0 (see the driver contract).  `value` times HBM-resident inputs, as the contract
says; the PCIe-inclusive rate of the same loop is `value_streamed_from_host`.  Also in the line: `roofline` for the
dominant kernel (HIP events on the library's stream, algorithmic bytes from vo_stage_bytes, PMC traffic from the
committed rocprofv3 pass), `cpu_baseline` (the CPU oracle timed on the host cores: one thread and all of them),
`config.sustained` (>= 5 s of back-to-back steps, 3 repeats, median) and the RANSAC iteration histogram.
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
PROFILE_TAG = "r02"       # profiles/<tag>_pmc_traffic.json holds this round's rocprofv3 PMC passes

STAGE_KERNELS = {"fast_score_nms": [("k_fast<false>", 1)], "gaussian_blur": [("k_blur", 1)],
                 "pyramid_resize": [("k_resize_strip", 7)],
                 "select_fast": [("k_sel_threshold", 1), ("k_sel_rows<false>", 1), ("k_sel_rows<true>", 1)],
                 "match_nn": [("k_nn_mfma<false>", 1)], "essential_ransac": [("k_ransac", 1)]}


def _pmc():
    for tag in (PROFILE_TAG, "r01"):
        try:
            return json.load(open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json"))), tag
        except (OSError, ValueError):
            continue
    return None, None


def pmc_traffic(stage, nframes):
    """HBM bytes per launch of `stage` from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, KiB ->
    bytes, tools/collect_traffic.py); None when no pass exists for this workload size."""
    t, _ = _pmc()
    try:
        if t is None or t.get("_meta", {}).get("frames_per_launch") != nframes:
            return None
        return float(sum(t[k]["hbm_bytes_per_launch"] * n for k, n in STAGE_KERNELS[stage]))
    except KeyError:
        return None


def pmc_valu(stage):
    """VALU wave-instructions per launch of `stage` (SQ_INSTS_VALU) from the committed PMC pass, or None."""
    t, _ = _pmc()
    try:
        return float(sum(t[k]["valu_wave_insts_per_launch"] * n for k, n in STAGE_KERNELS[stage])) if t else None
    except KeyError:
        return None


def usable_cores():
    """Host threads this process may really use: the affinity mask capped by the cgroup CPU quota (the GPU box shows
    all of the host's CPUs but grants a share of them)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0]); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(n, 64))


def cpu_baseline(frames, K, nfeatures, nlevels, match_mode, ratio, width, height, budget_s=10.0):
    """The CPU oracle ("port": scalar C restatement of the cv2 path, faithful 300-sweep root finder) on the host
    cores over a bounded sample of the same pairs: first one thread, then every core the process may use."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    O.lib()
    host = os.cpu_count() or 1
    cores = usable_cores()
    p = O.orb_params(nfeatures=nfeatures, nlevels=nlevels)
    n_pairs = len(frames) - 1

    def work(i):
        O.pair(frames[i], frames[i + 1], p, K, match_mode=match_mode, ratio=ratio, want_points=True)
    work(0)                                                          # warm (library load, page faults)
    t0 = time.perf_counter(); n1 = 0
    while time.perf_counter() - t0 < 3.0 or n1 < 2:                  # single thread: >= 3 s
        work(n1 % n_pairs); n1 += 1
    one = (time.perf_counter() - t0) / n1
    deadline = time.perf_counter() + budget_s                        # all cores: every thread works until the deadline
    done = [0] * cores

    def loop(t):
        i = t
        while time.perf_counter() < deadline:
            work(i % n_pairs); i += cores; done[t] += 1
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:                            # ctypes releases the GIL inside the oracle
        list(ex.map(loop, range(cores)))
    dt = time.perf_counter() - t0
    sample = sum(done)
    return {"value": round(sample / dt, 3), "unit": "frame-pairs/s", "cores": cores, "kind": "port",
            "single_thread_value": round(1.0 / one, 3), "host_cpu_count": host,
            "sample": f"{sample} pairs of the same {width}x{height} sequence via oracle/libvoo.so (scalar C restatement of the "
                      f"cv2 path, cv::solvePoly's 300 sweeps), {cores} host threads, {dt:.1f} s; single thread: {n1} pairs, "
                      f"{one * n1:.1f} s; cv2 is not importable on this box"}


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
    ap.add_argument("--distinct-frames", type=int, default=256,
                    help="rendered views of the closed synthetic flight; a chunk walks consecutive views (wrapping)")
    ap.add_argument("--pair-stride", type=int, default=1,
                    help="pair view k with view k + stride: a wider baseline lowers the inlier ratio and multiplies the RANSAC rounds")
    ap.add_argument("--matcher", choices=["crosscheck", "ratio", "crosscheck-legacy"], default="crosscheck")
    ap.add_argument("--ratio", type=float, default=0.8)
    ap.add_argument("--matcher-kernel", choices=["mfma_fp4", "mfma", "popcount"], default="mfma_fp4",
                    help="Hamming NN kernel: int8 MFMA over +1/-1 bytes (default) or XOR + popcount (same results)")
    ap.add_argument("--keypoint-order", choices=["canonical", "cv2"], default="canonical")
    ap.add_argument("--poly-solver", choices=["fast", "opencv300"], default="fast")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-stream-pass", action="store_true", help="skip the frames-streamed-from-host measurement")
    ap.add_argument("--no-sustain", action="store_true", help="skip the >= 5 s x 3 sustained passes")
    ap.add_argument("--sustain-seconds", type=float, default=5.0)
    ap.add_argument("--sustain-repeats", type=int, default=3)
    ap.add_argument("--workload", choices=["sequence", "independent"], default="sequence",
                    help="sequence: C+1 consecutive frames -> C pairs, each frame detected once (BASELINE config 2); "
                         "independent: C pairs with their own two frames each, 2C detections (BASELINE config 4 accounting)")
    ap.add_argument("--contexts", type=int, default=3, help="contexts (streams) per GPU alternating over the chunks")
    ap.add_argument("--chain-detect", type=int, default=0,
                    help="1: a context's detection starts after the previous context's detection (software pipeline); measured on "
                         "MI355X with the round-2 kernels: 3 contexts unchained 88.8 k pairs/s, 2 chained 85.9 k, 4-6 unchained 85-87 k")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed and run the trajectory gather even with one rank (exercises RCCL on a 1-GPU box)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearse the multi-process path on a box with fewer GPUs than ranks (all ranks share GPU 0; "
                         "records are gathered through torch.distributed instead of the library's RCCL call)")
    ap.add_argument("--gather", choices=["library", "torch"], default="library",
                    help="library: vo_pairs_gather (device-side pack + ncclAllGather on the ctx stream); torch: the records "
                         "bounce through the host into dist.all_gather_into_tensor (fallback, also taken if RCCL cannot be bound)")
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

    # Render the views BEFORE anything initialises the GPU or a process group, and in a child process: the renderer
    # forks workers.  Rank 0 fills the cache; the other ranks wait for it at init_process_group / the barrier below.
    from visual_odometry_amd import synth
    D = args.distinct_frames
    if rank == 0:
        synth.prerender(D, args.width, args.height, "/tmp", "loop")     # child process: this one never forks
    seq = None
    if use_dist:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")           # --force-dist outside torch.distributed.run
        os.environ.setdefault("MASTER_PORT", "29517")
        if args.dist_backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            device = 0
            dist.init_process_group("gloo", rank=rank, world_size=world)
    on_gpu = use_dist and args.dist_backend == "nccl"

    from visual_odometry_amd.frontend import (FrontEnd, MATCH_CROSSCHECK, MATCH_CROSSCHECK_LEGACY, MATCH_RATIO, chain_poses)
    from visual_odometry_amd.sharding import RECORD_WIDTH, pack_records

    C, S = args.pairs_per_step, max(1, args.pair_stride)
    if use_dist:
        dist.barrier()
    if seq is None:                                       # the cache rank 0 wrote (workers=1: never fork with a GPU context)
        seq = synth.sequence(D, args.width, args.height, cache_dir="/tmp", trajectory="loop", workers=1)
    K = seq["K"]
    start = (rank * 37) % D
    if args.workload == "sequence":
        order = (start + np.arange(C + S)) % D            # C + S consecutive views of the closed flight -> C pairs (k, k + S)
        frames = seq["frames"][order]
        pairs = np.stack([np.arange(C), np.arange(C) + S], axis=1).astype(np.int32)
    else:
        a = (start + np.arange(C)) % D
        frames = seq["frames"][np.stack([a, (a + S) % D], axis=1).ravel()]      # 2C frames, pair k = slots (2k, 2k+1)
        pairs = np.stack([2 * np.arange(C), 2 * np.arange(C) + 1], axis=1).astype(np.int32)
    NF = len(frames)
    match_mode = {"crosscheck": MATCH_CROSSCHECK, "ratio": MATCH_RATIO, "crosscheck-legacy": MATCH_CROSSCHECK_LEGACY}[args.matcher]

    # Several contexts (HIP streams, each with its own set of resident buffers) on the GPU: while one chunk is in its
    # latency-bound RANSAC / pose kernels the other chunks' streaming ORB kernels fill the machine.
    n_ctx = max(1, args.contexts)
    fes = []
    for c in range(n_ctx):
        fe_c = FrontEnd(args.height, args.width, max_frames=NF, max_pairs=C, nfeatures=args.nfeatures,
                        nlevels=args.nlevels, device=device, keypoint_order=args.keypoint_order)
        t_up = time.perf_counter()
        fe_c.upload(frames)                               # inputs resident in HBM before the timed region
        upload_s = time.perf_counter() - t_up
        fe_c.ctx.set_matcher_kernel(args.matcher_kernel)
        fe_c.ctx.set_poly_solver(args.poly_solver)
        fes.append(fe_c)
    fe = fes[0]
    opts = fe.make_opts(match_mode=match_mode, ratio=args.ratio, want_points=True)

    # The trajectory gather: 128 B per pair.  "library": one RCCL communicator per context, id from rank 0.
    gather_mode = None
    if use_dist:
        gather_mode = args.gather if on_gpu else "torch"
        if gather_mode == "library":
            try:
                for f in fes:
                    ident = torch.zeros(128, dtype=torch.uint8, device="cuda")
                    if rank == 0:
                        ident.copy_(torch.frombuffer(bytearray(f.ctx.comm_unique_id()), dtype=torch.uint8))
                    dist.broadcast(ident, 0)
                    f.ctx.comm_init(bytes(ident.cpu().numpy().tobytes()), rank, world)
                ok_t = torch.ones(1, device="cuda")
            except Exception as e:                        # RCCL not bindable / communicator refused: say so and fall back
                print(f"[bench rank {rank}] library gather unavailable ({e}); falling back to torch.distributed", file=sys.stderr)
                ok_t = torch.zeros(1, device="cuda")
            dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)   # all ranks take the same path
            if ok_t.item() < 1:
                gather_mode = "torch"
        if gather_mode == "torch":
            rec_t = torch.zeros((C, RECORD_WIDTH), dtype=torch.float64, pin_memory=on_gpu)
            mine_d = torch.empty((C, RECORD_WIDTH), dtype=torch.float64, device="cuda") if on_gpu else None
            out_d = torch.empty((world * C, RECORD_WIDTH), dtype=torch.float64, device="cuda" if on_gpu else "cpu")
    gathered = [None]
    in_flight = [None] * n_ctx
    counter = [0]
    iters_seen = []

    def consume(k, res):
        """Chunk k's device work has finished (f.wait()): gather its records across the ranks."""
        iters_seen.append(res["ransac_iters"].copy())
        if gather_mode == "library":
            gathered[0] = fes[k].gather_records(C, world, wait=True).reshape(world * C, RECORD_WIDTH)
        elif gather_mode == "torch":
            rec_t.numpy()[...] = pack_records(res)
            if on_gpu:
                mine_d.copy_(rec_t, non_blocking=False)
                dist.all_gather_into_tensor(out_d, mine_d)
            else:
                dist.all_gather_into_tensor(out_d, rec_t)
            gathered[0] = out_d

    staged = [None] * n_ctx                               # page-locked copies of the chunk (streamed-from-host pass)

    def step(stream_frames=False):
        """Enqueue one chunk on the next context; first retire (wait + gather) the chunk that context ran before."""
        k = counter[0] % n_ctx
        counter[0] += 1
        f = fes[k]
        if in_flight[k] is not None:
            f.wait()
            consume(k, in_flight[k])
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
                consume(k, in_flight[k])
                last = in_flight[k]
                in_flight[k] = None
        return last

    def sync():
        if use_dist:
            dist.barrier()
            if on_gpu:
                torch.cuda.synchronize()

    def timed(n_steps, stream_frames=False):
        sync()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step(stream_frames)
        res = drain()                                     # every enqueued chunk finished, results on the host
        sync()
        dt = time.perf_counter() - t0
        if use_dist:
            tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, res

    for _ in range(args.warmup):
        step()
    drain()
    iters_seen.clear()
    dt, res = timed(args.steps)                           # THE timed region: exactly --steps steps
    res = res.copy()
    iters_all = np.concatenate(iters_seen) if iters_seen else res["ransac_iters"]

    # >= sustain_seconds of back-to-back steps, `repeats` times, median; single GPU only
    sustained = None
    if not args.no_sustain and world == 1:
        n_sus = max(args.steps, int(np.ceil(args.sustain_seconds / max(dt / args.steps, 1e-6))))
        rates = []
        for _ in range(max(1, args.sustain_repeats)):
            d, _ = timed(n_sus)
            rates.append(C * n_sus / d)
        sustained = {"seconds_per_repeat": round(C * n_sus / float(np.median(rates)), 2), "steps_per_repeat": n_sus,
                     "repeats": len(rates), "pairs_per_s_median": round(float(np.median(rates)), 1),
                     "pairs_per_s_all": [round(r, 1) for r in rates]}

    # The same steps with every chunk's frames coming from (page-locked) host memory: the PCIe-inclusive rate.
    streamed = None
    if not args.no_stream_pass:
        for k in range(n_ctx):
            staged[k] = fes[k].pinned_frames(NF)
            staged[k].array[...] = frames
        for _ in range(args.warmup):
            step(True)
        drain()
        d, _ = timed(args.steps, True)
        streamed = world * C * args.steps / d

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
        hist_edges = [0, 8, 16, 32, 64, 128, 256, 512, 1001]
        hist = np.histogram(iters_all, bins=hist_edges)[0]
        line = {
            "metric": f"frame-pairs/sec ({args.width}x{args.height}, {args.nfeatures} ORB feats)", "value": round(value, 2),
            "unit": "frame-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8+f32+f64",
            "data": "synthetic",
            "value_streamed_from_host": round(streamed, 2) if streamed else None,
            "config": {"workload": f"seeded synthetic {args.width}x{args.height} drone flight ({D} distinct rendered views, closed loop), "
                                   f"{args.nfeatures} ORB features/frame, {args.nlevels} levels" +
                                   (" = BASELINE config 2; " if (args.width, args.height, args.nfeatures, args.nlevels) == (1280, 720, 2000, 8) else "; ") +
                                   (f"chunk of {NF} consecutive frames -> {C} pairs (k, k+{S}) per GPU per step, each frame detected once"
                                    if args.workload == "sequence" else
                                    f"{C} independent pairs per GPU per step, both frames of every pair detected ({NF} detections)"),
                       "pairs_per_step_per_gpu": C, "distinct_rendered_frames": D, "pair_stride": S,
                       "contexts_per_gpu": n_ctx, "keypoint_order": args.keypoint_order, "poly_solver": args.poly_solver,
                       "matcher": args.matcher, "matcher_kernel": args.matcher_kernel,
                       "ransac": "5-point, conf 0.99, 1 px, seed 2^64-1, <=1000 iters",
                       "ransac_iters": {"mean": round(float(iters_all.mean()), 1), "max": int(iters_all.max()),
                                        "histogram": {f"{hist_edges[i]}-{hist_edges[i + 1] - 1}": int(hist[i]) for i in range(len(hist))}},
                       "parallelism": (f"pair-sharded x{world}, one all-gather of 128 B/pair per step via " +
                                       ("vo_pairs_gather (device pack + ncclAllGather on the ctx stream)" if gather_mode == "library"
                                        else "torch.distributed.all_gather_into_tensor")) if use_dist else "single GPU",
                       "value_is": "HBM-resident inputs (the driver contract); value_streamed_from_host re-runs the same loop with "
                                   "every chunk's frames DMAed from page-locked host memory",
                       "sustained": sustained,
                       "unoverlapped_pageable_upload_ms_per_chunk": round(1000 * upload_s, 2),
                       "pairs_ok_last_step": ok,
                       "mean_matches_last_step": round(float(res["n_match"].mean()), 1),
                       "mean_inliers_last_step": round(float(res["n_inl"].mean()), 1)},
        }
        if prof:
            stages = {k: {"ms_per_launch": round(ms / n, 4), "launches": n, "ms_total": round(ms, 3)} for k, (ms, n) in prof.items()}
            dom = max((k for k in prof if k != "misc"), key=lambda k: prof[k][0])
            ms, n = prof[dom]
            b = fe.stage_bytes(dom, NF)
            ach = b / (ms / n * 1e-3) / 1e9 if b > 0 and ms > 0 else 0.0
            _, pmc_tag = _pmc()
            line["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                                "traffic": pmc_traffic(dom, NF),
                                "algorithmic_bytes_per_launch": b, "avg_launch_ms": round(ms / n, 4),
                                "measured": f"HIP events on the library's stream, {prof_steps} single-context steps right after the "
                                            f"timed region (the timed region overlaps {n_ctx} contexts); traffic from "
                                            f"profiles/{pmc_tag}_pmc_traffic.json"}
            hbm_stages = {}
            for k in ("pyramid_resize", "fast_score_nms", "gaussian_blur"):
                if k in prof:
                    bb = fe.stage_bytes(k, NF)
                    hbm_stages[k] = round(bb / (prof[k][0] / prof[k][1] * 1e-3) / 1e9, 1)
            insts = pmc_valu(dom)
            if insts:
                # issue bound: tools/ubench/valu_rates.hip (asm volatile) measures ~4.3 clk per wave-instruction and SIMD for
                # the packed-16 / perm / min-max / shift class this kernel is made of and ~2.5 clk for add / xor / mov / f32;
                # the weighted figure for the kernel's own opcode mix is in DESIGN.md section 6
                rate = insts * 64.0 / (ms / n * 1e-3) / 1e12
                line["roofline"]["valu"] = {
                    "wave_insts_per_launch": insts, "achieved_Tlaneops": round(rate, 2),
                    "peak_Tlaneops_if_every_instr_issued_in_2_clk": 78.6,
                    "measured_class_rates_Tlaneops": {"pk16_perm_minmax_shift_dot": 36.6, "add_xor_mov_f32": 60.0},
                    "frac_of_physical_peak": round(rate / 78.6, 3),
                    "note": "SQ_INSTS_VALU (committed PMC pass) x 64 lanes / event time"}
            line["stages"] = stages
            line["streaming_kernels_GBps"] = hbm_stages
        if not args.no_cpu_baseline and world == 1:
            n_cpu = min(D, 9)
            cpu_frames = seq["frames"][(start + S * np.arange(n_cpu)) % D]
            line["cpu_baseline"] = cpu_baseline(cpu_frames, K, args.nfeatures, args.nlevels, match_mode, args.ratio,
                                                args.width, args.height)
        if gathered[0] is not None:
            g = gathered[0].cpu().numpy() if hasattr(gathered[0], "cpu") else np.asarray(gathered[0])
            traj = chain_poses(g[:, :9].reshape(-1, 3, 3), g[:, 9:12])
            line["config"]["trajectory_poses_gathered"] = int(traj.shape[0])
            line["config"]["gathered_equals_local"] = bool(np.array_equal(g[:C], pack_records(res)))
        print(json.dumps(line), flush=True)
    if use_dist:
        for f in fes:
            if gather_mode == "library":
                f.ctx.comm_destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
