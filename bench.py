#!/usr/bin/env python3
"""bench.py — frame-pairs/sec of the MI355X visual-odometry front end (BASELINE.json metric).

A step = one pass of the whole per-pair hot path (detect + describe of every frame of a chunk, matching, 5-point
E-RANSAC, recoverPose, DLT triangulation, result download) over one chunk of a seeded synthetic drone sequence (a closed
flight of --distinct-frames rendered views).  --detector orb (default) is the north-star instantiation (ORB + Hamming);
--detector sift is the configuration the reference runs live (cv2.SIFT_create() + BFMatcher(NORM_L2, crossCheck=True),
src/visual_slam.py:17,19).

Workloads:
  sequence (default) / independent   the chunk is resident in HBM before the timed region; every rank runs its own chunk
                                     each step ("scaling": "weak"), the 128 B/pair records are all-gathered each step;
  batch --pairs N                    BASELINE config 4: N independent pairs cut over the ranks (sharding.shard_range), streamed
                                     from page-locked host memory through the chunk pipeline ("scaling": "strong");
  flight --frames N                  BASELINE config 5's shape: an N-frame sequence cut with one halo frame per rank, poses
                                     chained, ATE against the ground truth and against the CPU oracle's chain.

N > 1: one process per GPU.  `python3 bench.py --gpus N` alone starts its own N rank processes (launch_ranks: the parent
never touches the GPU); torch.distributed.run may be the launcher instead: it only sets RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_PORT.  Nothing here needs PyTorch by default: the RCCL id travels through a file (rendezvous.FileRendezvous), the
barrier, the max-over-ranks of the timing and the record gather go through the library's own communicator
(vo_comm_allgather_f64 / vo_pairs_gather).  --rendezvous torch keeps the torch.distributed path (gloo: ranks sharing one GPU).

Prints ONE JSON line on rank 0 (see the driver contract).  `value` times HBM-resident inputs; the PCIe-inclusive rate of the
same loop is `value_streamed_from_host`.  Also in the line: `roofline` (HIP events on the library's stream, algorithmic bytes
from vo_stage_bytes, PMC traffic from the committed rocprofv3 pass; the bound is named per stage), `cpu_baseline` (the CPU
oracle timed on the host cores), `config.sustained`, `config.faithful` (ORB: cv2 keypoint order + cv::solvePoly's 300 sweeps,
the configuration whose keypoint / match indices equal cv2's) and the RANSAC iteration histogram.
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
VALU_PEAK_T = 78.6        # 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz, lane-operations per second (one wave-instruction per 2 clk)
PROFILE_TAG = "r04"       # profiles/<tag>_pmc_traffic[_sift].json hold this round's rocprofv3 PMC passes

# live kernels of every stage (name as rocprofv3 prints it, launches per step) and what bounds the stage
STAGE_KERNELS = {"fast_score_nms": [("k_fast<false>", 1)], "gaussian_blur": [("k_blur_direct", 1)],
                 "pyramid_resize": [("k_resize_direct", 7)],
                 "select_fast": [("k_sel_threshold", 1), ("k_sel_rows<false>", 1), ("k_sel_rows<true>", 1)],
                 "harris": [("k_harris", 1)], "ic_angle": [("k_angle", 1)], "rbrief": [("k_brief", 1), ("k_brief_trig", 1), ("k_desc_expand", 1)],
                 "match_nn": [("k_nn_fp4<false>", 1)], "essential_ransac": [("k_ransac", 1)], "recover_pose": [("k_pose", 1)],
                 "cv2_keypoint_order": [("k_cv2_order", 1), ("k_sel_rows<false>", 1), ("k_sel_rows<true>", 1)],
                 "sift_descriptor": [("k_sb_descriptor", 1)], "sift_extrema": [("k_sb_extrema<5>", 9)],
                 "sift_scale_space": [("k_sb_sweep<11, true>", 1), ("k_sb_sweep<11, false>", 8), ("k_sb_sweep<13, false>", 8), ("k_sb_sweep<17, false>", 8),
                                      ("k_sb_sweep<21, false>", 8), ("k_sb_sweep<27, false>", 8)],
                 "sift_refine_orient": [("k_sb_refine", 1), ("k_sb_orient", 1)]}
STAGE_BOUND = {"gray": "hbm", "pyramid_resize": "hbm", "fast_score_nms": "hbm", "gaussian_blur": "hbm", "sift_scale_space": "hbm",
               "sift_extrema": "hbm", "match_nn": "mfma", "essential_ransac": "latency", "recover_pose": "latency", "triangulate": "latency",
               "sift_descriptor": "valu", "sift_refine_orient": "valu", "sift_sort_unique": "valu", "rbrief": "valu", "harris": "latency",
               "ic_angle": "latency", "select_fast": "latency", "select_harris": "latency", "match_select": "latency",
               "cv2_keypoint_order": "latency", "trajectory_gather": "latency"}


_PMC_CACHE = {}


def _pmc(detector):
    """The committed rocprofv3 PMC pass of this round for `detector` — ONLY if it was collected on the sources this library
    was built from (tools/collect_traffic.py stamps the file with tools/source_hash.py's hash and vo_version()): counters of
    an older kernel are not replayed, the fields that would quote them become null."""
    if detector in _PMC_CACHE:
        return _PMC_CACHE[detector]
    n = f"{PROFILE_TAG}_pmc_traffic{'_sift' if detector == 'sift' else ''}.json"
    out = (None, None)
    try:
        from tools.source_hash import source_hash
        t = json.load(open(os.path.join(ROOT, "profiles", n)))
        if t.get("_meta", {}).get("source_hash") == source_hash():
            out = (t, n)
        else:
            print(f"bench.py: profiles/{n} was collected on other sources (hash {t.get('_meta', {}).get('source_hash')}): "
                  f"roofline.traffic / valu are reported as null; re-run tools/collect_profiles.sh", file=sys.stderr)
    except (OSError, ValueError):
        pass
    _PMC_CACHE[detector] = out
    return out


def pmc_sum(detector, stage, field, nframes=None):
    """Sum of `field` over the kernels of `stage` from the committed rocprofv3 PMC passes (per launch); None when no pass
    exists for this workload size or a kernel of the stage is missing from it."""
    t, _ = _pmc(detector)
    try:
        if t is None or (nframes is not None and t.get("_meta", {}).get("frames_per_launch") != nframes):
            return None
        per_step = field.replace("_per_launch", "_per_step")
        if all(per_step in t[k] for k, _ in STAGE_KERNELS[stage]):          # the sum over every launch of a step (octaves, levels)
            return float(sum(t[k][per_step] for k, _ in STAGE_KERNELS[stage]))
        return float(sum(t[k][field] * n for k, n in STAGE_KERNELS[stage]))
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


def oracle_pair_fn(detector, K, nfeatures, nlevels, match_mode, ratio):
    """One frame pair through the CPU oracle, in the reference's order (src/visual_slam.py:294-298)."""
    from oracle import oracle as O
    O.lib()
    if detector == "orb":
        p = O.orb_params(nfeatures=nfeatures, nlevels=nlevels)
        return lambda a, b: O.pair(a, b, p, K, match_mode=match_mode, ratio=ratio, want_points=True)

    def sift_pair(a, b):
        d1, d2 = O.sift_detect_and_compute(a), O.sift_detect_and_compute(b)
        qi, ti, _ = O.match_l2(d1["desc"], d2["desc"], 2)
        p1, p2 = d1["xy"][qi].astype(np.float64), d2["xy"][ti].astype(np.float64)
        rc, E, mask, ninl = O.find_essential_ransac(p1, p2, K)
        out = dict(rc=rc, n_match=len(qi), n_inl=ninl, R=np.eye(3), t=np.zeros((3, 1)))
        if rc == 0:
            inl = mask > 0
            ng, R, t, _ = O.recover_pose(E[0], p1[inl], p2[inl], K)
            O.triangulate(K @ np.hstack([R.T, -R.T @ t]), K @ np.eye(3, 4), p1[inl].T, p2[inl].T)
            out.update(R=R, t=t)
        return out
    return sift_pair


def cpu_baseline(detector, frames, K, nfeatures, nlevels, match_mode, ratio, width, height, budget_s=10.0):
    """The CPU oracle ("port": scalar C restatement of the cv2 path, faithful 300-sweep root finder) on the host
    cores over a bounded sample of the same pairs: first one thread, then every core the process may use."""
    from concurrent.futures import ThreadPoolExecutor
    host = os.cpu_count() or 1
    cores = usable_cores()
    fn = oracle_pair_fn(detector, K, nfeatures, nlevels, match_mode, ratio)
    n_pairs = len(frames) - 1

    def work(i):
        fn(frames[i], frames[i + 1])
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
    what = "ORB + Hamming" if detector == "orb" else "SIFT + L2"
    return {"value": round(sample / dt, 3), "unit": "frame-pairs/s", "cores": cores, "kind": "port",
            "single_thread_value": round(1.0 / one, 3), "host_cpu_count": host,
            "sample": f"{sample} pairs of the same {width}x{height} sequence via oracle/libvoo.so (scalar C restatement of the "
                      f"cv2 {what} path, cv::solvePoly's 300 sweeps), {cores} host threads, {dt:.1f} s; single thread: {n1} pairs, "
                      f"{one * n1:.1f} s; cv2 is not importable on this box, so this is NOT a cv2 timing"}


def launch_ranks(n, argv):
    """One child process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* and a per-launch rendezvous nonce in the
    environment), all of them fresh interpreters: the launcher itself never initialises HIP, and nothing is exec'ed from a
    process that has.  Rank 0's stdout is this process's stdout (the one JSON line); the other ranks' stdout goes to stderr.
    Returns 0 when every rank did; when one fails the others are ended (by their exact pids) and its code is returned."""
    import secrets
    import signal
    import socket
    import subprocess
    with socket.socket() as sk:                          # a free port: torch-style launch variables, for whoever reads them
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    key = f"bench_{os.getpid()}_{secrets.token_hex(6)}"
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VO_RENDEZVOUS_KEY=key, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0

    def forward(signum, _frame):                         # the launcher is told to stop (a driver's time limit): so are the ranks
        for p in procs:
            if p.poll() is None:
                p.send_signal(signal.SIGTERM)
        raise KeyboardInterrupt
    old_handlers = {sg: signal.signal(sg, forward) for sg in (signal.SIGTERM, signal.SIGHUP)}
    try:
        live = set(range(n))
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 128 - code
                    print(f"bench.py: rank {r} exited with {code}; ending the other ranks", file=sys.stderr, flush=True)
                    for q in live:
                        procs[q].send_signal(signal.SIGTERM)
            time.sleep(0.05)
    except KeyboardInterrupt:
        rc = 130
    finally:
        deadline = time.time() + 10
        for p in procs:
            if p.poll() is None:
                try:
                    p.wait(max(0.1, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
        for sg, h in old_handlers.items():
            signal.signal(sg, h)
    return rc


class TorchCollectives:
    def __init__(self, dist, torch, on_gpu):
        self.dist, self.torch, self.on_gpu = dist, torch, on_gpu

    def barrier(self):
        self.dist.barrier()
        if self.on_gpu:
            self.torch.cuda.synchronize()

    def allreduce_max(self, v):
        t = self.torch.tensor([float(v)], dtype=self.torch.float64, device="cuda" if self.on_gpu else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


class NoCollectives:
    def barrier(self):
        pass

    def allreduce_max(self, v):
        return float(v)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--detector", choices=["orb", "sift"], default="orb")
    ap.add_argument("--pairs-per-step", type=int, default=0, help="pairs of a chunk (default 256 for ORB, 191 for SIFT: 192 frames = one launch chain)")
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--nfeatures", type=int, default=2000)
    ap.add_argument("--nlevels", type=int, default=8)
    ap.add_argument("--kp-cap", type=int, default=0, help="SIFT: keypoints kept per frame (0: a default from the frame size)")
    ap.add_argument("--distinct-frames", type=int, default=256,
                    help="rendered views of the closed synthetic flight; a chunk walks consecutive views (wrapping)")
    ap.add_argument("--pair-stride", type=int, default=1,
                    help="pair view k with view k + stride: a wider baseline lowers the inlier ratio and multiplies the RANSAC rounds")
    ap.add_argument("--matcher", choices=["crosscheck", "ratio", "crosscheck-legacy"], default="crosscheck")
    ap.add_argument("--ratio", type=float, default=0.8)
    ap.add_argument("--matcher-kernel", choices=["mfma_fp4", "mfma", "popcount"], default="mfma_fp4",
                    help="ORB: Hamming NN kernel: block-scaled FP4 MFMA (default), int8 MFMA or XOR + popcount (same results)")
    ap.add_argument("--keypoint-order", choices=["canonical", "cv2"], default="cv2",
                    help="cv2 (default): keypoint / match indices as cv2 numbers them; canonical: (level, y, x) order, same set")
    ap.add_argument("--poly-solver", choices=["fast", "opencv300"], default="fast")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-stream-pass", action="store_true", help="skip the frames-streamed-from-host measurement")
    ap.add_argument("--no-sustain", action="store_true", help="skip the >= 5 s x 3 sustained passes")
    ap.add_argument("--no-faithful-pass", action="store_true", help="ORB: skip the cv2-order + 300-sweep pass")
    ap.add_argument("--no-extras", action="store_true", help="skip the widened rows' passes (config.sift, config.jpeg_pipeline, config.pnp, config.tracks_pnp_chain)")
    ap.add_argument("--sustain-seconds", type=float, default=5.0)
    ap.add_argument("--sustain-repeats", type=int, default=3)
    ap.add_argument("--workload", choices=["sequence", "independent", "batch", "flight"], default="sequence",
                    help="sequence: C+1 consecutive resident frames -> C pairs, each frame detected once (BASELINE config 2); "
                         "independent: C pairs with their own two frames each, 2C detections (config 4 accounting); "
                         "batch: --pairs independent pairs cut over the ranks, streamed from host (config 4); "
                         "flight: a --frames sequence cut over the ranks with one halo frame each (config 5's shape)")
    ap.add_argument("--pairs", type=int, default=10000, help="--workload batch: total pairs")
    ap.add_argument("--frames", type=int, default=4541, help="--workload flight: total frames (KITTI-00 has 4541)")
    ap.add_argument("--no-oracle-chain", action="store_true", help="--workload flight: skip the CPU oracle's chain (ATE vs ground truth only)")
    ap.add_argument("--contexts", type=int, default=3, help="contexts (streams) per GPU alternating over the chunks")
    ap.add_argument("--chain-detect", type=int, default=0, help="1: a context's detection starts after the previous context's (software pipeline)")
    ap.add_argument("--rendezvous", choices=["file", "torch"], default="file",
                    help="file: no PyTorch anywhere (id through a file, collectives through the library's RCCL communicator); "
                         "torch: torch.distributed carries the id / barrier / reduction")
    ap.add_argument("--force-dist", action="store_true", help="create the communicator and run the gather even with one rank")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="--rendezvous torch only; gloo: rehearse N ranks on a box with fewer GPUs (all ranks share GPU 0, records gathered by torch)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python3 bench.py --gpus N` without a launcher: this process becomes the launcher.  It has not touched the GPU (and
        # never will): it starts N fresh rank processes of this script and relays rank 0's JSON line.
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world
    sift = args.detector == "sift"
    C = args.pairs_per_step or (191 if sift else 256)
    S = max(1, args.pair_stride)
    device = local_rank
    use_dist = world > 1 or args.force_dist
    strong = args.workload in ("batch", "flight")

    # Render the views BEFORE anything initialises the GPU or a communicator, and in a child process: the renderer
    # forks workers.  Rank 0 fills the cache; the other ranks wait for it at the first barrier.
    from visual_odometry_amd import synth
    from visual_odometry_amd.rendezvous import FileRendezvous, LibraryCollectives, init_library_comm
    D = args.distinct_frames
    extras = (not args.no_extras and world == 1 and args.gpus == 1 and args.detector == "orb" and args.workload == "sequence" and
              not args.force_dist and (args.width, args.height) == (1280, 720))
    if rank == 0:
        synth.prerender(D, args.width, args.height, "/tmp", "loop")     # child process: this one never forks
        if extras:
            synth.prerender(D, 1152, 648, "/tmp", "loop")               # the SIFT pass runs at the reference's working size
    rdv = dist = torch = None
    on_gpu = True
    if use_dist and args.rendezvous == "file":
        rdv = FileRendezvous(rank, world)
        rdv.barrier("frames_rendered")
    elif use_dist:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if args.dist_backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            device, on_gpu = 0, False
            dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()

    from visual_odometry_amd import _lib, sharding
    from visual_odometry_amd.frontend import (FrontEnd, MATCH_CROSSCHECK, MATCH_CROSSCHECK_LEGACY, MATCH_RATIO, chain_poses)
    from visual_odometry_amd.pipeline import ChunkPipeline
    from visual_odometry_amd.sharding import RECORD_WIDTH, pack_records

    stub = os.environ.get("VO_BENCH_STUB")                  # tests only (tests/test_bench_launcher.py): a module standing in for
    if stub:                                                # the HIP front end on a box without a GPU; the line is labelled, never a measurement
        import importlib
        sb = importlib.import_module(stub)
        FrontEnd, init_library_comm, LibraryCollectives = sb.FrontEnd, sb.init_library_comm, sb.LibraryCollectives

    seq = synth.sequence(D, args.width, args.height, cache_dir="/tmp", trajectory="loop", workers=1)   # the cache rank 0 wrote
    K = seq["K"]
    start = (rank * 37) % D
    match_mode = {"crosscheck": MATCH_CROSSCHECK, "ratio": MATCH_RATIO, "crosscheck-legacy": MATCH_CROSSCHECK_LEGACY}[args.matcher]
    if args.workload in ("sequence", "flight"):
        NF = C + S
        pairs = np.stack([np.arange(C), np.arange(C) + S], axis=1).astype(np.int32)
        view_idx = (start + np.arange(NF)) % D             # C + S consecutive views of the closed flight -> C pairs (k, k + S)
    else:
        NF = 2 * C
        pairs = np.stack([2 * np.arange(C), 2 * np.arange(C) + 1], axis=1).astype(np.int32)
        a0 = (start + np.arange(C)) % D
        view_idx = np.stack([a0, (a0 + S) % D], axis=1).ravel()        # 2C frames, pair k = slots (2k, 2k+1)

    # Several contexts (HIP streams, each with its own set of resident buffers) on the GPU: while one chunk is in its
    # latency-bound RANSAC / pose kernels the other chunks' streaming detection kernels fill the machine.
    n_ctx = max(1, args.contexts)

    def make_front_ends(keypoint_order):
        out = []
        for _ in range(n_ctx):
            if sift:
                f = FrontEnd(args.height, args.width, max_frames=NF, max_pairs=C, device=device, detector="sift", kp_cap=args.kp_cap)
            else:
                f = FrontEnd(args.height, args.width, max_frames=NF, max_pairs=C, nfeatures=args.nfeatures, nlevels=args.nlevels,
                             device=device, keypoint_order=keypoint_order)
                f.ctx.set_matcher_kernel(args.matcher_kernel)
            out.append(f)
        return out

    fes = make_front_ends(args.keypoint_order)
    upload_s = 0.0
    if not strong:
        frames = seq["frames"][view_idx]
        for f in fes:
            t_up = time.perf_counter()
            f.upload(frames)                              # inputs resident in HBM before the timed region
            upload_s = time.perf_counter() - t_up
    for f in fes:
        f.ctx.set_poly_solver(args.poly_solver)
    fe = fes[0]
    opts = fe.make_opts(match_mode=match_mode, ratio=args.ratio, want_points=True)

    # The trajectory gather: 128 B per pair.  ONE RCCL communicator per process (its id from rank 0), shared by the contexts.
    gather, coll = None, NoCollectives()
    if use_dist and rdv is not None:
        init_library_comm(fes, rdv, rank, world)
        gather, coll = "library", LibraryCollectives(fe.ctx, world)
    elif use_dist:
        coll = TorchCollectives(dist, torch, on_gpu)
        if on_gpu:
            ident = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if rank == 0:
                ident.copy_(torch.frombuffer(bytearray(fe.ctx.comm_unique_id()), dtype=torch.uint8))
            dist.broadcast(ident, 0)
            fe.ctx.comm_init(bytes(ident.cpu().numpy().tobytes()), rank, world)
            for f in fes[1:]:
                f.ctx.comm_share(fe.ctx)
            gather = "library"
        else:
            out_t = torch.empty((world * C, RECORD_WIDTH), dtype=torch.float64)

            def gather(rec):                              # noqa: F811  (gloo: the records bounce through the host)
                dist.all_gather_into_tensor(out_t, torch.from_numpy(np.ascontiguousarray(rec)))
                return out_t.numpy().reshape(world, C, RECORD_WIDTH)
    pipe = ChunkPipeline(fes, K, opts, world=world, rank=rank, gather=gather, gather_rows=C, chain_detect=bool(args.chain_detect))

    iters_seen, last = [], [None]

    def note(ret):
        if ret is not None:
            iters_seen.append(ret.results["ransac_iters"][:ret.n_pairs].copy())
            last[0] = ret

    # ------------------------------------------------------------------ strong-scaling workloads: fixed total work, streamed
    if strong:
        ring = _lib.PinnedArray((D, args.height, args.width), np.uint8)       # every rendered view once, page-locked
        ring.array[...] = seq["frames"]
        n_items = args.pairs if args.workload == "batch" else max(args.frames - 1, 0)

        def plan_chunk(a, b):
            n = b - a
            if args.workload == "flight":                 # pairs (g, g + 1), g in [a, b): frames a .. b, b is the halo frame
                return dict(pairs=np.stack([np.arange(n), np.arange(n) + 1], 1), n_frames=n + 1, uploads=sharding.ring_uploads(ring.array, a, n + 1))
            return dict(pairs=np.stack([2 * np.arange(n), 2 * np.arange(n) + 1], 1), n_frames=2 * n,     # pair p = views (2p, 2p+1)
                        uploads=sharding.ring_uploads(ring.array, 2 * a, 2 * n))
        warm = sharding.run_sharded_pipelined(min(n_items, 2 * C * world), rank, world, C, pipe, plan_chunk)   # warm-up: two chunks per rank
        del warm
        coll.barrier()
        for f in fes:
            f.wait()
        t0 = time.perf_counter()
        rec = sharding.run_sharded_pipelined(n_items, rank, world, C, pipe, plan_chunk)
        for f in fes:
            f.wait()
        coll.barrier()
        dt = coll.allreduce_max(time.perf_counter() - t0)
        _, rounds = sharding.sharded_plan(n_items, world, C)
        if rank == 0:
            ok = rec[:, 14] >= 0
            line = {"metric": f"frame-pairs/sec ({args.width}x{args.height}, " + ("SIFT feats" if sift else f"{args.nfeatures} ORB feats") + ")",
                    "value": round(n_items / dt, 2), "unit": "frame-pairs/s", "n_gpus": world, "steps": rounds, "warmup": 2,
                    "ms_per_step": round(1000 * dt / max(rounds, 1), 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                    "dtype": "u8+f32+f64", "data": "synthetic",
                    "config": {"workload": (f"BASELINE config 4: {n_items} independent {args.width}x{args.height} pairs (pair p = views 2p, 2p+1 of the closed "
                                            f"flight, both frames detected), " if args.workload == "batch" else
                                            f"BASELINE config 5's shape: a {args.frames}-frame {args.width}x{args.height} sequence (KITTI-00 has 4541 frames of 1241x376; "
                                            f"the dataset is not in the image, this is the synthetic closed flight), one halo frame per rank, ") +
                                           f"cut over {world} rank(s) in chunks of {C} pairs, frames DMAed from page-locked host memory",
                               "value_is": "PCIe-inclusive: every chunk's frames come from host memory (fixed total work: nothing to keep resident)",
                               "detector": args.detector, "pairs": n_items, "pairs_per_chunk": C, "contexts_per_gpu": n_ctx, "seconds": round(dt, 3),
                               "failed_pairs": int((~ok).sum()), "mean_inliers": round(float(rec[ok, 14].mean()), 1) if ok.any() else None,
                               "parallelism": f"pair-sharded x{world}, chunk pipeline, one all-gather of 128 B/pair per chunk via vo_pairs_gather" if use_dist else "single GPU, chunk pipeline"}}
            if args.workload == "flight":
                centres, bad = sharding.records_to_trajectory(rec)
                gt_rec = np.zeros((n_items, RECORD_WIDTH))
                for g in range(min(n_items, D)):
                    R, t = synth.relative_pose(seq["R"][g % D], seq["C"][g % D], seq["R"][(g + 1) % D], seq["C"][(g + 1) % D])
                    gt_rec[g, :9] = R.ravel(); gt_rec[g, 9:12] = t
                for g in range(D, n_items):
                    gt_rec[g] = gt_rec[g % D]
                gt_centres, _ = sharding.records_to_trajectory(gt_rec)
                line["config"]["ate_vs_ground_truth"] = round(sharding.ate_after_alignment(centres, gt_centres), 5)
                line["config"]["trajectory_extent_unit_steps"] = round(float(np.linalg.norm(gt_centres.max(0) - gt_centres.min(0))), 1)
                if not args.no_oracle_chain and not args.no_cpu_baseline:
                    # the reference CPU run's chain: the flight has D distinct pairs (g, g + 1), the oracle runs each once
                    from concurrent.futures import ThreadPoolExecutor
                    fn = oracle_pair_fn(args.detector, K, args.nfeatures, args.nlevels, match_mode, args.ratio)
                    t_or = time.perf_counter()
                    with ThreadPoolExecutor(usable_cores()) as ex:
                        outs = list(ex.map(lambda g: fn(seq["frames"][g % D], seq["frames"][(g + 1) % D]), range(min(n_items, D))))
                    o_rec = np.zeros((n_items, RECORD_WIDTH))
                    for g, o in enumerate(outs):
                        o_rec[g, :9] = np.asarray(o["R"]).ravel(); o_rec[g, 9:12] = np.asarray(o["t"]).ravel(); o_rec[g, 14] = o["n_inl"] if o["rc"] == 0 else -1
                    for g in range(D, n_items):
                        o_rec[g] = o_rec[g % D]
                    o_centres, _ = sharding.records_to_trajectory(o_rec)
                    dRt = np.abs(rec[:, :12] - o_rec[:, :12]).max(axis=1)
                    line["config"]["ate_vs_cpu_oracle_chain"] = round(sharding.ate_after_alignment(centres, o_centres), 8)
                    line["config"]["max_abs_diff_R_t_vs_cpu_oracle"] = float(dRt.max())
                    line["config"]["inlier_counts_equal_cpu_oracle"] = bool(np.array_equal(rec[:, 14], o_rec[:, 14]))
                    line["config"]["cpu_oracle_chain_seconds"] = round(time.perf_counter() - t_or, 1)
            print(json.dumps(line), flush=True)
        _shutdown(fes, gather, coll, rdv, dist)
        return

    # ------------------------------------------------------------------ the default workloads: resident chunk, K timed steps
    staged = None

    def step(stream=False):
        ups = [(staged.array, 0)] if stream else None
        note(pipe.submit(pairs, NF, uploads=ups))

    def drain():
        for r in pipe.drain():
            note(r)

    def timed(n_steps, stream=False):
        coll.barrier()
        for f in fes:
            f.wait()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step(stream)
        drain()                                           # every enqueued chunk finished, results on the host
        coll.barrier()
        return coll.allreduce_max(time.perf_counter() - t0)

    for _ in range(args.warmup):
        step()
    drain()
    iters_seen.clear()
    dt = timed(args.steps)                                # THE timed region: exactly --steps steps
    res = last[0].results[:C].copy()
    gathered = last[0].gathered.copy() if last[0].gathered is not None else None
    iters_all = np.concatenate(iters_seen) if iters_seen else res["ransac_iters"]

    # >= sustain_seconds of back-to-back steps, `repeats` times, median; single GPU only
    sustained = None
    if not args.no_sustain and world == 1:
        n_sus = max(args.steps, int(np.ceil(args.sustain_seconds / max(dt / args.steps, 1e-6))))
        rates = [C * n_sus / timed(n_sus) for _ in range(max(1, args.sustain_repeats))]
        sustained = {"seconds_per_repeat": round(C * n_sus / float(np.median(rates)), 2), "steps_per_repeat": n_sus,
                     "repeats": len(rates), "pairs_per_s_median": round(float(np.median(rates)), 1),
                     "pairs_per_s_all": [round(r, 1) for r in rates]}

    # The same steps with every chunk's frames coming from (page-locked) host memory: the PCIe-inclusive rate.
    streamed = None
    if not args.no_stream_pass:
        staged = fe.pinned_frames(NF)
        staged.array[...] = frames
        for _ in range(args.warmup):
            step(True)
        drain()
        streamed = world * C * args.steps / timed(args.steps, True)

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

    # ORB: the fully faithful configuration — cv2's keypoint order (the default) AND cv::solvePoly's fixed 300 Durand-Kerner sweeps
    # (the default solver stops at the rounding-noise floor: identical masks, [R|t] within 1e-4 of the 300-sweep result).  Same
    # loop, its own contexts, a short pass.
    faithful = None
    if not sift and not args.no_faithful_pass and world == 1 and (args.keypoint_order, args.poly_solver) != ("cv2", "opencv300"):
        fes2 = make_front_ends("cv2")                    # their own contexts (the first set stays alive: stage_bytes, communicators)
        for f in fes2:
            f.upload(frames)
            f.ctx.set_poly_solver("opencv300")
        pipe2 = ChunkPipeline(fes2, K, opts, chain_detect=bool(args.chain_detect))
        n_f = max(4, args.steps // 2)
        for _ in range(2):
            pipe2.submit(pairs, NF)
        pipe2.drain()
        t0 = time.perf_counter()
        for _ in range(n_f):
            pipe2.submit(pairs, NF)
        r2 = pipe2.drain()
        d2 = time.perf_counter() - t0
        faithful = {"pairs_per_s": round(C * n_f / d2, 1), "steps": n_f, "keypoint_order": "cv2", "poly_solver": "opencv300",
                    "pairs_ok_last_step": int((r2[-1].results["status"][:C] == 0).sum()),
                    "mean_inliers_last_step": round(float(r2[-1].results["n_inl"][:C].mean()), 1),
                    "what": "cv2's KeyPointsFilter::retainBest permutation replayed on the device + cv::solvePoly's fixed 300 Durand-Kerner "
                            "sweeps: keypoint / match indices, masks, E, R|t bit-identical to the oracle (tests/test_gpu_cv2_order.py, "
                            "tests/test_gpu_faithful.py); the headline value differs only in the root finder's exit rule"}
        del pipe2, fes2

    # The rows either side of the pair path, under the same clock (tools/bench_passes.py): the reference's live SIFT + L2
    # configuration at its working size, files in -> poses out, and solvePnPRansac.  Each runs on contexts of its own after the
    # timed region, like config.faithful.
    extra = {}
    if extras and not stub:
        from tools import bench_passes as BP
        try:
            seq_s = synth.sequence(D, 1152, 648, cache_dir="/tmp", trajectory="loop", workers=1)
            extra["sift"] = BP.sift_pass(seq_s["frames"][(start + np.arange(192)) % D], seq_s["K"], device=device, hbm_peak_gbs=HBM_PEAK_GBS)
            files = BP.jpeg_files(seq["frames"][:64])
            extra["jpeg_pipeline"] = BP.jpeg_pipeline_pass(files, args.width, args.height, K, nfeatures=args.nfeatures, device=device)
            extra["pnp"] = BP.pnp_pass(_lib.Context(device))[0]
            extra["tracks_pnp_chain"] = BP.chain_pass(seq["frames"][:65], K, nfeatures=args.nfeatures, device=device)
        except Exception as e:                            # noqa: BLE001 — an extra never takes the headline line down
            extra["error"] = f"{type(e).__name__}: {e}"

    ok = int((res["status"] == 0).sum())
    if rank == 0:
        value = world * C * args.steps / dt
        hist_edges = [0, 8, 16, 32, 64, 128, 256, 512, 1001]
        hist = np.histogram(iters_all, bins=hist_edges)[0]
        feat = "SIFT feats" if sift else f"{args.nfeatures} ORB feats"
        line = {
            "metric": f"frame-pairs/sec ({args.width}x{args.height}, {feat})", "value": round(value, 2),
            "unit": "frame-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8+f32+f64",
            "data": "synthetic" if not stub else f"STUB front end {stub}: launcher self-test, NOT a measurement",
            "value_streamed_from_host": round(streamed, 2) if streamed else None,
            "config": {"workload": f"seeded synthetic {args.width}x{args.height} drone flight ({D} distinct rendered views, closed loop), " +
                                   (f"cv2.SIFT_create() defaults (every keypoint kept, ~{int(res['n_kp1'].mean())} per frame) + BFMatcher(NORM_L2, crossCheck) = the "
                                    f"reference's live configuration (src/visual_slam.py:17,19); " if sift else
                                    f"{args.nfeatures} ORB features/frame, {args.nlevels} levels" +
                                    (" = BASELINE config 2; " if (args.width, args.height, args.nfeatures, args.nlevels) == (1280, 720, 2000, 8) else "; ")) +
                                   (f"chunk of {NF} consecutive frames -> {C} pairs (k, k+{S}) per GPU per step, each frame detected once"
                                    if args.workload == "sequence" else
                                    f"{C} independent pairs per GPU per step, both frames of every pair detected ({NF} detections)"),
                       "detector": args.detector, "pairs_per_step_per_gpu": C, "distinct_rendered_frames": D, "pair_stride": S,
                       "contexts_per_gpu": n_ctx, "keypoint_order": args.keypoint_order if not sift else "cv2 (KeyPoint_LessThan sort + removeDuplicatedSorted)",
                       "poly_solver": args.poly_solver, "matcher": args.matcher,
                       "matcher_kernel": "int8 MFMA on (v - 128), exact integer d^2" if sift else args.matcher_kernel,
                       "ransac": "5-point, conf 0.99, 1 px, seed 2^64-1, <=1000 iters",
                       "ransac_iters": {"mean": round(float(iters_all.mean()), 1), "max": int(iters_all.max()),
                                        "histogram": {f"{hist_edges[i]}-{hist_edges[i + 1] - 1}": int(hist[i]) for i in range(len(hist))}},
                       "n_ranks_in_communicator": fe.ctx.comm_info()[0] if gather == "library" else None,
                       "parallelism": (f"pair-sharded x{world}, one all-gather of 128 B/pair per step via " +
                                       ("vo_pairs_gather (device pack + ncclAllGather over the process's one communicator, collectives chained in submit order); rendezvous: " +
                                        ("a file, no PyTorch in the process" if rdv is not None else "torch.distributed")
                                        if gather == "library" else "torch.distributed.all_gather_into_tensor (gloo)")) if use_dist else "single GPU",
                       "value_is": "HBM-resident inputs (the driver contract); value_streamed_from_host re-runs the same loop with "
                                   "every chunk's frames DMAed from page-locked host memory",
                       "sustained": sustained, "faithful": faithful, **extra,
                       "unoverlapped_pageable_upload_ms_per_chunk": round(1000 * upload_s, 2),
                       "pairs_ok_last_step": ok,
                       "mean_keypoints_last_step": round(float(res["n_kp1"].mean()), 1),
                       "mean_matches_last_step": round(float(res["n_match"].mean()), 1),
                       "mean_inliers_last_step": round(float(res["n_inl"].mean()), 1)},
        }
        if prof:
            stages = {k: {"ms_per_launch": round(ms / n, 4), "launches": n, "ms_total": round(ms, 3), "bound": STAGE_BOUND.get(k, "latency")}
                      for k, (ms, n) in prof.items()}
            per_step = {k: ms / prof_steps for k, (ms, n) in prof.items()}          # a stage may be several launches per step (SIFT octaves)

            def hbm_entry(k):
                b = fe.stage_bytes(k, NF)
                ms = per_step[k]
                ach = b / (ms * 1e-3) / 1e9 if b > 0 and ms > 0 else 0.0
                _, pmc_file = _pmc(args.detector)
                return {"kernel": k, "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                        "traffic": pmc_sum(args.detector, k, "hbm_bytes_per_launch", NF), "algorithmic_bytes_per_launch": b,
                        "avg_launch_ms": round(ms, 4),
                        "measured": f"HIP events on the library's stream, {prof_steps} single-context steps right after the timed region "
                                    f"(the timed region overlaps {n_ctx} contexts); all launches of the stage in one step count as one "
                                    f"launch; traffic from profiles/{pmc_file}"}
            # algorithmic bytes of one frame pair (its own accounting, from the level sizes) x pairs/s / peak
            Ppx = fe.stage_bytes("fast_score_nms", 1) if not sift else 0.0
            e2e = None
            if Ppx:
                N = float(args.nfeatures)
                frame_b = fe.stage_bytes("pyramid_resize", 1) + 3.0 * Ppx + (2 * N * 49 + N * 749 + N * 961) + N * 60
                pair_b = (frame_b if args.workload == "sequence" else 2 * frame_b) + (2 * N * 32 + 16 * N) + 65 * N
                e2e = {"algorithmic_bytes_per_pair": int(pair_b), "mode": "streaming sequence (frame + matcher + geometry)" if args.workload == "sequence" else "independent pairs (2 x frame + matcher + geometry)",
                       "achieved": round(pair_b * value / 1e9, 1), "unit": "GB/s", "frac": round(pair_b * value / (HBM_PEAK_GBS * 1e9), 4)}
            dom = max((k for k in per_step if k != "misc"), key=lambda k: per_step[k])
            hbm_dom = max((k for k in per_step if STAGE_BOUND.get(k) == "hbm"), key=lambda k: per_step[k], default=None)
            if STAGE_BOUND.get(dom) == "hbm":
                line["roofline"] = hbm_entry(dom)
            else:
                # the slowest stage of this run is not a streaming kernel: its HBM fraction would be meaningless.  Named with what
                # bounds it; the slowest HBM-bound stage follows as `roofline_hbm`
                line["roofline"] = {"kernel": dom, "bound": STAGE_BOUND.get(dom, "latency"), "achieved": None, "peak": None, "unit": None, "frac": None,
                                    "traffic": None, "avg_launch_ms": round(per_step[dom], 4),
                                    "note": "the dominant stage is bound by dependent-instruction latency / vector issue, not by HBM: see roofline_hbm "
                                            "for the dominant streaming kernel and DESIGN.md section 6"}
                if hbm_dom:
                    line["roofline_hbm"] = hbm_entry(hbm_dom)
            if e2e:
                line["roofline"]["achieved_end_to_end"] = e2e
            insts = pmc_sum(args.detector, dom, "valu_wave_insts_per_launch", NF)
            if insts:
                # issue bound: tools/ubench/valu_rates.hip (asm volatile) measures ~4.3 clk per wave-instruction and SIMD for the
                # packed-16 / perm / min-max / shift / 3-operand class and ~2.5 clk for add / xor / mov / f32; `mix_weighted` prices
                # the kernel's own opcode mix at those rates (DESIGN.md section 6 for k_fast: 343 M half-rate-class + 58 M full-rate)
                rate = insts * 64.0 / (per_step[dom] * 1e-3) / 1e12
                valu = {"wave_insts_per_launch": insts, "achieved_Tlaneops": round(rate, 2),
                        "peak_Tlaneops_if_every_instr_issued_in_2_clk": VALU_PEAK_T,
                        "measured_class_rates_Tlaneops": {"pk16_perm_minmax_shift_dot_3op": 36.6, "add_xor_mov_f32": 60.0},
                        "frac_of_physical_peak": round(rate / VALU_PEAK_T, 3),
                        "note": "SQ_INSTS_VALU (committed PMC pass) x 64 lanes / event time"}
                mix = (_pmc(args.detector)[0] or {}).get("_meta", {}).get("isa_mix", {}).get(STAGE_KERNELS[dom][0][0]) if dom in STAGE_KERNELS else None
                if mix:                                   # the kernel's own opcode classes (tools/isa_mix.py --json on the build's ISA, stored with the counters)
                    hr, fr = mix["half_rate_class_frac"], mix["full_rate_class_frac"]
                    mix_ms = (hr * insts * 64 / 36.6e12 + fr * insts * 64 / 60.0e12) * 1e3
                    valu["isa_class_mix"] = {"pk16_perm_minmax_shift_dot_3op": hr, "add_xor_mov_f32": fr, "source": "tools/isa_mix.py (static, per opcode of the kernel body)"}
                    valu["mix_weighted_issue_bound_ms"] = round(mix_ms, 4)
                    valu["frac_of_mix_weighted_bound"] = round(mix_ms / per_step[dom], 3)
                line["roofline"]["valu"] = valu
            line["stages"] = stages
            line["streaming_kernels_GBps"] = {k: round(fe.stage_bytes(k, NF) / (per_step[k] * 1e-3) / 1e9, 1)
                                              for k in per_step if STAGE_BOUND.get(k) == "hbm" and fe.stage_bytes(k, NF) > 0}
        if not args.no_cpu_baseline and world == 1:
            n_cpu = min(D, 9)
            cpu_frames = seq["frames"][(start + S * np.arange(n_cpu)) % D]
            line["cpu_baseline"] = cpu_baseline(args.detector, cpu_frames, K, args.nfeatures, args.nlevels, match_mode, args.ratio,
                                                args.width, args.height)
        if gathered is not None:
            g = np.asarray(gathered).reshape(-1, RECORD_WIDTH)
            traj = chain_poses(g[:, :9].reshape(-1, 3, 3), g[:, 9:12])
            line["config"]["trajectory_poses_gathered"] = int(traj.shape[0])
            line["config"]["gathered_equals_local"] = bool(np.array_equal(g[rank * C:(rank + 1) * C], pack_records(res)))
        print(json.dumps(line), flush=True)
    _shutdown(fes, gather, coll, rdv, dist)


def _shutdown(fes, gather, coll, rdv, dist):
    coll.barrier()
    if gather == "library":
        for f in fes:
            f.ctx.comm_destroy()
    if rdv is not None:
        rdv.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
