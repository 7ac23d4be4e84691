"""CPU: the N > 1 path — pair sharding and the final record gather — with world_size 2 over gloo."""
import os
import socket

import numpy as np
import pytest


def test_shard_ranges_cover_everything():
    from visual_odometry_amd.sharding import shard_range, sequence_shard
    for n in (0, 1, 7, 8, 10000, 10001):
        for world in (1, 2, 4, 8):
            parts = [shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in parts) - min(b - a for a, b in parts) <= 1
    lo, hi, flo, fhi = sequence_shard(10001, 3, 8)      # 10k pairs on 8 ranks: 1250 pairs + 1 halo frame each
    assert (hi - lo, fhi - flo) == (1250, 1251) and flo == lo


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from visual_odometry_amd import _lib
    from visual_odometry_amd.sharding import gather_records, pack_records, shard_range
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_pairs = 11
    lo, hi = shard_range(n_pairs, rank, world)
    res = np.zeros(hi - lo, _lib.PAIR_RESULT_DTYPE)
    for i, p in enumerate(range(lo, hi)):              # stand-in per-pair results keyed by the global pair id
        res["R"][i] = np.eye(3).ravel() * (p + 1); res["t"][i] = [p, 0, 1]
        res["n_kp1"][i] = 2000 + p; res["n_match"][i] = 900 + p; res["n_inl"][i] = 500 + p; res["n_good"][i] = 499
    counts = [shard_range(n_pairs, r, world)[1] - shard_range(n_pairs, r, world)[0] for r in range(world)]
    out = gather_records(pack_records(res), counts, dist)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, out))


def test_gather_world_size_2_gloo():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(outs[0], outs[1])              # every rank holds the whole trajectory record
    assert outs[0].shape == (11, 16)
    assert np.array_equal(outs[0][:, 9], np.arange(11))  # in global pair order
    assert np.array_equal(outs[0][:, 12], 2000 + np.arange(11))


def _driver_worker(rank, world, port, q, n_items, chunk):
    """The N > 1 driver loop (sharding.run_sharded) with stand-in per-pair results: record k of the sequence encodes its
    own global pair id and the two frame ids it was computed from, so halo handling and global order are checkable."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from visual_odometry_amd import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames_touched = set()
    calls = []

    def process_chunk(lo, hi):                            # a sequence block: frames lo .. hi inclusive (hi = halo)
        calls.append((lo, hi))
        frames_touched.update(range(lo, hi + 1))
        rec = np.zeros((hi - lo, sharding.RECORD_WIDTH))
        for i, g in enumerate(range(lo, hi)):
            rec[i, :9] = np.eye(3).ravel(); rec[i, 9:12] = [1.0, 0.0, 0.0]
            rec[i, 12] = g; rec[i, 13] = g + 1; rec[i, 14] = 100 + g; rec[i, 15] = rank
        return rec

    def gather_chunk(rec):
        out = torch.empty((world * chunk, sharding.RECORD_WIDTH), dtype=torch.float64)
        dist.all_gather_into_tensor(out, torch.from_numpy(rec))
        return out.numpy().reshape(world, chunk, sharding.RECORD_WIDTH)

    allrec = sharding.run_sharded(n_items, rank, world, chunk, process_chunk, gather_chunk)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, allrec, sorted(frames_touched), calls))


@pytest.mark.parametrize("n_frames,chunk", [(24, 4), (11, 8), (3, 4)])
def test_sharded_sequence_driver_world_size_2_gloo(n_frames, chunk):
    import torch.multiprocessing as mp
    from visual_odometry_amd import sharding
    n_items = n_frames - 1
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_driver_worker, args=(r, 2, port, q, n_items, chunk)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        r, rec, touched, calls = q.get(timeout=120)
        got[r] = (rec, touched, calls)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(got[0][0], got[1][0])                           # every rank ends with the whole record table
    rec = got[0][0]
    assert rec.shape == (n_items, 16) and np.array_equal(rec[:, 12], np.arange(n_items))    # global pair order
    assert np.array_equal(rec[:, 13], np.arange(n_items) + 1)             # pair g was computed from frames (g, g + 1)
    for r in range(2):
        lo, hi, flo, fhi = sharding.sequence_shard(n_frames, r, 2)
        assert np.all(rec[lo:hi, 15] == r)                                # ... by the rank that owns it
        assert got[r][1] == list(range(flo, fhi))                         # its own frames plus exactly one halo frame
        assert all(b - a <= chunk for a, b in got[r][2])
    centres, bad = sharding.records_to_trajectory(rec)
    assert bad == 0 and np.allclose(centres[-1], [-n_items, 0, 0])       # x_{k+1} = x_k + (1,0,0): the camera moves along -x


def test_ate_alignment_removes_similarity():
    from visual_odometry_amd.sharding import ate_after_alignment
    rng = np.random.default_rng(0)
    p = np.cumsum(rng.normal(size=(50, 3)), axis=0)
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    q *= np.sign(np.linalg.det(q))
    assert ate_after_alignment(2.5 * p @ q.T + [3, -1, 2], p) < 1e-9
    assert ate_after_alignment(p + rng.normal(0, 0.1, p.shape), p) > 0.05


class _FakeFrontEnd:
    """Stands in for frontend.FrontEnd in the CPU tests of the chunk pipeline: same methods, results computed on the host from
    the 'frames' it was handed (a frame = one byte holding its global index), delivered only at wait() like the real, asynchronous one."""

    def __init__(self, max_frames, max_pairs, log):
        from visual_odometry_amd import _lib
        self.max_frames, self.max_pairs, self.log = max_frames, max_pairs, log
        self.slots = np.full(max_frames, -1, np.int64)
        self._res = np.zeros(max_pairs, _lib.PAIR_RESULT_DTYPE)
        self._pending = None
        self.detected = 0

    def make_opts(self, **kw):
        return None

    def upload(self, arr, first_slot=0, wait=True):
        assert not wait and first_slot + len(arr) <= self.max_frames
        self.slots[first_slot:first_slot + len(arr)] = np.asarray(arr).reshape(len(arr), -1)[:, 0]
        self.log.append(("upload", first_slot, len(arr)))

    def detect(self, first, count, wait=True, after=None):
        assert not wait
        self.detected += count

    def run_pairs(self, pairs, K, opts, wait=True):
        assert not wait and self._pending is None
        self._pending = np.array(pairs, np.int64).reshape(-1, 2)
        return self._res, None

    def wait(self):
        if self._pending is None:
            return
        p, self._pending = self._pending, None
        r = self._res
        r[:] = 0
        for i, (a, b) in enumerate(p):
            ga, gb = self.slots[a], self.slots[b]
            r["R"][i] = np.eye(3).ravel(); r["t"][i] = [1.0, 0.0, 0.0]
            r["n_kp1"][i] = ga; r["n_match"][i] = gb; r["n_inl"][i] = 100 + ga; r["n_good"][i] = 7


def _pipelined_worker(rank, world, port, q, n_frames, chunk, n_ctx, workload):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from visual_odometry_amd import sharding
    from visual_odometry_amd.pipeline import ChunkPipeline
    dist.init_process_group("gloo", rank=rank, world_size=world)
    D = 10                                                           # ten distinct views, walked cyclically
    ring = (np.arange(D, dtype=np.uint8)[:, None, None] * np.ones((1, 2, 3), np.uint8))
    log = []
    nf = 2 * chunk if workload == "batch" else chunk + 1
    fes = [_FakeFrontEnd(nf, chunk, log) for _ in range(n_ctx)]

    def gather(rec):                                                 # the callable form of the gather: gloo
        out = torch.empty((world * chunk, sharding.RECORD_WIDTH), dtype=torch.float64)
        dist.all_gather_into_tensor(out, torch.from_numpy(np.ascontiguousarray(rec)))
        return out.numpy().reshape(world, chunk, sharding.RECORD_WIDTH)

    pipe = ChunkPipeline(fes, np.eye(3), opts="unused", world=world, rank=rank, gather=gather, gather_rows=chunk)
    n_items = n_frames - 1 if workload == "sequence" else n_frames

    def plan(a, b):
        n = b - a
        if workload == "sequence":
            return dict(pairs=np.stack([np.arange(n), np.arange(n) + 1], 1), n_frames=n + 1, uploads=sharding.ring_uploads(ring, a, n + 1))
        return dict(pairs=np.stack([2 * np.arange(n), 2 * np.arange(n) + 1], 1), n_frames=2 * n, uploads=sharding.ring_uploads(ring, 2 * a, 2 * n))

    rec = sharding.run_sharded_pipelined(n_items, rank, world, chunk, pipe, plan)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, rec, sum(f.detected for f in fes), log))


@pytest.mark.parametrize("n,chunk,n_ctx,workload", [(24, 4, 3, "sequence"), (11, 8, 2, "sequence"), (3, 4, 1, "sequence"), (9, 2, 3, "batch")])
def test_pipelined_sharded_driver_world_size_2_gloo(n, chunk, n_ctx, workload):
    """sharding.run_sharded_pipelined + pipeline.ChunkPipeline at world size 2: contexts alternate, chunks retire out of band,
    exhausted ranks pad the collective, ring uploads wrap — and every rank still ends with every record in global order."""
    import torch.multiprocessing as mp
    from visual_odometry_amd import sharding
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pipelined_worker, args=(r, 2, port, q, n, chunk, n_ctx, workload)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        r, rec, detected, log = q.get(timeout=120)
        got[r] = (rec, detected, log)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(got[0][0], got[1][0])
    rec = got[0][0]
    n_items = n - 1 if workload == "sequence" else n
    assert rec.shape == (n_items, 16) and not np.isnan(rec).any()
    g = np.arange(n_items)
    if workload == "sequence":                                        # pair g = frames (g, g + 1) = views (g % 10, (g + 1) % 10)
        assert np.array_equal(rec[:, 12], g % 10) and np.array_equal(rec[:, 13], (g + 1) % 10)
    else:                                                             # pair p = views (2p, 2p + 1)
        assert np.array_equal(rec[:, 12], (2 * g) % 10) and np.array_equal(rec[:, 13], (2 * g + 1) % 10)
    assert np.array_equal(rec[:, 14], 100 + rec[:, 12])
    for r in range(2):
        lo, hi = sharding.shard_range(n_items, r, 2)
        per = (hi - lo) + (-(-(hi - lo) // chunk) if workload == "sequence" else (hi - lo))      # frames detected: pairs + one halo per chunk / two per pair
        assert got[r][1] == per


def test_file_rendezvous_two_processes(tmp_path):
    """rendezvous.FileRendezvous: rank 0's bytes reach the other rank, barriers meet, the directory goes away at close."""
    import subprocess
    import sys
    code = ("import os,sys; sys.path.insert(0, %r); from visual_odometry_amd.rendezvous import FileRendezvous\n"
            "r=int(sys.argv[1]); rdv=FileRendezvous(r, 2, key='t', root=sys.argv[2], timeout=30)\n"
            "rdv.barrier('a'); a=rdv.broadcast(b'x'*128 if r==0 else b'', 'id'); b=rdv.broadcast(bytes(range(7)) if r==0 else b'')\n"
            "rdv.barrier(); print(len(a), a[:1], list(b)); rdv.barrier('end')\n"
            "import time; time.sleep(0.2 if r == 0 else 0); rdv.close()\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ps = [subprocess.Popen([sys.executable, "-c", code, str(r), str(tmp_path)], stdout=subprocess.PIPE, text=True) for r in (1, 0)]
    outs = [p.communicate(timeout=60)[0].strip() for p in ps]
    assert all(p.returncode == 0 for p in ps)
    assert outs[0] == outs[1] == "128 b'x' [0, 1, 2, 3, 4, 5, 6]"
    assert not os.path.exists(os.path.join(str(tmp_path), "vo_rdv_t"))
