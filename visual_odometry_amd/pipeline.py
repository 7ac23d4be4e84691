"""The chunk pipeline of one GPU: several contexts (HIP streams, each with its own resident buffers) alternate over chunks
of frames, so that one chunk's DMA upload and latency-bound geometry kernels run beside another chunk's streaming detection
kernels.  bench.py, examples/sharded_run.py and sharding.run_sharded_pipelined all drive this one loop.

A chunk = frames (already resident in the context's slots, or slices of page-locked host memory that are DMAed into them) +
the pairs to run on them.  submit() only enqueues: upload -> detect -> match / E-RANSAC / pose / DLT -> (optionally) the
all-gather of the chunk's 128-byte records over the context's RCCL communicator.  A chunk is retired — waited for, its
results handed to the caller — when its context is needed again, or by drain()."""
from __future__ import annotations

import numpy as np

from . import _lib
from .sharding import RECORD_WIDTH, pack_records


class Retired:
    """Results of a finished chunk.  `results` (structured array) and `gathered` ([world, C, 16]) are views of the context's
    reused page-locked buffers: copy what must outlive the next chunk on that context."""
    __slots__ = ("tag", "results", "points", "gathered", "n_pairs")

    def __init__(self, tag, results, points, gathered, n_pairs):
        self.tag, self.results, self.points, self.gathered, self.n_pairs = tag, results, points, gathered, n_pairs

    def records(self):
        """[n_pairs, 16] float64 of this rank's chunk (from the gathered block when there is one)."""
        return pack_records(self.results[:self.n_pairs])


class ChunkPipeline:
    def __init__(self, front_ends, K, opts=None, world=1, rank=0, gather=None, gather_rows=None, chain_detect=False):
        """front_ends: FrontEnd objects of ONE GPU (one context each).  gather: None, "library" (vo_pairs_gather: device pack +
        ncclAllGather on the context stream; every context needs ctx.comm_init) or a callable rec[C, 16] -> [world, C, 16] run at
        retirement (the gloo / torch fall-back and the CPU tests).  gather_rows: rows every rank contributes per chunk (the
        collective needs one size: short chunks are padded)."""
        self.fes = list(front_ends)
        self.K = np.ascontiguousarray(K, np.float64).reshape(3, 3)
        self.opts = opts or self.fes[0].make_opts(want_points=False)
        self.world, self.rank = int(world), int(rank)
        self.gather = gather
        self.gather_rows = int(gather_rows or self.fes[0].max_pairs)
        self.chain_detect = bool(chain_detect)
        self._n = 0
        self._inflight = [None] * len(self.fes)

    # ------------------------------------------------------------------ enqueue
    def submit(self, pairs, n_frames, uploads=None, tag=None, detect=True):
        """pairs: [B, 2] slot indices (B may be 0: an exhausted rank still takes part in the collective).  uploads: a list of
        (page-locked uint8 array [n, H, W], first_slot) DMAed into the slots before the detection; None: the frames are
        resident already.  Returns the Retired chunk this context ran before, or None."""
        k = self._n % len(self.fes)
        self._n += 1
        fe = self.fes[k]
        old = self._retire(k)
        for arr, first in (uploads or ()):
            fe.upload(arr, first_slot=first, wait=False)
        if detect and n_frames > 0:
            fe.detect(0, n_frames, wait=False, after=self.fes[(k - 1) % len(self.fes)] if self.chain_detect else None)
        pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        res, X = fe.run_pairs(pairs, self.K, self.opts, wait=False)
        g = None
        if self.gather == "library":
            g = fe.gather_records(self.gather_rows, self.world, wait=False)
        self._inflight[k] = (tag, res, X, g, len(pairs))
        return old

    def _retire(self, k):
        cur = self._inflight[k]
        if cur is None:
            return None
        self._inflight[k] = None
        tag, res, X, g, n = cur
        self.fes[k].wait()
        if callable(self.gather):
            rec = np.zeros((self.gather_rows, RECORD_WIDTH), np.float64)
            rec[:, 14] = _lib.VO_ERR_NOT_CONFIGURED            # padding rows, as the library marks them
            rec[:n] = pack_records(res[:n])
            g = np.asarray(self.gather(rec)).reshape(self.world, self.gather_rows, RECORD_WIDTH)
        return Retired(tag, res, X, g, n)

    def drain(self):
        """Retire everything in flight, oldest first."""
        out = []
        for j in range(len(self.fes)):
            k = (self._n + j) % len(self.fes)
            r = self._retire(k)
            if r is not None:
                out.append(r)
        return out
