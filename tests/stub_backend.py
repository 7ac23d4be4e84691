"""A stand-in for the HIP front end, for ONE purpose: tests/test_bench_launcher.py starts `bench.py --gpus 2` on a machine
without a GPU and checks the launcher — rank processes, environment, file rendezvous, the relayed JSON line, exit codes.
bench.py loads it only when VO_BENCH_STUB names it and then labels its line "data": "STUB ..." — never a measurement.
Same method names as frontend.FrontEnd / _lib.Context / rendezvous.{init_library_comm, LibraryCollectives}."""
import os

import numpy as np

from visual_odometry_amd import _lib


class _Ctx:
    def __init__(self):
        self.rdv, self.rank, self.world, self._seq = None, 0, 1, 0

    def set_poly_solver(self, kind): pass
    def set_matcher_kernel(self, kind): pass
    def set_keypoint_order(self, kind): pass
    def comm_destroy(self): pass

    def comm_info(self):
        return self.world, self.rank

    def allgather(self, values, world):
        """all-gather of a few doubles through files of the rendezvous directory (what RCCL does on the GPUs)"""
        v = np.ascontiguousarray(values, np.float64).ravel()
        if self.rdv is None or world == 1:
            return v[None].copy()
        self._seq += 1
        base = os.path.join(self.rdv.dir, f"{self.rdv.session}_stubag_{self._seq}")
        self.rdv._write_atomic(f"{base}.{self.rank}", v.tobytes())
        out = []
        for r in range(world):
            self.rdv._wait_for(f"{base}.{r}")
            out.append(np.frombuffer(open(f"{base}.{r}", "rb").read(), np.float64))
        self.rdv._write_atomic(f"{base}.ack{self.rank}", b"")      # rank 0 removes the directory after the last barrier:
        if self.rank == 0:                                         # it leaves a collective only when everybody has read
            for r in range(world):
                self.rdv._wait_for(f"{base}.ack{r}")
        return np.stack(out)


class FrontEnd:
    def __init__(self, height, width, max_frames, max_pairs, device=0, **kw):
        if os.environ.get("VO_STUB_FAIL_RANK") == os.environ.get("RANK", "0"):
            raise SystemExit(7)                                    # the launcher test's failing rank
        if os.environ.get("VO_STUB_HANG_RANK") == os.environ.get("RANK", "0"):
            import time                                            # the launcher test's rank that never finishes
            open(os.path.join(os.environ["VO_RENDEZVOUS_DIR"], "hung_rank_pid"), "w").write(str(os.getpid()))
            time.sleep(600)
        self.h, self.w, self.max_frames, self.max_pairs, self.device = height, width, max_frames, max_pairs, device
        self.detector, self.kp_cap = kw.get("detector", "orb"), 64
        self.ctx = _Ctx()
        self._res = np.zeros(max_pairs, _lib.PAIR_RESULT_DTYPE)
        self._gath = None

    def make_opts(self, **kw):
        return None

    def upload(self, frames, first_slot=0, wait=True): pass
    def detect(self, first_slot, count, wait=True, after=None): pass
    def wait(self): pass

    def run_pairs(self, pairs, K, opts=None, want_points=False, wait=True):
        n = len(pairs)
        r = self._res
        r[:] = 0
        r["R"][:n] = np.eye(3).ravel(); r["t"][:n] = [1.0, 0.0, 0.0]
        r["n_kp1"][:n] = 100; r["n_match"][:n] = 50; r["n_inl"][:n] = 40 + self.ctx.rank; r["n_good"][:n] = 39
        r["ransac_iters"][:n] = 10
        self._n = n
        return r[:n], None

    def gather_records(self, B, world=1, wait=True):
        from visual_odometry_amd.sharding import pack_records
        rec = np.zeros((B, _lib.VO_RECORD_DOUBLES))
        rec[:self._n] = pack_records(self._res[:self._n])
        return self.ctx.allgather(rec.ravel(), world).reshape(world, B, _lib.VO_RECORD_DOUBLES)


def init_library_comm(front_ends, rdv, rank, world):
    ident = rdv.broadcast(b"\x01" * 128 if rank == 0 else b"", name="rccl_id")
    assert ident == b"\x01" * 128
    for fe in front_ends:
        fe.ctx.rdv, fe.ctx.rank, fe.ctx.world = rdv, rank, world
    for fe in front_ends[1:]:
        fe.ctx = front_ends[0].ctx                                 # one "communicator" per process


class LibraryCollectives:
    def __init__(self, ctx, world):
        self.ctx, self.world = ctx, world

    def barrier(self):
        self.ctx.allgather([0.0], self.world)

    def allreduce_max(self, v):
        return float(self.ctx.allgather([float(v)], self.world).max())
