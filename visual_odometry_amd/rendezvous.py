"""Launcher plumbing for N > 1 ranks on ONE node without PyTorch: the 128-byte RCCL communicator id travels through a file,
everything after that (barrier, max-over-ranks of the timing, the trajectory gather itself) goes through the library's own
communicator (vo_comm_init / vo_comm_allgather_f64 / vo_pairs_gather).  torch.distributed.run may still be the process
launcher — it only sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT; nothing here imports torch.

The rendezvous directory is keyed by MASTER_PORT and the launcher's pid (all ranks of a launch are children of one agent
process), so two launches on one box do not meet."""
from __future__ import annotations

import os
import shutil
import time


class FileRendezvous:
    def __init__(self, rank, world, key=None, root=None, timeout=600.0):
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        if key is None:
            key = f"{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}_{os.getppid()}"
        root = root or os.environ.get("VO_RENDEZVOUS_DIR") or "/tmp"
        self.dir = os.path.join(root, f"vo_rdv_{key}")
        os.makedirs(self.dir, exist_ok=True)
        self._seq = 0

    def _wait_for(self, path):
        t0 = time.monotonic()
        while not os.path.exists(path):
            if time.monotonic() - t0 > self.timeout:
                raise TimeoutError(f"rank {self.rank}: nobody wrote {path} within {self.timeout:.0f} s")
            time.sleep(0.002)

    def broadcast(self, payload, name=None):
        """Rank 0's `payload` (bytes) on every rank."""
        self._seq += 1
        path = os.path.join(self.dir, f"{name or 'bcast'}_{self._seq}.bin")
        if self.rank == 0:
            tmp = path + ".tmp"
            with open(tmp, "wb") as f:
                f.write(payload)
            os.replace(tmp, path)                        # atomic: a reader never sees half a file
            return bytes(payload)
        self._wait_for(path)
        with open(path, "rb") as f:
            return f.read()

    def barrier(self, name=None):
        """Host-side barrier through marker files (used before a communicator exists, e.g. around the frame cache)."""
        self._seq += 1
        base = os.path.join(self.dir, f"{name or 'barrier'}_{self._seq}")
        with open(f"{base}.{self.rank}", "wb"):
            pass
        for r in range(self.world):
            self._wait_for(f"{base}.{r}")

    def close(self):
        """Call after a barrier of the ranks (LibraryCollectives.barrier): nobody reads the directory any more, rank 0 removes it."""
        if self.rank == 0:
            shutil.rmtree(self.dir, ignore_errors=True)


class LibraryCollectives:
    """barrier() and allreduce_max() over a context's RCCL communicator (vo_comm_allgather_f64)."""

    def __init__(self, ctx, world):
        self.ctx, self.world = ctx, int(world)

    def barrier(self):
        if self.world > 1:
            self.ctx.allgather([0.0], self.world)

    def allreduce_max(self, value):
        if self.world <= 1:
            return float(value)
        return float(self.ctx.allgather([float(value)], self.world).max())


def init_library_comm(front_ends, rdv, rank, world):
    """One RCCL communicator per context; the id of each travels from rank 0 through the file rendezvous."""
    for i, fe in enumerate(front_ends):
        ident = rdv.broadcast(fe.ctx.comm_unique_id() if rank == 0 else b"", name=f"rccl_id_{i}")
        fe.ctx.comm_init(ident, rank, world)
