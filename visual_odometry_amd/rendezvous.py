"""Launcher plumbing for N > 1 ranks on ONE node without PyTorch: the 128-byte RCCL communicator id travels through a file,
everything after that (barrier, max-over-ranks of the timing, the trajectory gather itself) goes through the library's own
communicator (vo_comm_init / vo_comm_allgather_f64 / vo_pairs_gather).  torch.distributed.run may still be the process
launcher — it only sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT; nothing here imports torch.

The rendezvous directory is keyed by VO_RENDEZVOUS_KEY (a per-launch nonce: bench.py's own launcher sets one) or, without
it, by MASTER_PORT and the launcher's pid.  Whatever the key, a launch never trusts what an earlier one left behind:
rank 0 empties the directory and opens a *session* (a fresh nonce + its own pid and start time); the other ranks adopt a
session only while the process that opened it is alive, and every file of the launch carries the session's nonce."""
from __future__ import annotations

import os
import secrets
import shutil
import time


def _proc_start(pid):
    """Start time (clock ticks since boot) of a live process, None when it does not exist."""
    try:
        with open(f"/proc/{int(pid)}/stat", "rb") as f:
            return int(f.read().rsplit(b")", 1)[1].split()[19])
    except (OSError, ValueError, IndexError):
        return None


class FileRendezvous:
    def __init__(self, rank, world, key=None, root=None, timeout=600.0):
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        if key is None:
            key = os.environ.get("VO_RENDEZVOUS_KEY") or \
                f"{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}_{os.getppid()}"
        root = root or os.environ.get("VO_RENDEZVOUS_DIR") or "/tmp"
        self.dir = os.path.join(root, f"vo_rdv_{key}")
        os.makedirs(self.dir, mode=0o700, exist_ok=True)
        st = os.stat(self.dir)
        if st.st_uid != os.getuid():
            raise PermissionError(f"rendezvous directory {self.dir} belongs to uid {st.st_uid}, not to this user")
        self._seq = 0
        self.session = self._open_session() if self.rank == 0 else self._join_session()

    # ------------------------------------------------------------------ the session handshake
    def _session_path(self):
        return os.path.join(self.dir, "session")

    def _open_session(self):
        for name in os.listdir(self.dir):                 # whatever a crashed launch with the same key left behind
            p = os.path.join(self.dir, name)
            shutil.rmtree(p, ignore_errors=True) if os.path.isdir(p) else _unlink(p)
        nonce = secrets.token_hex(8)
        self._write_atomic(self._session_path(), f"{nonce} {os.getpid()} {_proc_start(os.getpid())}".encode())
        for r in range(1, self.world):
            self._wait_for(os.path.join(self.dir, f"{nonce}_hello.{r}"))
        self._write_atomic(os.path.join(self.dir, f"{nonce}_go"), b"")
        return nonce

    def _read_session(self):
        """(nonce) of a session whose rank 0 is alive, else None."""
        try:
            with open(self._session_path(), "rb") as f:
                nonce, pid, start = f.read().decode().split()
        except (OSError, ValueError):
            return None
        return nonce if str(_proc_start(pid)) == start else None

    def _join_session(self):
        t0 = time.monotonic()
        nonce = None
        while True:
            cur = self._read_session()
            if cur is not None and cur != nonce:          # a (new) live session: say hello to it
                nonce = cur
                self._write_atomic(os.path.join(self.dir, f"{nonce}_hello.{self.rank}"), b"")
            if nonce is not None and cur == nonce and os.path.exists(os.path.join(self.dir, f"{nonce}_go")):
                return nonce
            if time.monotonic() - t0 > self.timeout:
                raise TimeoutError(f"rank {self.rank}: no live rank 0 opened a session in {self.dir} within {self.timeout:.0f} s "
                                   f"(stale directory of an earlier launch, or rank 0 never started)")
            time.sleep(0.002)

    # ------------------------------------------------------------------ files of the session
    @staticmethod
    def _write_atomic(path, payload):
        tmp = f"{path}.tmp{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(payload)
        os.replace(tmp, path)                             # atomic: a reader never sees half a file

    def _wait_for(self, path):
        t0 = time.monotonic()
        while not os.path.exists(path):
            if time.monotonic() - t0 > self.timeout:
                raise TimeoutError(f"rank {self.rank}: nobody wrote {path} within {self.timeout:.0f} s")
            time.sleep(0.002)

    def broadcast(self, payload, name=None):
        """Rank 0's `payload` (bytes) on every rank."""
        self._seq += 1
        path = os.path.join(self.dir, f"{self.session}_{name or 'bcast'}_{self._seq}.bin")
        if self.rank == 0:
            self._write_atomic(path, payload)
            return bytes(payload)
        self._wait_for(path)
        with open(path, "rb") as f:
            return f.read()

    def barrier(self, name=None):
        """Host-side barrier through marker files (used before a communicator exists, e.g. around the frame cache)."""
        self._seq += 1
        base = os.path.join(self.dir, f"{self.session}_{name or 'barrier'}_{self._seq}")
        self._write_atomic(f"{base}.{self.rank}", b"")
        for r in range(self.world):
            self._wait_for(f"{base}.{r}")

    def close(self):
        """Call after a barrier of the ranks (LibraryCollectives.barrier): nobody reads the directory any more, rank 0 removes it."""
        if self.rank == 0:
            shutil.rmtree(self.dir, ignore_errors=True)


def _unlink(path):
    try:
        os.unlink(path)
    except OSError:
        pass


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
    """ONE RCCL communicator per process: the first context creates it (its id travels from rank 0 through the file
    rendezvous), the other contexts of the GPU share it (vo_comm_share).  Every all-gather waits (event) for the one
    submitted before it — the same order on every rank — so no two collectives of a process are ever in flight at once."""
    first = front_ends[0].ctx
    ident = rdv.broadcast(first.comm_unique_id() if rank == 0 else b"", name="rccl_id")
    first.comm_init(ident, rank, world)
    for fe in front_ends[1:]:
        fe.ctx.comm_share(first)
