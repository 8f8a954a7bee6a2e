"""Multi-GPU host logic: one process per GPU, utterances sharded, bank broadcast once.

The data path has no collective (utterances are independent, get_train_data.py:1269-1321);
the only exchange is the RCCL broadcast of the bank (tsr_bank_broadcast), which needs the
128-byte RCCL unique id handed from rank 0 to the others over some host channel.  That channel
is `HostGroup` below: a star of TCP connections on the loopback interface built from the
standard library only (no torch, no MPI) -- rendezvous, barrier, all-gather of small Python
objects, max-over-ranks of the timings.  A launcher such as `python -m torch.distributed.run`
merely starts the processes and sets RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.

Every helper that takes a `group` accepts a HostGroup or anything with torch.distributed's
`all_gather_object/get_rank/get_world_size` (the module itself works), so the same logic is
covered by the gloo tests."""
import os
import pickle
import socket
import struct
import tempfile
import time

import numpy as np

_MAGIC = b"TSRCTRL1"


def shard_utterances(lengths, world_size):
    """Contiguous shards balanced by cumulative frame count (greedy prefix split).
    Returns bounds i64 [world_size+1]: rank r owns utterances [bounds[r], bounds[r+1])."""
    lengths = np.asarray(lengths, np.int64)
    cum = np.concatenate(([0], np.cumsum(lengths)))
    total = int(cum[-1])
    bounds = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        i = int(np.searchsorted(cum, target, side="left"))
        if i > 0 and abs(cum[i - 1] - target) <= abs(cum[min(i, len(lengths))] - target):
            i -= 1
        bounds.append(min(max(i, bounds[-1]), len(lengths)))
    bounds.append(len(lengths))
    return np.asarray(bounds, np.int64)


# ---- host control plane ------------------------------------------------------------------
def _send(sock, obj):
    data = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    sock.sendall(struct.pack("<Q", len(data)) + data)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(n - len(buf), 1 << 20))
        if not chunk:
            raise ConnectionError("host control plane: peer closed the connection")
        buf += chunk
    return bytes(buf)


def _recv(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    return pickle.loads(_recv_exact(sock, n))


class HostGroup:
    """Rank 0 listens on (addr, port); ranks 1..world-1 connect.  Collectives are an all-gather of
    pickled objects through rank 0 -- a few hundred bytes per call, used a handful of times per
    run, never inside the data path."""

    def __init__(self, rank, world_size, addr="127.0.0.1", port=0, rendezvous_file=None, timeout=120.0):
        self.rank, self.world_size = int(rank), int(world_size)
        self._peers, self._sock, self._file = [], None, None
        if self.world_size == 1:
            return
        deadline = time.time() + timeout
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, int(port)))
            srv.listen(self.world_size)
            srv.settimeout(timeout)
            self.port = srv.getsockname()[1]
            if rendezvous_file:  # publish the port (atomic rename) for ranks that cannot know it
                tmp = "%s.%d" % (rendezvous_file, os.getpid())
                with open(tmp, "w") as f:
                    f.write("%d\n" % self.port)
                os.replace(tmp, rendezvous_file)
                self._file = rendezvous_file
            peers = {}
            while len(peers) < self.world_size - 1:
                conn, _ = srv.accept()
                conn.settimeout(timeout)
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                hello = _recv_exact(conn, len(_MAGIC) + 8)
                r, w = struct.unpack("<ii", hello[len(_MAGIC):])
                if hello[:len(_MAGIC)] != _MAGIC or w != self.world_size or not 0 < r < w or r in peers:
                    conn.close()
                    continue
                peers[r] = conn
            srv.close()
            self._peers = [peers[r] for r in range(1, self.world_size)]
            for c in self._peers:
                _send(c, "go")
        else:
            last = None
            while True:
                try:
                    p = int(port)
                    if rendezvous_file:  # re-read on every attempt: a stale file from an older run is overwritten
                        with open(rendezvous_file) as f:
                            p = int(f.read().strip())
                    s = socket.create_connection((addr, p), timeout=5.0)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    s.settimeout(timeout)
                    s.sendall(_MAGIC + struct.pack("<ii", self.rank, self.world_size))
                    if _recv(s) == "go":
                        self._sock, self.port = s, p
                        break
                except (OSError, ValueError, ConnectionError, EOFError) as e:
                    last = e
                if time.time() > deadline:
                    raise RuntimeError("host control plane: rank %d could not reach rank 0 (%r)" % (self.rank, last))
                time.sleep(0.05)

    @classmethod
    def from_env(cls, timeout=120.0):
        """RANK / WORLD_SIZE / MASTER_ADDR as set by the launcher.  The launcher's own store owns
        MASTER_PORT, so rank 0 takes an ephemeral port (or TSR_CTRL_PORT) and publishes it in a file
        named after MASTER_PORT in the temp directory (one node: the file system is shared)."""
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        if addr == "localhost":
            addr = "127.0.0.1"
        if "TSR_CTRL_PORT" in os.environ:
            return cls(rank, world, addr, int(os.environ["TSR_CTRL_PORT"]), timeout=timeout)
        tag = "%s_%s_%s" % (addr, os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"))
        path = os.path.join(tempfile.gettempdir(), "tsr_ctrl_%s.port" % tag.replace("/", "_"))
        if rank == 0 and world > 1:
            try:
                os.unlink(path)
            except OSError:
                pass
        return cls(rank, world, addr, 0, rendezvous_file=path, timeout=timeout)

    def allgather(self, obj):
        if self.world_size == 1:
            return [obj]
        if self.rank == 0:
            parts = [obj] + [_recv(c) for c in self._peers]
            for c in self._peers:
                _send(c, parts)
            return parts
        _send(self._sock, obj)
        return _recv(self._sock)

    def barrier(self):
        self.allgather(None)

    def get_rank(self):
        return self.rank

    def get_world_size(self):
        return self.world_size

    def close(self):
        for c in self._peers:
            c.close()
        if self._sock is not None:
            self._sock.close()
        self._peers, self._sock = [], None
        if self._file:
            try:
                os.unlink(self._file)
            except OSError:
                pass
            self._file = None


def _allgather(group, obj):
    if hasattr(group, "allgather"):
        return group.allgather(obj)
    parts = [None] * group.get_world_size()  # torch.distributed (module or look-alike)
    group.all_gather_object(parts, obj)
    return parts


def exchange_unique_id(group, make_id, src=0):
    """Rank `src` calls make_id() (tsr_hip.unique_id); everyone returns the same 128 bytes."""
    uid = _allgather(group, make_id() if group.get_rank() == src else None)[src]
    if not isinstance(uid, (bytes, bytearray)) or len(uid) != 128:
        raise RuntimeError("RCCL unique id hand-over failed")
    return bytes(uid)


def broadcast_object(group, obj, src=0):
    """Python object of rank `src` on every rank (host channel; small objects only)."""
    return _allgather(group, obj if group.get_rank() == src else None)[src]


def placeholder_bank(lengths, feature_shape):
    """A bank of the right geometry (zero weights) for ranks that will receive the real one
    by broadcast: tsr_bank_broadcast requires identical template lengths on every rank."""
    shape = tuple(int(x) for x in np.atleast_1d(feature_shape))
    return tuple((np.zeros((int(L),) + shape, np.float32), 0.0) for L in lengths)


def max_over_ranks(group, *values):
    """Element-wise MAX of python floats across ranks (bench timing contract)."""
    rows = _allgather(group, [float(v) for v in values])
    return [max(r[i] for r in rows) for i in range(len(values))]


def gather_scores(group, scores, ids, bounds_frames):
    """All-gather per-rank (scores, ids) shards back into corpus order on every rank (host side;
    only for callers that want the whole detection array in one place)."""
    parts = _allgather(group, (np.asarray(scores), np.asarray(ids)))
    s = np.concatenate([p[0] for p in parts])
    i = np.concatenate([p[1] for p in parts])
    if bounds_frames is not None and len(s) != int(bounds_frames[-1]):
        raise RuntimeError("gathered %d frames, expected %d" % (len(s), int(bounds_frames[-1])))
    return s, i
