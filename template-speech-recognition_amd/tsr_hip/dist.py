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
import hashlib
import hmac
import io
import ipaddress
import os
import secrets
import socket
import stat
import struct
import tempfile
import time

import numpy as np

_MAGIC = b"TSRCTRL2"
_MAX_MSG = 1 << 31          # a control message is never larger than a gathered score shard
_HANDSHAKE_TIMEOUT = 2.0    # a peer that does not complete the handshake within this is dropped


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


# ---- wire format: a closed, tagged encoding (nothing on the wire can name code) ------------
# N none | T/F bool | I int64 | D float64 | S utf-8 | B bytes | L list | U tuple | M dict (str keys)
# | A numpy array in .npy form (np.load(allow_pickle=False): numeric dtypes only)
def _enc(obj, out):
    if obj is None:
        out.append(b"N")
    elif isinstance(obj, (bool, np.bool_)):
        out.append(b"T" if obj else b"F")
    elif isinstance(obj, (int, np.integer)):
        out.append(b"I" + struct.pack("<q", int(obj)))
    elif isinstance(obj, (float, np.floating)):
        out.append(b"D" + struct.pack("<d", float(obj)))
    elif isinstance(obj, str):
        b = obj.encode("utf-8")
        out.append(b"S" + struct.pack("<Q", len(b)) + b)
    elif isinstance(obj, (bytes, bytearray)):
        out.append(b"B" + struct.pack("<Q", len(obj)) + bytes(obj))
    elif isinstance(obj, (list, tuple)):
        out.append((b"L" if isinstance(obj, list) else b"U") + struct.pack("<Q", len(obj)))
        for x in obj:
            _enc(x, out)
    elif isinstance(obj, dict):
        out.append(b"M" + struct.pack("<Q", len(obj)))
        for k, v in obj.items():
            if not isinstance(k, str):
                raise TypeError("host control plane: dict keys must be str")
            _enc(k, out)
            _enc(v, out)
    elif isinstance(obj, np.ndarray):
        if obj.dtype.hasobject:
            raise TypeError("host control plane: object arrays are not sent")
        f = io.BytesIO()
        np.save(f, obj, allow_pickle=False)
        b = f.getvalue()
        out.append(b"A" + struct.pack("<Q", len(b)) + b)
    else:
        raise TypeError("host control plane: cannot send %s" % type(obj).__name__)


def encode(obj):
    out = []
    _enc(obj, out)
    return b"".join(out)


def _dec(buf, i, depth=0):
    if depth > 16:
        raise ValueError("host control plane: message nested too deeply")
    tag = buf[i:i + 1]
    i += 1
    if tag == b"N":
        return None, i
    if tag in (b"T", b"F"):
        return tag == b"T", i
    if tag == b"I":
        return struct.unpack_from("<q", buf, i)[0], i + 8
    if tag == b"D":
        return struct.unpack_from("<d", buf, i)[0], i + 8
    if tag in (b"S", b"B", b"A"):
        (n,) = struct.unpack_from("<Q", buf, i)
        i += 8
        if n > len(buf) - i:
            raise ValueError("host control plane: truncated message")
        raw = bytes(buf[i:i + n])
        if tag == b"S":
            return raw.decode("utf-8"), i + n
        if tag == b"B":
            return raw, i + n
        return np.load(io.BytesIO(raw), allow_pickle=False), i + n
    if tag in (b"L", b"U"):
        (n,) = struct.unpack_from("<Q", buf, i)
        i += 8
        if n > len(buf) - i:
            raise ValueError("host control plane: truncated message")
        items = []
        for _ in range(n):
            x, i = _dec(buf, i, depth + 1)
            items.append(x)
        return (items if tag == b"L" else tuple(items)), i
    if tag == b"M":
        (n,) = struct.unpack_from("<Q", buf, i)
        i += 8
        if n > len(buf) - i:
            raise ValueError("host control plane: truncated message")
        d = {}
        for _ in range(n):
            k, i = _dec(buf, i, depth + 1)
            if not isinstance(k, str):
                raise ValueError("host control plane: dict key is not a string")
            d[k], i = _dec(buf, i, depth + 1)
        return d, i
    raise ValueError("host control plane: unknown tag %r" % tag)


def decode(data):
    obj, i = _dec(memoryview(data).tobytes(), 0)
    if i != len(data):
        raise ValueError("host control plane: trailing bytes in message")
    return obj


def _send(sock, obj):
    data = encode(obj)
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
    if n > _MAX_MSG:
        raise ConnectionError("host control plane: oversized message announced")
    return decode(_recv_exact(sock, n))


def _loopback(addr):
    """The control plane is one node by construction; it only ever binds or dials a
    loopback address.  Anything else is refused, not silently rewritten."""
    if addr == "localhost":
        return "127.0.0.1"
    try:
        ok = ipaddress.ip_address(addr).is_loopback
    except ValueError:
        ok = False
    if not ok:
        raise ValueError("host control plane: %r is not a loopback address (single-node control plane)" % (addr,))
    return addr


def _private_dir():
    """A directory only this user can enter (0700, owned by us, not a symlink) for the rendezvous file."""
    base = os.environ.get("XDG_RUNTIME_DIR") or tempfile.gettempdir()
    d = os.path.join(base, "tsr_ctrl_%d" % os.getuid())
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise RuntimeError("host control plane: %s is not a private directory of this user" % d)
    return d


def _mac(secret, *parts):
    return hmac.new(secret, b"|".join(parts), hashlib.sha256).digest()


class HostGroup:
    """Rank 0 listens on a loopback (addr, port); ranks 1..world-1 connect and prove knowledge of the
    job's shared secret in both directions (HMAC-SHA256 challenge) before any message is read.
    Collectives are an all-gather of encoded objects through rank 0 -- a few hundred bytes per call, used
    a handful of times per run, never inside the data path.

    The secret is `secret` / $TSR_CTRL_SECRET (hex) when the launcher provides one (bench.py's own
    launcher does); otherwise rank 0 draws one and publishes "port secret" in `rendezvous_file`, created
    O_EXCL with mode 0600 inside a 0700 directory of this user."""

    def __init__(self, rank, world_size, addr="127.0.0.1", port=0, rendezvous_file=None, timeout=120.0, secret=None):
        self.rank, self.world_size = int(rank), int(world_size)
        self._peers, self._sock, self._file = [], None, None
        if self.world_size == 1:
            return
        addr = _loopback(addr)
        if secret is None and os.environ.get("TSR_CTRL_SECRET"):
            secret = bytes.fromhex(os.environ["TSR_CTRL_SECRET"])
        if secret is None and not rendezvous_file:
            raise ValueError("host control plane: a shared secret (secret= / TSR_CTRL_SECRET) or a rendezvous file is required")
        deadline = time.time() + timeout
        if self.rank == 0:
            if secret is None:
                secret = secrets.token_bytes(32)
            srv = socket.socket(socket.AF_INET6 if ":" in addr else socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, int(port)))
            srv.listen(self.world_size + 8)
            self.port = srv.getsockname()[1]
            if rendezvous_file:
                try:
                    os.unlink(rendezvous_file)   # a stale file of an older job with the same tag
                except OSError:
                    pass
                fd = os.open(rendezvous_file, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
                with os.fdopen(fd, "w") as f:
                    f.write("%d %s\n" % (self.port, secret.hex()))
                self._file = rendezvous_file
            peers = {}
            try:
                while len(peers) < self.world_size - 1:
                    left = deadline - time.time()
                    if left <= 0:
                        raise RuntimeError("host control plane: only %d of %d ranks arrived within %.0f s"
                                           % (len(peers) + 1, self.world_size, timeout))
                    srv.settimeout(min(left, 1.0))
                    try:
                        conn, _ = srv.accept()
                    except socket.timeout:
                        continue
                    r = self._admit(conn, secret, peers)
                    if r is None:       # a stray, short, silent or unauthenticated connection: drop it, go on
                        conn.close()
                        continue
                    conn.settimeout(timeout)
                    peers[r] = conn
            except BaseException:
                for c in peers.values():
                    c.close()
                self.close()
                raise
            finally:
                srv.close()
            self._peers = [peers[r] for r in range(1, self.world_size)]
            for c in self._peers:
                _send(c, "go")
        else:
            last = None
            while True:
                s = None
                try:
                    p, sec = int(port), secret
                    if rendezvous_file:  # re-read on every attempt: rank 0 replaces a stale file of an older run
                        p, sec = self._read_rendezvous(rendezvous_file, sec)
                    s = socket.create_connection((addr, p), timeout=5.0)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    s.settimeout(timeout)
                    nonce = secrets.token_bytes(16)
                    s.sendall(_MAGIC + struct.pack("<ii", self.rank, self.world_size) + nonce)
                    reply = _recv_exact(s, 48)          # rank 0's nonce + its proof over ours
                    hello = struct.pack("<ii", self.rank, self.world_size)
                    if not hmac.compare_digest(reply[16:], _mac(sec, b"srv", nonce, hello)):
                        raise ConnectionError("host control plane: rank 0 did not prove the job secret")
                    s.sendall(_mac(sec, b"cli", reply[:16], hello))
                    if _recv(s) == "go":
                        self._sock, self.port = s, p
                        break
                    raise ConnectionError("host control plane: unexpected reply")
                except (OSError, ValueError, ConnectionError, EOFError, struct.error) as e:
                    last = e
                    if s is not None:
                        s.close()
                if time.time() > deadline:
                    raise RuntimeError("host control plane: rank %d could not reach rank 0 (%r)" % (self.rank, last))
                time.sleep(0.05)

    def _admit(self, conn, secret, peers):
        """Rank 0's side of the handshake; returns the peer's rank or None.  Never raises on a bad peer."""
        try:
            conn.settimeout(_HANDSHAKE_TIMEOUT)
            conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            hello = _recv_exact(conn, len(_MAGIC) + 8 + 16)
            if hello[:len(_MAGIC)] != _MAGIC:
                return None
            body, nonce = hello[len(_MAGIC):len(_MAGIC) + 8], hello[len(_MAGIC) + 8:]
            r, w = struct.unpack("<ii", body)
            if w != self.world_size or not 0 < r < w or r in peers:
                return None
            mine = secrets.token_bytes(16)
            conn.sendall(mine + _mac(secret, b"srv", nonce, body))
            proof = _recv_exact(conn, 32)
            if not hmac.compare_digest(proof, _mac(secret, b"cli", mine, body)):
                return None
            return r
        except (OSError, ConnectionError, struct.error):
            return None

    @staticmethod
    def _read_rendezvous(path, secret):
        fd = os.open(path, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
        with os.fdopen(fd) as f:
            st = os.fstat(f.fileno())
            if st.st_uid != os.getuid() or (st.st_mode & 0o077):
                raise ValueError("host control plane: rendezvous file is not private to this user")
            fields = f.read().split()
        if len(fields) != 2:
            raise ValueError("host control plane: rendezvous file incomplete")
        return int(fields[0]), (secret if secret is not None else bytes.fromhex(fields[1]))

    @classmethod
    def from_env(cls, timeout=120.0):
        """RANK / WORLD_SIZE / MASTER_ADDR as set by the launcher.  The launcher's own store owns
        MASTER_PORT, so rank 0 takes an ephemeral port (or TSR_CTRL_PORT, which then needs
        TSR_CTRL_SECRET) and publishes port and secret in a 0600 file named after MASTER_PORT in a
        private directory (one node: the file system is shared)."""
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        if world == 1:
            return cls(rank, world)
        # The control plane is one node by construction, so the socket always binds and dials 127.0.0.1; MASTER_ADDR
        # (the node's own name under `torchrun --standalone` / c10d rendezvous) only names the job in the rendezvous
        # tag.  An explicit addr= given to the constructor is still held to loopback by _loopback().
        addr = "127.0.0.1"
        if "TSR_CTRL_PORT" in os.environ:
            return cls(rank, world, addr, int(os.environ["TSR_CTRL_PORT"]), timeout=timeout)
        tag = "%s_%s_%s" % (os.environ.get("MASTER_ADDR", addr), os.environ.get("MASTER_PORT", "0"),
                            os.environ.get("TORCHELASTIC_RUN_ID", "none"))
        tag = "".join(ch if ch.isalnum() or ch in "._-" else "_" for ch in tag)
        path = os.path.join(_private_dir(), "%s.port" % tag)
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
