"""world_size-2 tests (CPU) of the multi-GPU host logic: utterance sharding, the RCCL unique-id
hand-over, placeholder banks, max-over-ranks timing and score gathering -- once over the
package's own stdlib-socket control plane (tsr_hip.dist.HostGroup, what bench.py uses) and once
over torch.distributed/gloo.  The GPU work itself needs no collective; what is exercised here is
everything around it that bench.py relies on for N > 1."""
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "template-speech-recognition_amd"))

from tsr_hip import dist as tdist  # noqa: E402


def test_shard_utterances_balanced_and_contiguous():
    rng = np.random.default_rng(0)
    lengths = rng.integers(50, 400, size=1000)
    for world in (1, 2, 3, 8):
        b = tdist.shard_utterances(lengths, world)
        assert b[0] == 0 and b[-1] == 1000 and np.all(np.diff(b) >= 0) and len(b) == world + 1
        frames = np.array([lengths[b[r]:b[r + 1]].sum() for r in range(world)])
        assert frames.sum() == lengths.sum()
        assert frames.max() - frames.min() <= 2 * lengths.max()
    assert tdist.shard_utterances([], 4).tolist() == [0, 0, 0, 0, 0]
    assert tdist.shard_utterances([7], 2).tolist() in ([0, 0, 1], [0, 1, 1])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _exercise(dist, rank, world):
    uid = tdist.exchange_unique_id(dist, lambda: bytes(range(128)))
    assert uid == bytes(range(128))
    # utterance-sharded scoring with a stand-in scorer (per-frame sum): every rank handles
    # its own utterances, nothing is exchanged until the gather
    rng = np.random.default_rng(5)
    lengths = rng.integers(3, 30, size=41)
    feats = rng.integers(0, 2, size=(int(lengths.sum()), 6)).astype(np.uint8)
    offs = np.concatenate(([0], np.cumsum(lengths)))
    b = tdist.shard_utterances(lengths, world)
    lo, hi = int(offs[b[rank]]), int(offs[b[rank + 1]])
    local_scores = feats[lo:hi].sum(1).astype(np.float32)
    local_ids = np.full(hi - lo, rank, np.int32)
    s, i = tdist.gather_scores(dist, local_scores, local_ids, offs)
    assert np.array_equal(s, feats.sum(1).astype(np.float32))
    assert np.array_equal(i[:int(offs[b[1]])], np.zeros(int(offs[b[1]]), np.int32))
    mx = tdist.max_over_ranks(dist, float(rank + 1), 10.0 - rank)
    assert mx == [float(world), 10.0]
    pb = tdist.placeholder_bank([3, 5], (4, 2))
    assert [w.shape for w, _ in pb] == [(3, 4, 2), (5, 4, 2)] and not pb[0][0].any()
    assert tdist.broadcast_object(dist, {"bank": rank} if rank == 0 else None) == {"bank": 0}
    dist.barrier()


def _worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _exercise(dist, rank, world)
        out.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def _worker_sockets(rank, world, port, out, via_env):
    try:
        if via_env:  # the way bench.py builds it under a launcher: env + port file
            os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                              TORCHELASTIC_RUN_ID="pytest%d" % port)
            os.environ.pop("TSR_CTRL_PORT", None)
            g = tdist.HostGroup.from_env(timeout=60)
        else:
            g = tdist.HostGroup(rank, world, "127.0.0.1", port, timeout=60, secret=b"pytest-secret-%d" % port)
        try:
            _exercise(g, rank, world)
            out.put((rank, "ok"))
        finally:
            g.close()
    except Exception as e:  # pragma: no cover
        out.put((rank, repr(e)))


def _run(target, world, extra=()):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, out) + tuple(extra)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(out.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    return res


def test_world_size_2_host_group_sockets():
    assert _run(_worker_sockets, 2, (False,)) == {0: "ok", 1: "ok"}


def test_world_size_3_host_group_from_env():
    assert _run(_worker_sockets, 3, (True,)) == {0: "ok", 1: "ok", 2: "ok"}


def test_host_group_world_size_1_is_local():
    g = tdist.HostGroup(0, 1)
    assert g.allgather(5) == [5] and tdist.max_over_ranks(g, 1.5, 2.0) == [1.5, 2.0]
    g.barrier()
    g.close()


def test_wire_encoding_is_closed_and_round_trips():
    obj = {"a": [1, 2.5, None, True, "x", b"\x00\x01"], "t": (np.arange(6, dtype=np.float32).reshape(2, 3), 3),
           "bank": ((np.ones((2, 3), np.float32), 1.5),)}
    back = tdist.decode(tdist.encode(obj))
    assert back["a"] == obj["a"] and isinstance(back["t"], tuple) and back["t"][1] == 3
    assert np.array_equal(back["t"][0], obj["t"][0]) and back["t"][0].dtype == np.float32
    assert np.array_equal(back["bank"][0][0], obj["bank"][0][0]) and back["bank"][0][1] == 1.5
    import pickle
    import pytest
    with pytest.raises(TypeError):
        tdist.encode(object())                       # nothing outside the closed set is sent
    with pytest.raises(TypeError):
        tdist.encode(np.array([object()], dtype=object))
    with pytest.raises(ValueError):
        tdist.decode(pickle.dumps({"x": 1}))         # ... and a pickle is not a message
    with pytest.raises(ValueError):
        tdist.decode(tdist.encode([1, 2]) + b"N")


def test_control_plane_is_loopback_only_and_needs_a_secret():
    import pytest
    with pytest.raises(ValueError, match="loopback"):
        tdist.HostGroup(0, 2, "10.1.2.3", 0, secret=b"s")
    with pytest.raises(ValueError, match="loopback"):
        tdist.HostGroup(1, 2, "example.org", 1234, secret=b"s")
    with pytest.raises(ValueError, match="secret"):
        tdist.HostGroup(0, 2, "127.0.0.1", 0)


def _worker_hostile(rank, world, port, out):
    """Rank 0 must survive stray / short / silent / unauthenticated connections made before rank 1 arrives."""
    import struct
    import time
    try:
        secret = b"right-secret"
        if rank == 1:
            time.sleep(0.3)
            s1 = socket.create_connection(("127.0.0.1", port))    # silent: never sends a byte
            s2 = socket.create_connection(("127.0.0.1", port))    # short: closes mid-handshake
            s2.sendall(b"TSR")
            s2.close()
            s3 = socket.create_connection(("127.0.0.1", port))    # right magic and rank, wrong secret
            s3.sendall(b"TSRCTRL2" + struct.pack("<ii", 1, 2) + b"n" * 16)
            s3.recv(48)
            s3.sendall(b"x" * 32)
            try:
                bad = tdist.HostGroup(1, 2, "127.0.0.1", port, timeout=3, secret=b"wrong-secret")
                bad.close()
                out.put((rank, "a wrong secret was admitted"))
                return
            except RuntimeError:
                pass
            s1.close()
            s3.close()
        g = tdist.HostGroup(rank, world, "127.0.0.1", port, timeout=60, secret=secret)
        try:
            assert g.allgather(rank) == [0, 1]
            out.put((rank, "ok"))
        finally:
            g.close()
    except Exception as e:  # pragma: no cover
        out.put((rank, repr(e)))


def test_rank0_survives_stray_and_unauthenticated_connections():
    assert _run(_worker_hostile, 2) == {0: "ok", 1: "ok"}


def test_rendezvous_file_is_private():
    import stat
    d = tdist._private_dir()
    st = os.lstat(d)
    assert stat.S_ISDIR(st.st_mode) and not (st.st_mode & 0o077) and st.st_uid == os.getuid()


def test_product_package_does_not_import_torch():
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); import tsr_hip.dist, tsr_hip.synth; "
            "assert 'torch' not in sys.modules" % os.path.join(ROOT, "template-speech-recognition_amd"))
    assert subprocess.run([sys.executable, "-c", code]).returncode == 0


def test_world_size_2_gloo():
    assert _run(_worker, 2) == {0: "ok", 1: "ok"}
