"""Multi-GPU host logic: one process per GPU, utterances sharded, bank broadcast once.

The data path has no collective (utterances are independent, get_train_data.py:1269-1321);
the only exchange is the RCCL broadcast of the bank, which needs the 128-byte RCCL unique id
handed from rank 0 to the others over some host channel -- here torch.distributed (gloo or
nccl), used purely as control plane."""
import numpy as np


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


def exchange_unique_id(dist, make_id, src=0):
    """Rank `src` calls make_id() (tsr_hip.unique_id); everyone returns the same 128 bytes."""
    box = [make_id() if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    uid = box[0]
    if not isinstance(uid, (bytes, bytearray)) or len(uid) != 128:
        raise RuntimeError("RCCL unique id hand-over failed")
    return bytes(uid)


def placeholder_bank(lengths, feature_shape):
    """A bank of the right geometry (zero weights) for ranks that will receive the real one
    by broadcast: tsr_bank_broadcast requires identical template lengths on every rank."""
    shape = tuple(int(x) for x in np.atleast_1d(feature_shape))
    return tuple((np.zeros((int(L),) + shape, np.float32), 0.0) for L in lengths)


def max_over_ranks(dist, *values):
    """Element-wise MAX of python floats across ranks (bench timing contract)."""
    import torch
    t = torch.tensor(values, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(x) for x in t]


def gather_scores(dist, scores, ids, bounds_frames):
    """All-gather per-rank (scores, ids) shards back into corpus order on every rank (host side;
    only for callers that want the whole detection array in one place)."""
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, (np.asarray(scores), np.asarray(ids)))
    s = np.concatenate([p[0] for p in parts])
    i = np.concatenate([p[1] for p in parts])
    if bounds_frames is not None and len(s) != int(bounds_frames[-1]):
        raise RuntimeError("gathered %d frames, expected %d" % (len(s), int(bounds_frames[-1])))
    return s, i
