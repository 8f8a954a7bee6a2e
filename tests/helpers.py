"""Golden-fixture helpers shared by the CPU and GPU parity tests."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def unpack(bits, shape):
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape))
    return np.unpackbits(bits)[:n].reshape(shape).astype(np.uint8)


def golden_bank(g, prefix, M):
    return tuple((g["%sW%d" % (prefix, m)], float(g["%sc%d" % (prefix, m)])) for m in range(int(M)))


def rel_err(a, b):
    """The parity measure of BASELINE.md section 3: |a-b| / max(|b|, 1)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0)) if a.size else 0.0
