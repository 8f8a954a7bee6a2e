"""Drop-in for the reference's spread_waliji_patches.pyx (:15-59): the same box
dilation of a part-id map as code_parts.spread_patches."""
import numpy as np

from tsr_hip import _lib
from tsr_hip.api import default_ctx
from ._buffers import buffer


def spread_waliji_patches(X, spread_0_dim, spread_1_dim, num_parts):
    """X int64 [S, T] (0 = no part) -> uint8 [S, T, num_parts]."""
    X = buffer(X, np.int64, 2, "X", "int64_t")
    num_parts = int(num_parts)
    out = np.zeros(X.shape + (num_parts,), np.uint8)
    _lib.check(_lib.lib().tsr_spread_patches(default_ctx()._h, _lib.ptr(X), X.shape[0], X.shape[1],
                                             int(spread_0_dim), int(spread_1_dim), num_parts,
                                             _lib.ptr(out)))
    return out
