"""Drop-in for the reference's code_parts.pyx on MI355X: code_parts,
code_parts_fast, spread_patches, spread_graph_patches.

Not provided: code_parts_mask / get_parts_mask index their
mask with absolute coordinates (code_parts.pyx:44) and read out of bounds in the
reference; code_parts_mag_aux relies on Python-2 integer division.
"""
import numpy as np

from tsr_hip import _lib
from tsr_hip.api import default_ctx
from ._buffers import buffer


def _code(X, log_parts, log_invparts, threshold, fast):
    X = buffer(X, np.uint8, 3, "X", "UINT_t")
    lp = buffer(log_parts, np.float64, 4, "log_parts", "DTYPE_t")
    lip = buffer(log_invparts, np.float64, 4, "log_invparts", "DTYPE_t")
    if lip.shape != lp.shape or lp.shape[3] != X.shape[2]:
        raise ValueError("log_parts / log_invparts / X disagree on their shapes")
    K, P0, P1, E = lp.shape
    if X.shape[0] < P0 or X.shape[1] < P1:
        raise ValueError("the map %r is smaller than the %dx%d patch" % (X.shape[:2], P0, P1))
    threshold = int(threshold)
    if not -2 ** 31 <= threshold < 2 ** 31:
        raise OverflowError("value too large to convert to int")
    out = np.zeros((X.shape[0] - P0 + 1, X.shape[1] - P1 + 1, K + 1), np.float64)
    _lib.check(_lib.lib().tsr_code_parts(default_ctx()._h, _lib.ptr(X), X.shape[0], X.shape[1], E,
                                         _lib.ptr(lp), _lib.ptr(lip), K, P0, P1, threshold, fast,
                                         _lib.ptr(out)))
    return out


def code_parts(X, log_parts, log_invparts, threshold):
    """code_parts.pyx:70-135.  X uint8 [S, T, E]; log_parts, log_invparts float64
    [K, P0, P1, E] -> float64 [S-P0+1, T-P1+1, K+1].  Slot 0 is -inf where the patch
    holds >= threshold edges (the part log-likelihoods follow), else the row is 0.
    A negative threshold makes every location background, as in the compiled
    reference (unsigned comparison)."""
    return _code(X, log_parts, log_invparts, threshold, 0)


def code_parts_fast(X, log_parts, log_invparts, threshold):
    """code_parts.pyx:247-295.  Same sums; slot 0 is sum_k(out[k+1] - 1) for coded
    locations and 1.0 for background ones."""
    return _code(X, log_parts, log_invparts, threshold, 1)


def code_parts_mag_aux(X_edges, X_mags, X_aux, log_parts, log_invparts, log_mag_parts, log_mag_invparts,
                       log_aux_parts, log_aux_invparts, X_thresh_mask):
    """code_parts.pyx:138-242.  Edge patches plus magnitude and auxiliary channels; a location is coded
    iff X_thresh_mask at the patch centre is set.  Returns float64 [S-P0+1, T-P1+1, K+1], bit-identical to
    the reference compiled with Python-2 semantics (its `/` on unsigned ints is integer division)."""
    Xe = buffer(X_edges, np.uint8, 3, "X_edges", "UINT_t")
    Xm = buffer(X_mags, np.uint8, 2, "X_mags", "UINT_t")
    Xa = buffer(X_aux, np.uint8, 2, "X_aux", "UINT_t")
    lp = buffer(log_parts, np.float64, 4, "log_parts", "DTYPE_t")
    lip = buffer(log_invparts, np.float64, 4, "log_invparts", "DTYPE_t")
    lmp = buffer(log_mag_parts, np.float64, 2, "log_mag_parts", "DTYPE_t")
    lmip = buffer(log_mag_invparts, np.float64, 2, "log_mag_invparts", "DTYPE_t")
    lap = buffer(log_aux_parts, np.float64, 2, "log_aux_parts", "DTYPE_t")
    laip = buffer(log_aux_invparts, np.float64, 2, "log_aux_invparts", "DTYPE_t")
    mk = buffer(X_thresh_mask, np.uint8, 2, "X_thresh_mask", "UINT_t")
    K, P0, P1, E = lp.shape
    X0, X1 = Xe.shape[0], Xe.shape[1]
    MD, AD = Xm.shape[1], Xa.shape[1]
    if (lip.shape != lp.shape or Xe.shape[2] != E or Xm.shape[0] != X0 or Xa.shape[0] != X0 or mk.shape != (X0, X1)
            or lmp.shape != (K, P0 + MD) or lmip.shape != lmp.shape or lap.shape != (K, P0 * AD) or laip.shape != lap.shape):
        raise ValueError("code_parts_mag_aux: inconsistent shapes (the reference indexes out of bounds here)")
    if X0 < P0 or X1 < P1:
        raise ValueError("patch larger than the map")
    out = np.empty((X0 - P0 + 1, X1 - P1 + 1, K + 1), np.float64)
    _lib.check(_lib.lib().tsr_code_parts_mag_aux(default_ctx()._h, _lib.ptr(Xe), _lib.ptr(Xm), _lib.ptr(Xa), X0, X1, E, MD, AD,
                                                 _lib.ptr(lp), _lib.ptr(lip), _lib.ptr(lmp), _lib.ptr(lmip), _lib.ptr(lap),
                                                 _lib.ptr(laip), K, P0, P1, _lib.ptr(mk), _lib.ptr(out)))
    return out


def spread_patches(X, spread_0_dim, spread_1_dim, num_parts):
    """code_parts.pyx:406-457.  X int64 [S, T] of part ids (0 = none) ->
    uint8 [S, T, num_parts]; out[i, j, p-1] = 1 iff part p occurs within the clipped
    (2*spread_0_dim+1) x (2*spread_1_dim+1) box around (i, j)."""
    X = buffer(X, np.int64, 2, "X", "int64_t")
    num_parts = int(num_parts)
    out = np.zeros(X.shape + (num_parts,), np.uint8)
    _lib.check(_lib.lib().tsr_spread_patches(default_ctx()._h, _lib.ptr(X), X.shape[0], X.shape[1],
                                             int(spread_0_dim), int(spread_1_dim), num_parts,
                                             _lib.ptr(out)))
    return out


def spread_graph_patches(X, part_graph):
    """code_parts.pyx:459-503.  In place on X uint8 [S, T, K]; returns None."""
    Xc = buffer(X, np.uint8, 3, "X", "uint8_t")
    g = buffer(part_graph, np.uint8, 2, "part_graph", "uint8_t")
    K = Xc.shape[2]
    if g.shape[0] < K or g.shape[1] < K:
        raise ValueError("part_graph must be at least [%d, %d]" % (K, K))
    g = np.ascontiguousarray(g[:K, :K])
    _lib.check(_lib.lib().tsr_spread_graph_patches(default_ctx()._h, _lib.ptr(Xc), Xc.shape[0],
                                                   Xc.shape[1], K, _lib.ptr(g)))
    if Xc is not X:  # non-contiguous or bool input: write the result back through the view
        X[...] = Xc.view(X.dtype) if X.dtype == np.bool_ else Xc
    return None
