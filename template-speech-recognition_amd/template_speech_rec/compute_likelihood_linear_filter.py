"""Drop-in for the reference's compute_likelihood_linear_filter.pyx on MI355X.

Same functions, argument order, dtypes and return values.  Every result is a
freshly allocated NumPy array; inputs are never modified.  Raw scores are
accumulated in the reference's order (f32, tau -> frequency -> part), so they
are bit-identical to the Cython module's.
"""
import numpy as np

from tsr_hip import _lib
from tsr_hip.api import default_ctx
from ._buffers import buffer

_i64 = _lib._i64


def _flat(a, lead=1):
    return int(np.prod(a.shape[lead:], dtype=np.int64))


def _check_window(T, L):
    if T < L:
        # the reference wraps `F_dim_0 - LF_dim_0 + 1` in a uint16 and reads out of
        # bounds (compute_likelihood_linear_filter.pyx:157); refuse instead
        raise ValueError("filter is longer than the utterance (T=%d < L=%d)" % (T, L))


def detect(F, LF):
    """compute_likelihood_linear_filter.pyx:135-176.
    F uint8 [T, F, K], LF float32 [L, F, K] -> float32 [T]; zeros for t > T-L."""
    F = buffer(F, np.uint8, 3, "F", "UINT_t")
    LF = buffer(LF, np.float32, 3, "LF", "DTYPE_t")
    T, L, D = F.shape[0], LF.shape[0], _flat(F)
    if _flat(LF) != D:
        raise ValueError("F and LF disagree on the feature shape: %r vs %r" % (F.shape[1:], LF.shape[1:]))
    _check_window(T, L)
    out = np.zeros(T, np.float32)
    _lib.check(_lib.lib().tsr_detect_u8(default_ctx()._h, _lib.ptr(F), T, D, _lib.ptr(LF), L, _lib.ptr(out)))
    return out


def _detect_f32(F, LF, ndim):
    F = buffer(F, np.float32, ndim, "F", "DTYPE_t")
    LF = buffer(LF, np.float32, ndim, "LF", "DTYPE_t")
    T, L, D = F.shape[0], LF.shape[0], _flat(F)
    if _flat(LF) != D:
        raise ValueError("F and LF disagree on the feature shape: %r vs %r" % (F.shape[1:], LF.shape[1:]))
    _check_window(T, L)
    out = np.zeros(T, np.float32)
    _lib.check(_lib.lib().tsr_detect_f32(default_ctx()._h, _lib.ptr(F), T, D, _lib.ptr(LF), L, _lib.ptr(out)))
    return out


def _check_float_features(F):
    """The buffer acquisition detect_float / detect_float2d would perform on a feature window."""
    return buffer(F, np.float32, 3 if getattr(F, "ndim", 0) != 2 else 2, "F", "DTYPE_t")


def detect_float(F, LF):
    """:178-219.  float32 features [T, F, K] (e.g. background-padded windows)."""
    return _detect_f32(F, LF, 3)


def detect_float2d(F, LF):
    """:281-318.  float32 [T, D] x [L, D]."""
    return _detect_f32(F, LF, 2)


def detect_spectral(F, LF):
    """:322-359.  Identical arithmetic to detect_float2d."""
    return _detect_f32(F, LF, 2)


def detect_float_max(F, LF, cs):
    """:221-278.  LF float32 [M, L, F, K], cs float32 [M] ->
    (float32 [T], uint16 [T]); first maximum wins; zeros for t > T-L."""
    F = buffer(F, np.float32, 3, "F", "DTYPE_t")
    LF = buffer(LF, np.float32, 4, "LF", "DTYPE_t")
    cs = buffer(cs, np.float32, 1, "cs", "DTYPE_t")
    T, M, L, D = F.shape[0], LF.shape[0], LF.shape[1], _flat(F)
    if _flat(LF, 2) != D or cs.shape[0] != M:
        raise ValueError("F, LF and cs disagree on their shapes")
    _check_window(T, L)
    out = np.zeros(T, np.float32)
    ids = np.zeros(T, np.uint16)
    _lib.check(_lib.lib().tsr_detect_f32_max(default_ctx()._h, _lib.ptr(F), T, D, _lib.ptr(LF), M, L,
                                             _lib.ptr(cs), _lib.ptr(out), _lib.ptr(ids)))
    return out, ids


def get_max_detect_examples(F, LF, lengths):
    """:21-76.  F uint8 [N, T, F, K], LF float64 [L, F, K], lengths int64 [N] ->
    float64 [N]; the filter is truncated at the end of each example."""
    F = buffer(F, np.uint8, 4, "F", "UINT_t")
    LF = buffer(LF, np.float64, 3, "LF", "float64_t")
    lengths = buffer(lengths, np.int64, 1, "lengths", "int64_t")
    N, T, L, D = F.shape[0], F.shape[1], LF.shape[0], _flat(F, 2)
    if _flat(LF) != D or lengths.shape[0] != N:
        raise ValueError("F, LF and lengths disagree on their shapes")
    out = np.zeros(N, np.float64)
    _lib.check(_lib.lib().tsr_max_detect_examples_u8(default_ctx()._h, _lib.ptr(F), N, T, D, _lib.ptr(LF),
                                                     L, _lib.ptr(lengths), _lib.ptr(out)))
    return out


def detect_examples(F, LF, lengths, max_length):
    """:79-131.  Dead in the reference: it declares a float64 result buffer and
    allocates float32 (:107-109), so every call raises before computing anything.
    Kept with the same outcome."""
    buffer(F, np.uint8, 4, "F", "UINT_t")
    buffer(LF, np.float64, 3, "LF", "float64_t")
    buffer(lengths, np.int64, 1, "lengths", "int64_t")
    raise ValueError("Buffer dtype mismatch, expected 'float64_t' but got 'float'")
