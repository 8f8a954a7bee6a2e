"""Argument checking that mirrors Cython's typed-buffer acquisition, so callers see
the same exceptions as with the reference's compiled modules (wrong dtype or rank ->
ValueError, non-array -> TypeError; non-contiguous arrays and bool-for-uint8 are
accepted)."""
import numpy as np

_C_NAMES = {np.dtype(np.float64): "double", np.dtype(np.float32): "float",
            np.dtype(np.int64): "long", np.dtype(np.int32): "int", np.dtype(np.uint8): "unsigned char",
            np.dtype(np.int8): "signed char", np.dtype(np.uint16): "unsigned short",
            np.dtype(np.int16): "short", np.dtype(np.uint32): "unsigned int",
            np.dtype(np.uint64): "unsigned long", np.dtype(np.bool_): "bool"}


def buffer(a, dtype, ndim, argname, typedef):
    if not isinstance(a, np.ndarray):
        raise TypeError("Argument '%s' has incorrect type (expected numpy.ndarray, got %s)"
                        % (argname, type(a).__name__))
    if a.ndim != ndim:
        raise ValueError("Buffer has wrong number of dimensions (expected %d, got %d)" % (ndim, a.ndim))
    want = np.dtype(dtype)
    if a.dtype != want and not (want == np.uint8 and a.dtype == np.bool_):
        raise ValueError("Buffer dtype mismatch, expected '%s' but got '%s'"
                         % (typedef, _C_NAMES.get(a.dtype, str(a.dtype))))
    return np.ascontiguousarray(a.view(np.uint8) if a.dtype == np.bool_ else a)
