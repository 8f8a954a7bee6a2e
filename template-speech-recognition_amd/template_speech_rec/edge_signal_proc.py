"""Drop-ins for the thresholding / spreading / magnitude functions of the reference's edge_signal_proc.py
(SURVEY 8(f) N3), on the GPU through libtsr_hip.so:

    threshold_edgemap               :1184-1252 (over threshold_edge_block :1257-1307)
    spread_edgemap                  :1397-1404 (over _spread_edges :1406-1417)
    magnitude_features              :600-670
    magnitude_features_whole_block  :673-747

Same signatures, same in-place behaviour, identical results (every comparison is the reference's float64
comparison on the same operands; the order statistics come from a radix select instead of a sort).  The
spectrogram and the 8-orientation stencil themselves (`_edge_map_no_threshold` + `_edge_map_threshold_segments`)
are served corpus-wide by get_train_data.get_edge_features / get_edge_corpus.
"""
import ctypes as C

import numpy as np

from tsr_hip import api
from tsr_hip._lib import check, lib, ptr

ABST_THRESHOLD = np.array([0.025, 0.025, 0.015, 0.015, 0.02, 0.02, 0.02, 0.02])


def _in_place(E):
    """(contiguous float64 work array, True when it IS the caller's array)."""
    if isinstance(E, np.ndarray) and E.dtype == np.float64 and E.ndim == 2 and E.flags.c_contiguous and E.flags.writeable:
        return E, True
    if not isinstance(E, np.ndarray) or E.ndim != 2:
        raise ValueError("expected a 2-d ndarray (rows = orientation blocks x frequency, columns = time)")
    return np.ascontiguousarray(E, dtype=np.float64), False


def threshold_edgemap(E, quantile_level, edge_feature_row_breaks, report_level=False, abst_threshold=ABST_THRESHOLD,
                      critical_bands=None, convert_to_binary=True, ctx=None):
    """In place: within every row block [breaks[i-1], breaks[i]) keep (as 1, or as its value when
    convert_to_binary is False) what is positive and at least max(the block's quantile, abst_threshold[i-1]);
    everything else becomes 0.  Returns (E_new, thresholds) when report_level, like the reference."""
    if critical_bands is not None:
        raise NotImplementedError("threshold_edgemap: critical_bands are not supported")
    ctx = ctx or api.default_ctx()
    breaks = np.ascontiguousarray(edge_feature_row_breaks, dtype=np.int64)
    nb = len(breaks) - 1
    abst = np.ascontiguousarray(abst_threshold, dtype=np.float64)
    if abst.shape[0] < nb:
        raise IndexError("abst_threshold has %d entries for %d row blocks" % (abst.shape[0], nb))
    work, same = _in_place(E)
    if nb > 0 and int(breaks[-1]) > work.shape[0]:
        raise ValueError("edge_feature_row_breaks reach past the %d rows of E" % work.shape[0])
    taus = np.empty(max(nb, 1), np.float64)
    check(lib().tsr_threshold_edgemap(ctx._h, ptr(work), ptr(breaks), nb, work.shape[1], float(quantile_level), ptr(abst),
                                      int(bool(convert_to_binary)), ptr(taus)))
    if not same:
        E[...] = work
    if report_level:
        return work.copy(), taus[:nb]
    return None


def spread_edgemap(T, edge_feature_row_breaks, edge_orientation, spread_length=3, ctx=None):
    """In place: every row block becomes the maximum over shifts of itself along its orientation
    (edge_orientation[i] = (row step, column step)), k = -spread_length .. spread_length, zeros outside."""
    ctx = ctx or api.default_ctx()
    breaks = np.ascontiguousarray(edge_feature_row_breaks, dtype=np.int64)
    nb = len(breaks) - 1
    orient = np.ascontiguousarray(np.asarray(edge_orientation)[:nb], dtype=np.int32).reshape(nb, 2)
    work, same = _in_place(T)
    if nb > 0 and int(breaks[-1]) > work.shape[0]:
        raise ValueError("edge_feature_row_breaks reach past the %d rows of T" % work.shape[0])
    check(lib().tsr_spread_edgemap(ctx._h, ptr(work), ptr(breaks), nb, work.shape[1], ptr(orient), int(spread_length)))
    if not same:
        T[...] = work


def _magnitude(S, block_length, spread_radius, threshold_quantile, mag_smooth_freq, mag_downsample_freq, mag_spread_length,
               whole, ctx):
    ctx = ctx or api.default_ctx()
    S = np.ascontiguousarray(S, dtype=np.float64)
    if S.ndim != 2:
        raise ValueError("S must be [frequency, time]")
    F0, T = S.shape
    down = int(mag_downsample_freq)
    F = -(-F0 // down) if down > 0 else F0
    out = np.empty((T, F), np.uint8)
    check(lib().tsr_magnitude_features(ctx._h, ptr(S), F0, T, int(block_length), int(spread_radius), float(threshold_quantile),
                                       int(mag_smooth_freq), down, int(mag_spread_length), int(whole), ptr(out)))
    return out


def magnitude_features(S, block_length, spread_radius, threshold_quantile, mag_smooth_freq, mag_downsample_freq,
                       mag_spread_length=1, ctx=None):
    """S [frequency, time] -> uint8 [time, frequency']: per frequency bin, the frames whose value reaches the
    bin's quantile inside overlapping windows of block_length frames, OR-ed over +-mag_spread_length frames."""
    return _magnitude(S, block_length, spread_radius, threshold_quantile, mag_smooth_freq, mag_downsample_freq,
                      mag_spread_length, False, ctx)


def magnitude_features_whole_block(S, block_length, spread_radius, threshold_quantile, mag_smooth_freq,
                                   mag_downsample_freq, ctx=None):
    """As magnitude_features with one quantile per window over all bins and the square spread of radius
    spread_radius."""
    return _magnitude(S, block_length, spread_radius, threshold_quantile, mag_smooth_freq, mag_downsample_freq, 0, True, ctx)
