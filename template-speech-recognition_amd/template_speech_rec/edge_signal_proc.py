"""Drop-ins for the thresholding / spreading / magnitude functions of the reference's edge_signal_proc.py
(SURVEY 8(f) N3), on the GPU through libtsr_hip.so:

    threshold_edgemap               :1184-1252 (over threshold_edge_block :1257-1307), critical_bands included
    spread_edgemap                  :1397-1404 (over _spread_edges :1406-1417)
    _edge_map_threshold_segments    :1537-1675, both spread types ('unidirectional', 'all'), every keyword
    compute_edge_feature_row_breaks :1679-1693
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


def _threshold_rows(work, r0, r1, quantile_level, abst, convert_to_binary, ctx):
    """threshold_edge_block on rows [r0, r1) of the contiguous f64 array `work`, in place -> the quantile value."""
    sub = work[r0:r1]
    tau = np.empty(1, np.float64)
    check(lib().tsr_threshold_edgemap(ctx._h, ptr(sub), ptr(np.array([0, r1 - r0], np.int64)), 1, work.shape[1],
                                      float(quantile_level), ptr(np.array([abst], np.float64)), int(bool(convert_to_binary)),
                                      ptr(tau)))
    return tau[0]


def threshold_edgemap(E, quantile_level, edge_feature_row_breaks, report_level=False, abst_threshold=ABST_THRESHOLD,
                      critical_bands=None, convert_to_binary=True, ctx=None):
    """In place: within every row block [breaks[i-1], breaks[i]) keep (as 1, or as its value when
    convert_to_binary is False) what is positive and at least max(the block's quantile, abst_threshold[i-1]);
    everything else becomes 0.  Returns (E_new, thresholds) when report_level, like the reference (None otherwise).

    critical_bands (:1208-1250) is reproduced as the reference executes it, not as its docstring means it: band
    (c0, c1) of the row block [start, end) thresholds the rows [start + c0, end + c1) -- the slice the reference
    takes, clipped to the array like any NumPy slice -- in place, and then assigns those rows to the c1 - c0 rows of
    the band: whenever the two counts differ (every partition into more than one band, and a single band unless the
    block is the last one of the array) that assignment raises NumPy's ValueError, after the rows have been
    thresholded.  The same ValueError, at the same point, with the same rows already changed, is raised here."""
    ctx = ctx or api.default_ctx()
    breaks = np.ascontiguousarray(edge_feature_row_breaks, dtype=np.int64)
    nb = len(breaks) - 1
    abst = np.ascontiguousarray(abst_threshold, dtype=np.float64)
    if abst.shape[0] < nb:
        raise IndexError("abst_threshold has %d entries for %d row blocks" % (abst.shape[0], nb))
    work, same = _in_place(E)
    if nb > 0 and int(breaks[-1]) > work.shape[0]:
        raise ValueError("edge_feature_row_breaks reach past the %d rows of E" % work.shape[0])
    if critical_bands is not None:
        cbs = np.asarray(critical_bands)
        taus2 = np.empty((8, len(cbs))) if report_level else None
        E_new = np.empty(work.shape)
        rows = work.shape[0]
        try:
            for i in range(nb):
                start, end = int(breaks[i]), int(breaks[i + 1])
                assert end - start == cbs[-1, -1]
                for ci, cb in enumerate(cbs):
                    r0, r1 = min(max(start + int(cb[0]), 0), rows), min(max(end + int(cb[1]), 0), rows)
                    r1 = max(r1, r0)
                    tau = _threshold_rows(work, r0, r1, quantile_level, abst[i], convert_to_binary, ctx) if r1 > r0 else np.nan
                    dst = E_new[start + int(cb[0]):start + int(cb[1])]
                    if dst.shape[0] != r1 - r0:
                        raise ValueError("could not broadcast input array from shape (%d,%d) into shape (%d,%d)"
                                         % (r1 - r0, work.shape[1], dst.shape[0], work.shape[1]))
                    dst[...] = work[r0:r1]
                    if report_level:
                        # (a block without positive values returns the bare array there: the tuple assignment fails)
                        if np.isnan(tau):
                            raise ValueError("too many values to unpack")
                        taus2[i, ci] = tau
                if convert_to_binary and not report_level and end > start:
                    blk = E_new[start:end]
                    assert np.max(blk) <= 1 and np.min(blk) >= 0
        finally:
            if not same:
                E[...] = work
        return (E_new, taus2) if report_level else None
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
    (edge_orientation[i] = (row step, column step)), k = -spread_length .. spread_length, zeros outside.
    edge_orientation=None: the (2 spread_length + 1)^2 box maximum of spread_type='all'."""
    ctx = ctx or api.default_ctx()
    breaks = np.ascontiguousarray(edge_feature_row_breaks, dtype=np.int64)
    nb = len(breaks) - 1
    orient = None if edge_orientation is None else \
        np.ascontiguousarray(np.asarray(edge_orientation)[:nb], dtype=np.int32).reshape(nb, 2)
    work, same = _in_place(T)
    if nb > 0 and int(breaks[-1]) > work.shape[0]:
        raise ValueError("edge_feature_row_breaks reach past the %d rows of T" % work.shape[0])
    check(lib().tsr_spread_edgemap(ctx._h, ptr(work), ptr(breaks), nb, work.shape[1], ptr(orient), int(spread_length)))
    if not same:
        T[...] = work


EDGE_ORIENTATIONS = np.array([[1., 0.], [-1., 0.], [0., 1.], [0., -1.], [1., 1.], [-1., -1.], [1., -1.], [-1., 1.]])


def compute_edge_feature_row_breaks(base_height, edge_orientations):
    """:1679-1693: cumulative block heights; orientations without a vertical component are three rows taller."""
    heights = np.zeros(9)
    heights[1:] += base_height * np.ones(8) + 3 * (np.asarray(edge_orientations)[:, 0] == 0)
    return np.cumsum(heights)


def _edge_map_threshold_segments(E, block_length, spread_length, threshold=.7, edge_orientations=EDGE_ORIENTATIONS,
                                 edge_feature_row_breaks=None, abst_threshold=ABST_THRESHOLD, verbose=False,
                                 critical_bands=None, return_thresholds=False, spread_type='unidirectional',
                                 convert_to_binary=True, ctx=None):
    """:1537-1675, in place on E [rows, time]: the trailing partial block first -- thresholded and spread on a copy
    of the LAST block_length columns, only its own columns written back -- then every full block; per block
    threshold_edgemap, then the spread: along each row block's orientation ('unidirectional') or SciPy's
    (2 s + 1)^2 maximum filter with zeros outside the row block ('all'); any other spread_type spreads nothing.
    return_thresholds reproduces the reference's early return (the list is handed back from inside the loop over
    the full blocks, after the first of them; without a full block the function falls off its end)."""
    ctx = ctx or api.default_ctx()
    height, length = E.shape
    if edge_feature_row_breaks is None:
        base_height = (height - 6) // 8
        assert base_height * 8 == height - 6
        edge_feature_row_breaks = compute_edge_feature_row_breaks(base_height, edge_orientations)
    breaks = np.asarray(edge_feature_row_breaks).astype(np.int64)
    block_length = int(block_length)
    view_thresholds = [] if return_thresholds else None

    def process(block):          # block: contiguous f64 [rows, block_length]
        if verbose or return_thresholds:
            _, view = threshold_edgemap(block, threshold, breaks, report_level=True, abst_threshold=abst_threshold,
                                        critical_bands=critical_bands, convert_to_binary=convert_to_binary, ctx=ctx)
            if verbose:
                print(view)
            if return_thresholds:
                view_thresholds.append(view)
        else:
            threshold_edgemap(block, threshold, breaks, report_level=False, abst_threshold=abst_threshold,
                              critical_bands=critical_bands, convert_to_binary=convert_to_binary, ctx=ctx)
        if spread_length > 0 and spread_type == 'unidirectional':
            spread_edgemap(block, breaks, edge_orientations, spread_length=spread_length, ctx=ctx)
        elif spread_length > 0 and spread_type == 'all':
            # (the reference walks i = 0 .. 7 over breaks[i] : breaks[i + 1])
            spread_edgemap(block, breaks[:9], None, spread_length=spread_length, ctx=ctx)

    last = length % block_length
    if last > 0:
        last_E = np.ascontiguousarray(E[:, -block_length:], dtype=np.float64)
        process(last_E)
        E[:, -last:] = last_E[:, -last:]
    for b in range(length // block_length):
        cols = slice(b * block_length, (b + 1) * block_length)
        blk = np.ascontiguousarray(E[:, cols], dtype=np.float64)
        try:
            process(blk)
        finally:
            E[:, cols] = blk
        if return_thresholds:
            return view_thresholds
    return None


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
