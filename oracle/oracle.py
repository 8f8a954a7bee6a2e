"""CPU oracle for the template-scoring path -- TEST INFRASTRUCTURE ONLY.

Python face of oracle/tsr_oracle.c (ctypes) plus NumPy restatements of the
reference's Python-2 glue that cannot be imported under Python 3.  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product path (template-speech-recognition_amd/) never does.

Pinned bit-for-bit against the compiled reference in oracle/_ref (see
oracle/build_ref.py) and the golden vectors under tests/golden/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "tsr_oracle.c")
_LIB = os.path.join(_HERE, "liboracle.so")
_lib = None

_i64, _i32, _int = C.c_int64, C.c_int32, C.c_int
_p = C.c_void_p


# TSR_ORACLE_SANITIZE=1: the same source built with AddressSanitizer + UndefinedBehaviorSanitizer (CPU build
# only; tests/test_sanitizers_cpu.py runs the oracle tests against it in a child process with libasan preloaded)
_SANITIZE = os.environ.get("TSR_ORACLE_SANITIZE") == "1"
if _SANITIZE:
    _LIB = os.path.join(_HERE, "liboracle_san.so")


def build(force=False):
    """gcc -O2 -ffp-contract=off (no -ffast-math): summation order preserved."""
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(_SRC):
        flags = ["-O2"]
        if _SANITIZE:
            flags = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
                     "-fno-sanitize-recover=undefined"]
        subprocess.check_call(["gcc"] + flags + ["-ffp-contract=off", "-fPIC", "-shared",
                               "-fvisibility=hidden", "-o", _LIB, _SRC, "-lm"])
    return _LIB


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
    return _lib


def _ptr(a):
    return a.ctypes.data_as(_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


# ----------------------------------------------------------------------------
# compute_likelihood_linear_filter.pyx
# ----------------------------------------------------------------------------
def detect(F, LF):
    """compute_likelihood_linear_filter.pyx:135-176. F u8 [T,...], LF f32 [L,...]."""
    F = _c(F, np.uint8)
    LF = _c(LF, np.float32)
    T, L = F.shape[0], LF.shape[0]
    D = int(np.prod(F.shape[1:]))
    assert int(np.prod(LF.shape[1:])) == D and T >= L
    out = np.empty(T, np.float32)
    lib().tsr_oracle_detect_u8(_ptr(F), _i64(T), _i64(D), _ptr(LF), _i64(L), _ptr(out))
    return out


def detect_float(F, LF):
    """:178-219 / :281-318 / :322-359 (3-D or 2-D f32 features)."""
    F = _c(F, np.float32)
    LF = _c(LF, np.float32)
    T, L = F.shape[0], LF.shape[0]
    D = int(np.prod(F.shape[1:]))
    assert int(np.prod(LF.shape[1:])) == D and T >= L
    out = np.empty(T, np.float32)
    lib().tsr_oracle_detect_f32(_ptr(F), _i64(T), _i64(D), _ptr(LF), _i64(L), _ptr(out))
    return out


detect_float2d = detect_float
detect_spectral = detect_float


def detect_float_max(F, LF, cs):
    """:221-278. LF f32 [M,L,...], cs f32 [M] -> (f32 [T], u16 [T])."""
    F = _c(F, np.float32)
    LF = _c(LF, np.float32)
    cs = _c(cs, np.float32)
    T, M, L = F.shape[0], LF.shape[0], LF.shape[1]
    D = int(np.prod(F.shape[1:]))
    assert int(np.prod(LF.shape[2:])) == D and T >= L and cs.shape == (M,)
    out = np.empty(T, np.float32)
    ids = np.empty(T, np.uint16)
    lib().tsr_oracle_detect_float_max(_ptr(F), _i64(T), _i64(D), _ptr(LF), _i64(M), _i64(L),
                                      _ptr(cs), _ptr(out), _ptr(ids))
    return out, ids


def get_max_detect_examples(F, LF, lengths):
    """:21-76. F u8 [N,T,...], LF f64 [L,...], lengths i64 [N] -> f64 [N]."""
    F = _c(F, np.uint8)
    LF = _c(LF, np.float64)
    lengths = _c(lengths, np.int64)
    N, T, L = F.shape[0], F.shape[1], LF.shape[0]
    D = int(np.prod(F.shape[2:]))
    out = np.empty(N, np.float64)
    lib().tsr_oracle_get_max_detect_examples(_ptr(F), _i64(N), _i64(T), _i64(D), _ptr(LF),
                                             _i64(L), _ptr(lengths), _ptr(out))
    return out


def pack_bank(linear_filters_cs):
    """((W_m f32 [L_m,...], c_m), ...) -> (W_cat f32, woff i64 [M], L i32 [M], c f64 [M], D)."""
    Ws = [np.ascontiguousarray(w, np.float32) for w, _ in linear_filters_cs]
    D = int(np.prod(Ws[0].shape[1:]))
    L = np.array([w.shape[0] for w in Ws], np.int32)
    woff = np.concatenate(([0], np.cumsum(L.astype(np.int64) * D)[:-1])).astype(np.int64)
    W_cat = np.concatenate([w.reshape(-1) for w in Ws]).astype(np.float32)
    c = np.array([float(cc) for _, cc in linear_filters_cs], np.float64)
    return W_cat, woff, L, c, D


def score_bank(E, linear_filters_cs):
    """get_train_data.py:962-992 epilogue for ONE utterance -> (scores f32 [T], ids i32 [T])."""
    F = _c(E, np.uint8)
    T = F.shape[0]
    W_cat, woff, L, c, D = pack_bank(linear_filters_cs)
    assert int(np.prod(F.shape[1:])) == D
    scores = np.empty(T, np.float32)
    ids = np.empty(T, np.int32)
    scratch = np.empty(T, np.float32)
    lib().tsr_oracle_score_bank_u8(_ptr(F), _i64(T), _i64(D), _ptr(W_cat), _ptr(woff), _ptr(L),
                                   _ptr(c), _i64(len(L)), _ptr(scores), _ptr(ids), _ptr(scratch))
    return scores, ids


def score_bank_all(E, linear_filters_cs):
    """detect(E, W_m) + f32(c_m) for every template -> f32 [M,T]."""
    F = _c(E, np.uint8)
    T = F.shape[0]
    W_cat, woff, L, c, D = pack_bank(linear_filters_cs)
    out = np.empty((len(L), T), np.float32)
    lib().tsr_oracle_score_bank_all_u8(_ptr(F), _i64(T), _i64(D), _ptr(W_cat), _ptr(woff),
                                       _ptr(L), _ptr(c), _i64(len(L)), _ptr(out))
    return out


def score_corpus(feats_cat, frame_offsets, linear_filters_cs):
    """Serial corpus loop (get_train_data.py:1269) -> (scores f32 [sumT], ids i32 [sumT])."""
    feats = _c(feats_cat, np.uint8)
    foff = _c(frame_offsets, np.int64)
    N = len(foff) - 1
    W_cat, woff, L, c, D = pack_bank(linear_filters_cs)
    total = int(foff[-1])
    assert feats.size == total * D
    scores = np.empty(total, np.float32)
    ids = np.empty(total, np.int32)
    scratch = np.empty(int(np.max(np.diff(foff))) if N else 1, np.float32)
    lib().tsr_oracle_score_corpus_u8(_ptr(feats), _ptr(foff), _i64(N), _i64(D), _ptr(W_cat),
                                     _ptr(woff), _ptr(L), _ptr(c), _i64(len(L)), _ptr(scores),
                                     _ptr(ids), _ptr(scratch))
    return scores, ids


# ----------------------------------------------------------------------------
# code_parts.pyx / spread_waliji_patches.pyx
# ----------------------------------------------------------------------------
def _code_parts(fn, X, log_parts, log_invparts, threshold):
    X = _c(X, np.uint8)
    lp = _c(log_parts, np.float64)
    lip = _c(log_invparts, np.float64)
    X0, X1, E = X.shape
    K, P0, P1, E2 = lp.shape
    assert E2 == E and lip.shape == lp.shape and X0 >= P0 and X1 >= P1
    out = np.empty((X0 - P0 + 1, X1 - P1 + 1, K + 1), np.float64)
    fn(_ptr(X), _i64(X0), _i64(X1), _i64(E), _ptr(lp), _ptr(lip), _i64(K), _i64(P0), _i64(P1),
       _int(int(threshold)), _ptr(out))
    return out


def code_parts(X, log_parts, log_invparts, threshold):
    """code_parts.pyx:70-135."""
    return _code_parts(lib().tsr_oracle_code_parts, X, log_parts, log_invparts, threshold)


def code_parts_fast(X, log_parts, log_invparts, threshold):
    """code_parts.pyx:247-295."""
    return _code_parts(lib().tsr_oracle_code_parts_fast, X, log_parts, log_invparts, threshold)


def argmax_last(A):
    """np.argmax(A, -1) (first maximum), get_train_data.py:2621, :184."""
    A = _c(A, np.float64)
    out = np.empty(A.shape[:-1], np.int64)
    lib().tsr_oracle_argmax_last(_ptr(A), _i64(out.size), _i64(A.shape[-1]), _ptr(out))
    return out


def spread_patches(X, spread_0_dim, spread_1_dim, num_parts):
    """code_parts.pyx:406-457 == spread_waliji_patches.pyx:15-59."""
    X = _c(X, np.int64)
    out = np.empty(X.shape + (int(num_parts),), np.uint8)
    lib().tsr_oracle_spread_patches(_ptr(X), _i64(X.shape[0]), _i64(X.shape[1]),
                                    _int(int(spread_0_dim)), _int(int(spread_1_dim)),
                                    _i64(int(num_parts)), _ptr(out))
    return out


spread_waliji_patches = spread_patches


def spread_graph_patches(X, part_graph):
    """code_parts.pyx:459-503; mutates X in place, returns None."""
    assert X.dtype == np.uint8 and X.flags.c_contiguous
    g = _c(part_graph, np.uint8)
    lib().tsr_oracle_spread_graph_patches(_ptr(X), _i64(X.shape[0]), _i64(X.shape[1]),
                                          _i64(X.shape[2]), _ptr(g))


# ----------------------------------------------------------------------------
# NumPy restatements of the Python-2 glue (cannot be imported under Python 3)
# ----------------------------------------------------------------------------
def score_template_background_section(template, bgd, E):
    """test_template.py:25-41 with front/back pad 0: (sum E*log W, C), f64.
    template, E: [D, L] (features x time); bgd: [D]."""
    assert E.shape[1] == template.shape[1]
    T = np.asarray(template, np.float64)
    U_bgd = np.tile(np.asarray(bgd, np.float64), (T.shape[1], 1)).transpose()
    C_exp_inv_long = (1. - T) / (1. - U_bgd)
    Cc = np.log(C_exp_inv_long).sum()
    expW = (T / U_bgd) / C_exp_inv_long
    return (E * np.log(expW)).sum(), Cc


def construct_linear_filter(T, bgd, min_prob=.01, all_cs=False):
    """estimate_template.py:230-251: W = log((T/B)/((1-T)/(1-B))) -> f32, c f64."""
    T = np.clip(np.asarray(T, np.float64), min_prob, 1 - min_prob)
    Bgd = np.tile(np.asarray(bgd, np.float64), (T.shape[0],) + (1,) * (T.ndim - 1))
    C_exp_inv = (1. - T) / (1. - Bgd)
    if all_cs:
        c = np.cumsum(np.log(C_exp_inv.reshape(len(C_exp_inv), -1)).sum(1))
    else:
        c = np.log(C_exp_inv).sum()
    expW = (T / Bgd) / C_exp_inv
    return np.log(expW).astype(np.float32), c


def construct_linear_filters(Ts, bgd, min_prob=.01, all_cs=False):
    """estimate_template.py:156-172 (non-spectral branch)."""
    return tuple(construct_linear_filter(np.clip(T, min_prob, 1 - min_prob),
                                         np.clip(bgd, min_prob, 1 - min_prob), all_cs=all_cs)
                 for T in Ts)


def reorg_part_for_fast_filtering(part, feature_types=8):
    """get_train_data.py:41-51: [feature_types*H, T] -> [T, H, feature_types]."""
    H = part.shape[0] // feature_types
    return np.array([part[i * H:(i + 1) * H].T for i in range(feature_types)]).swapaxes(0, 1).swapaxes(1, 2)


def collapse_to_grid(E_coded, grid_time, grid_frequency):
    """get_train_data.py:72-81."""
    return E_coded[::grid_time, ::grid_frequency]


def get_part_features(E, logParts, logInvParts, bernsteinEdgeThreshold, spreadRadiusX,
                      spreadRadiusY, numParts, partGraph=None):
    """get_train_data.py:2619-2631 (the non-mag/aux branch)."""
    out = code_parts(np.asarray(E).astype(np.uint8), logParts, logInvParts, bernsteinEdgeThreshold)
    max_responses = argmax_last(out)
    spread = spread_patches(max_responses, spreadRadiusX, spreadRadiusY, numParts)
    if partGraph is not None:
        spread_graph_patches(spread, partGraph)
    return spread


def waliji_feature_map(E2, log_part_blocks, log_invpart_blocks):
    """get_train_data.py:183-191: code_parts_fast(thr 10) -> argmax -> spread (radius =
    patch dims) -> collapse_to_grid(patch dims).  E2 is the [T,F,8] edge map."""
    F = code_parts_fast(np.asarray(E2).astype(np.uint8), log_part_blocks, log_invpart_blocks, 10)
    F = argmax_last(F)
    F = spread_patches(F, log_part_blocks.shape[1], log_part_blocks.shape[2],
                       log_part_blocks.shape[0])
    return collapse_to_grid(F, log_part_blocks.shape[1], log_part_blocks.shape[2])


def compute_detection_E(E, linear_filters_cs):
    """get_train_data.py:962-992 written with NumPy on top of detect(): used to pin
    tsr_oracle_score_bank_u8 (f32 + f32(c), strict '>' ids, np.maximum)."""
    E = np.asarray(E).astype(np.uint8)
    row = detect(E, linear_filters_cs[0][0]) + np.float32(linear_filters_cs[0][1])
    ids = np.zeros(E.shape[0], np.int32)
    for m, (W, c) in enumerate(linear_filters_cs[1:], start=1):
        v = detect(E, W) + np.float32(c)
        ids[v > row] = m
        row = np.maximum(v, row)
    return row.astype(np.float32), ids


# ---- classification over phone windows, restated from text --------------------
# The callers below are Python-2 modules that cannot be imported; they are restated line for line
# with the NumPy-1.x promotion (f32 array + f64 scalar -> f32) made explicit.  `_detect` /
# `_detect_float` default to this oracle's C restatements and are swapped for the compiled
# reference's functions when the golden vectors are generated.
def get_classify_windows(phns, flts):
    """get_train_data.py:710-716."""
    flts = np.asarray(flts)
    window_starts = np.maximum(0, flts[:-1] + (flts[:-1] - flts[1:]) / 3.).astype(np.int16)
    window_ends = np.minimum(flts[-1], flts[:-1] + (flts[1:] - flts[:-1]) / 3.).astype(np.int16)
    return window_starts, window_ends


def compute_classification_E(E, phns, E_flts, utt_id, classify_array, classify_template_ids,
                             classify_template_lengths, classify_locs, linear_filters_cs,
                             _detect=None):
    """get_train_data.py:718-773 (_compute_classification_E): per phone window the first maximum
    of detect + c per template, then the best template (strict '<': the earlier one on ties)."""
    _detect = _detect or detect
    window_starts, window_ends = get_classify_windows(phns, E_flts)
    E8 = np.asarray(E).astype(np.uint8)
    for filter_id, (cur_filt, cur_c) in enumerate(linear_filters_cs):
        detection_scores = _detect(E8, cur_filt) + np.float32(cur_c)
        for phn_id, (w_start, w_end) in enumerate(zip(window_starts, window_ends)):
            w_start, w_end = int(w_start), int(w_end)
            t = w_start + int(np.argmax(detection_scores[w_start:w_end]))
            if filter_id == 0 or classify_array[utt_id, phn_id] < detection_scores[t]:
                classify_locs[utt_id, phn_id] = t
                classify_template_ids[utt_id, phn_id] = filter_id
                classify_template_lengths[utt_id, phn_id] = len(cur_filt)
                classify_array[utt_id, phn_id] = detection_scores[t]


def get_isolated_classify_windows(E, phns, flts, bgd, linear_filters_cs, max_length=None):
    """get_train_data.py:781-835.  NOTE: like the reference, shifts `flts` IN PLACE by the front
    padding.  Returns (E_windows, window_starts, window_ends, flts[0])."""
    if max_length is None:
        max_filter_length = max(len(lfc[0]) for lfc in linear_filters_cs)
    else:
        max_filter_length = max_length
    max_phn_length_third = int(np.ceil(np.max(flts[1:] - flts[:-1]) / 3.))
    ones = (1,) * len(bgd.shape)
    E = np.vstack((
        np.tile(bgd, (max_filter_length + max_phn_length_third,) + ones).astype(np.float32),
        E.astype(np.float32),
        np.tile(bgd, (max(0, int(flts[-2]) + max_phn_length_third + max_filter_length - int(flts[-1])),)
                + ones).astype(np.float32)))
    flts += max_filter_length + max_phn_length_third
    window_starts = (flts[:-1] + (flts[:-1] - flts[1:]) / 3.).astype(np.int16)
    window_ends = (flts[:-1] + (flts[1:] - flts[:-1]) / 3.).astype(np.int16)
    phn_segment_ends = (flts[1:] + (flts[1:] - flts[:-1]) / 3.).astype(np.int16)
    E_windows = tuple(
        np.vstack((E[int(ws):int(pse)].astype(np.float32),
                   np.tile(bgd, (max(0, int(we) + max_filter_length - int(pse)),) + ones)))
        for ws, we, pse in zip(window_starts, window_ends, phn_segment_ends))
    return E_windows, window_starts, window_ends, flts[0]


def compute_isolated_classification_E(E, phns, E_flts, utt_id, classify_array, classify_template_ids,
                                      classify_template_lengths, classify_locs, linear_filters_cs, bgd,
                                      _detect_float=None):
    """get_train_data.py:838-925 without the optional SVM stage: every phone gets its own
    background-padded f32 window, scored with detect_float (detect_float2d for 2-d features) + c;
    the arg-max runs over the first w_end - w_start positions only."""
    _detect_float = _detect_float or detect_float
    E_windows, window_starts, window_ends, flt_front_pad = get_isolated_classify_windows(
        E, phns, E_flts, bgd, linear_filters_cs)
    for phn_id, (E_window, w_start, w_end) in enumerate(zip(E_windows, window_starts, window_ends)):
        w_start, w_end = int(w_start), int(w_end)
        for filter_id, (cur_filt, cur_c) in enumerate(linear_filters_cs):
            detection_scores = _detect_float(E_window, cur_filt) + np.float32(cur_c)
            t = w_start + int(np.argmax(detection_scores[:w_end - w_start])) - int(flt_front_pad)
            v = detection_scores[t - w_start + int(flt_front_pad)]
            if filter_id == 0 or classify_array[utt_id, phn_id] < v:
                classify_locs[utt_id, phn_id] = t
                classify_template_ids[utt_id, phn_id] = filter_id
                classify_template_lengths[utt_id, phn_id] = len(cur_filt)
                classify_array[utt_id, phn_id] = v


# ---- peak picking, restated from text: roc_functions.py is a Python-2 module with
# no native part.  The restatements are checked against the reference's own functions, executed where they lie
# through oracle/py2ref.py (tests/test_reference_text.py); golden g19 holds that code's outputs.
def get_max_above_threshold(detection_vals, detection_length, threshold, C0):
    """roc_functions.py:177-187."""
    l = detection_length - C0 + 1
    v = np.asarray(detection_vals)
    if l - 1 <= 1:
        return np.zeros(0, np.uint16)
    return np.arange(1, l - 1, dtype=np.uint16)[(v[1:l - 1] >= v[:l - 2]) * (v[1:l - 1] >= v[2:l])
                                                * (v[1:l - 1] >= np.float32(threshold))]


def get_max_detection_in_syllable_windows(detection_array, example_start_end_times, detection_lengths,
                                          window_start, window_end, return_argsort_idx=False):
    """roc_functions.py:88-139."""
    max_detect_vals, utterance_ids = [], []
    for example_idx, start_end_times in enumerate(example_start_end_times):
        for start_time, end_time in start_end_times:
            detect_vals = detection_array[example_idx, max(start_time + window_start, 0):
                                          min(start_time + window_end, detection_array.shape[1])]
            maybe = detect_vals[1:-1][(detect_vals[1:-1] >= detect_vals[:-2]) * (detect_vals[1:-1] >= detect_vals[2:])]
            max_detect_vals.append(maybe.max() if len(maybe) > 0 else -np.inf)
            utterance_ids.append(example_idx)
    if return_argsort_idx:
        return np.sort(max_detect_vals), np.argsort(max_detect_vals), np.array(utterance_ids)
    return np.sort(max_detect_vals)


# ---- edge front-end, restated from text: edge_signal_proc.py is a Python-2 module
# without a native part.  Checked against the reference's own functions run through oracle/py2ref.py
# (tests/test_reference_text.py: identical maps); golden g18 holds that code's outputs.
EDGE_ORIENTATIONS = np.array([[0, 1], [0, -1], [1, 0], [-1, 0], [1, 1], [-1, -1], [-1, 1], [1, -1]])
ABST_THRESHOLD = np.array([0.025, 0.025, 0.015, 0.015, 0.02, 0.02, 0.02, 0.02])


def edge_map_no_threshold(S):
    """edge_signal_proc.py:1697-1836 for S [H, T] (frequency x time) -> E f64 [8H, T]: per orientation the
    first difference along it, kept where it is a local maximum (>= both neighbours along the same
    orientation) and of the orientation's sign."""
    S = np.asarray(S, np.float64)
    H, T = S.shape
    E = np.zeros((8 * H, T))

    def keep(d, lo, hi, sign):            # d, lo, hi: equally shaped difference arrays
        d = sign * d
        return d * ((d >= sign * lo) * (d >= sign * hi) * (d >= 0))

    if T >= 4:
        d = S[:, 1:] - S[:, :-1]          # along time
        for e, sg in ((0, 1.0), (1, -1.0)):
            E[e * H:(e + 1) * H, 1:-2] = keep(d[:, 1:-1], d[:, :-2], d[:, 2:], sg)
    if H >= 4:
        d = S[1:, :] - S[:-1, :]          # along frequency
        for e, sg in ((2, 1.0), (3, -1.0)):
            E[e * H + 1:(e + 1) * H - 2, :] = keep(d[1:-1, :], d[:-2, :], d[2:, :], sg)
    if H >= 4 and T >= 4:
        d = S[1:, 1:] - S[:-1, :-1]       # diagonal
        for e, sg in ((4, 1.0), (5, -1.0)):
            E[e * H + 1:(e + 1) * H - 2, 1:-2] = keep(d[1:-1, 1:-1], d[:-2, :-2], d[2:, 2:], sg)
        d = S[:-1, 1:] - S[1:, :-1]       # anti-diagonal
        for e, sg in ((6, 1.0), (7, -1.0)):
            E[e * H + 1:(e + 1) * H - 2, 1:-2] = keep(d[1:-1, 1:-1], d[2:, :-2], d[:-2, 2:], sg)
    return E


def _threshold_edge_block(B, quantile_level, abst):
    """edge_signal_proc.py:1257-1307 (convert_to_binary=True), in place on the view B."""
    pos = B > 0
    vals = np.sort(B[pos].ravel())
    if vals.shape[0] <= 0:
        B[:] = 0
        return
    tau = vals[int(quantile_level * vals.shape[0])]
    B[:] = (pos & (B >= max(tau, abst))).astype(np.float64)


def _spread_edges(B, direction, spread_length):
    """edge_signal_proc.py:1406-1418: max over shifts k*direction, k in [-s, s], zeros outside."""
    F, T = B.shape
    s = spread_length
    Z = np.zeros((F + 2 * s, T + 2 * s))
    Z[s:s + F, s:s + T] = B
    out = np.zeros(B.shape)
    for k in range(-s, s + 1):
        out = np.maximum(out, Z[s + k * direction[0]:s + F + k * direction[0],
                               s + k * direction[1]:s + T + k * direction[1]])
    return out


def edge_map_threshold_segments(E, block_length, spread_length, threshold=.7, abst_threshold=ABST_THRESHOLD):
    """edge_signal_proc.py:1537-1675 (unidirectional spread, binary output), in place on E [8H, T]: the
    trailing partial block is thresholded and spread on a copy of the LAST block_length columns and only its
    own columns are written back; then every full block in turn."""
    H8, T = E.shape
    H = H8 // 8
    num_full, last = T // block_length, T % block_length

    def process(Bk):
        for e in range(8):
            sub = Bk[e * H:(e + 1) * H]
            _threshold_edge_block(sub, threshold, abst_threshold[e])
            if spread_length > 0:
                sub[:] = _spread_edges(sub, EDGE_ORIENTATIONS[e], spread_length)

    if last > 0:
        last_E = E[:, -block_length:].copy()
        process(last_E)
        E[:, -last:] = last_E[:, -last:]
    for b in range(num_full):
        process(E[:, b * block_length:(b + 1) * block_length])
    return E


def get_edge_features(S, block_length=40, spread_length=1, threshold=.7, abst_threshold=ABST_THRESHOLD):
    """get_train_data.py:2547-2557 (default branch): S [H, T] -> binary edge map uint8 [T, H, 8]."""
    E = edge_map_no_threshold(S)
    edge_map_threshold_segments(E, block_length, spread_length, threshold, abst_threshold)
    return reorg_part_for_fast_filtering(E).astype(np.uint8)


# ---- the rest of SURVEY 8(f) N3: whole-array thresholding / spreading and the magnitude branches ----------
def threshold_edgemap(E, quantile_level, edge_feature_row_breaks, abst_threshold=ABST_THRESHOLD, convert_to_binary=True):
    """edge_signal_proc.py:1184-1252 with critical_bands=None: every row block [breaks[i-1], breaks[i]) of E
    (all columns) through threshold_edge_block (:1257-1307), in place.  Returns the blocks' quantile values
    (what report_level=True hands back; nan for a block without positive values)."""
    taus = np.full(len(edge_feature_row_breaks) - 1, np.nan)
    for i in range(1, len(edge_feature_row_breaks)):
        B = E[int(edge_feature_row_breaks[i - 1]):int(edge_feature_row_breaks[i]), :]
        pos = B > 0
        vals = np.sort(B[pos].ravel())
        if vals.shape[0] <= 0:
            B[:] = 0
            continue
        tau = vals[int(quantile_level * vals.shape[0])]
        taus[i - 1] = tau
        sig = pos & (B >= max(tau, abst_threshold[i - 1]))
        B[:] = np.where(sig, 1.0 if convert_to_binary else B, 0.0)
    return taus


def spread_edgemap(T, edge_feature_row_breaks, edge_orientation, spread_length=3):
    """edge_signal_proc.py:1397-1404: _spread_edges on every row block along its own orientation, in place."""
    for i in range(1, len(edge_feature_row_breaks)):
        a, b = int(edge_feature_row_breaks[i - 1]), int(edge_feature_row_breaks[i])
        T[a:b, :] = _spread_edges(T[a:b], [int(x) for x in edge_orientation[i - 1]], spread_length)


def _max_filter_freq(S, size):
    """scipy.ndimage.maximum_filter(S, (size, 1)) (mode 'reflect', origin 0) without SciPy: output row i is the
    maximum of rows i - size//2 ... i - size//2 + size - 1, indices reflected at the ends (d c b a | a b c d | ...)."""
    H = S.shape[0]
    lo = size // 2
    idx = np.arange(-lo, H - lo + size - 1)
    period = 2 * H
    idx = np.mod(idx, period)
    idx = np.where(idx >= H, period - 1 - idx, idx)
    P = S[idx]
    return np.max(np.stack([P[j:j + H] for j in range(size)]), axis=0)


def _magnitude_windows(num_frames, block_length):
    """The (estimation window, frames it sets, nominal window length of the rank rule) triples of
    edge_signal_proc.py:648-666 / :720-738."""
    bq = block_length // 4
    wins = [(0, min(3 * bq, num_frames), 0, min(2 * bq, num_frames), None)]   # first: rank int(q * 3 bq [* F] + .5)
    nxt = 2 * bq
    while nxt < num_frames:
        a = nxt - bq
        b = min(num_frames, a + 4 * bq)
        last = min(nxt + 2 * bq, num_frames)
        wins.append((a, b, nxt, last, b - a))
        nxt = last
    return bq, wins


def magnitude_features(S, block_length, spread_radius, threshold_quantile, mag_smooth_freq, mag_downsample_freq,
                       mag_spread_length=1):
    """edge_signal_proc.py:600-670: S [F, T] -> uint8 [T, F'], per frequency bin the frames at or above the
    bin's own quantile inside overlapping windows, then an OR over +-mag_spread_length frames."""
    S = np.asarray(S, np.float64)
    if mag_smooth_freq > 0:
        S = _max_filter_freq(S, mag_smooth_freq)
    if mag_downsample_freq > 0:
        S = S[::mag_downsample_freq]
    S = S.T
    T = S.shape[0]
    E = np.zeros(S.shape, np.uint8)
    bq, wins = _magnitude_windows(T, block_length)
    for a, b, s0, s1, n in wins:
        block = np.sort(S[a:b], axis=0)
        k = int(threshold_quantile * 3 * bq + .5) if n is None else int(threshold_quantile * n)
        E[s0:s1] = S[s0:s1] >= block[k]
    m = mag_spread_length
    if m > 0:
        Z = np.zeros((T + 2 * m, E.shape[1]), np.uint8)
        Z[m:m + T] = E
        E = np.max(np.stack([Z[j:j + T] for j in range(2 * m + 1)]), axis=0).astype(np.uint8)
    return E


def whole_block_spread_min_count(spread_radius):
    """magnitude_features_whole_block's spread (:741-744) is correlate(E, w) >= 1 with w = 1/r everywhere and 1
    in the centre, E uint8: SciPy accumulates the products in double and casts the sum to uint8 (truncation), so
    a frame/bin with a clear centre is set iff the n-fold repeated addition of fl(1/r) reaches 1.0.  Returns the
    smallest such n (the sum is non-decreasing in n)."""
    x, s, n = 1.0 / spread_radius, 0.0, 0
    while s < 1.0:
        s += x
        n += 1
    return n


def magnitude_features_whole_block(S, block_length, spread_radius, threshold_quantile, mag_smooth_freq,
                                   mag_downsample_freq):
    """edge_signal_proc.py:673-747: as magnitude_features with ONE quantile per window over all bins, and a
    square spread: a cell is set if it was set or at least `whole_block_spread_min_count` of the other cells of
    its (2r+1)^2 neighbourhood were."""
    S = np.asarray(S, np.float64)
    if mag_smooth_freq > 0:
        S = _max_filter_freq(S, mag_smooth_freq)
    if mag_downsample_freq > 0:
        S = S[::mag_downsample_freq]
    S = S.T
    T, F = S.shape
    E = np.zeros(S.shape, np.uint8)
    bq, wins = _magnitude_windows(T, block_length)
    for a, b, s0, s1, n in wins:
        block = np.sort(S[a:b].ravel())
        k = int(threshold_quantile * 3 * bq * F + .5) if n is None else int(threshold_quantile * n * F)
        E[s0:s1] = S[s0:s1] >= block[k]
    r = spread_radius
    if r > 0:
        Z = np.zeros((T + 2 * r, F + 2 * r), np.int64)
        Z[r:r + T, r:r + F] = E
        box = sum(Z[i:i + T, j:j + F] for i in range(2 * r + 1) for j in range(2 * r + 1))
        E = ((E > 0) | (box - E >= whole_block_spread_min_count(r))).astype(np.uint8)
    return E.astype(np.uint8)


def get_edge_features_with_parameters(S, p):
    """get_train_data.py:2489-2557 with its branches; `p` carries the EdgemapParameters fields.  The
    `magnitude_features` branch is dead in the reference (it calls esp.magnitude_features with one positional
    argument too few, :2494-2498) and is not restated."""
    S = np.asarray(S, np.float64)
    if p.auxiliary_data:
        A, S = S[-3:], S[:-3]
    if p.magnitude_and_edge_features:
        E2 = magnitude_features_whole_block(S.copy(), p.magnitude_block_length, p.magnitude_spread,
                                            p.magnitude_threshold, p.mag_smooth_freq, p.mag_downsample_freq)
        E = edge_map_no_threshold(S)
        edge_map_threshold_segments(E, p.block_length, p.spread_length, p.threshold)   # (default abst_threshold, :2509-2515)
        if not p.auxiliary_data:
            return np.hstack((E.T, E2))
        AB = magnitude_features(A, p.magnitude_block_length, 0, p.auxiliary_threshold, 0, 0, mag_spread_length=1)
        return np.hstack((E.T, E2, AB))
    if p.magnitude_thresholded_edges:
        E2 = magnitude_features_whole_block(S.copy(), p.magnitude_block_length, p.magnitude_spread,
                                            p.magnitude_threshold, p.mag_smooth_freq, p.mag_downsample_freq)
        E = edge_map_no_threshold(S)
        E2 = E2[::-1].T                                  # (the reference reverses the TIME axis here, :2537)
        for i in range(8):
            E[E2.shape[0] * i:E2.shape[0] * (i + 1)] *= E2
        edge_map_threshold_segments(E, p.block_length, p.spread_length, p.threshold, p.abst_threshold)
        return reorg_part_for_fast_filtering(E)
    E = edge_map_no_threshold(S)
    edge_map_threshold_segments(E, p.block_length, p.spread_length, p.threshold, p.abst_threshold)
    return reorg_part_for_fast_filtering(E)


GAUSSIAN5BY5 = np.array([[0.19785663, 0.36326131, 0.44481078, 0.36326131, 0.19785663],
                         [0.36326131, 0.66694134, 0.81666476, 0.66694134, 0.36326131],
                         [0.44481078, 0.81666476, 1., 0.81666476, 0.44481078],
                         [0.36326131, 0.66694134, 0.81666476, 0.66694134, 0.36326131],
                         [0.19785663, 0.36326131, 0.44481078, 0.36326131, 0.19785663]], dtype=np.float32)


def gaussian_edge_count(E_edges):
    """get_train_data.py:2592-2598: scipy.signal.convolve(E_edges, tile(gaussian5by5, 8 edges), 'same')[:, :, 4]:
    with the same 5x5 weights for every edge type this is the Gaussian-weighted number of edges around (t, f),
    zeros outside.  f64 sums here; the reference's are float32 (direct) or FFT-based, equal to ~1e-6."""
    cnt = np.asarray(E_edges, np.float64).sum(2)
    T, F = cnt.shape
    Z = np.zeros((T + 4, F + 4))
    Z[2:2 + T, 2:2 + F] = cnt
    g = GAUSSIAN5BY5.astype(np.float64)
    return sum(g[i, j] * Z[4 - i:4 - i + T, 4 - j:4 - j + F] for i in range(5) for j in range(5))


def get_part_features_mag_aux(E, p):
    """get_train_data.py:2577-2614, the mag_data and auxiliary_data branch: E [T, 8 F + mags + aux] ->
    spread part map [T, F - 4, numParts] (zero rows appended up to T frames)."""
    E = np.asarray(E).T.astype(np.uint8)
    nm, na = p.num_mag_channels, p.num_auxiliary_data
    E_edges, E_mags, E_aux = E[:-nm - na].T, E[-nm - na:-na].T, E[-na:].T
    F = E_edges.shape[1] // 8
    E_edges = E_edges.reshape(len(E_edges), F, 8)
    mask = (gaussian_edge_count(E_edges) > 12).astype(np.uint8)
    out = code_parts_mag_aux(E_edges, E_mags, E_aux, p.logParts, p.logInvParts, p.logMagFeatures, p.logInvMagFeatures,
                             p.logAuxiliaryFeatures, p.logInvAuxiliaryFeatures, mask)
    spread = spread_patches(argmax_last(out), p.spreadRadiusX, p.spreadRadiusY, p.numParts)
    if len(spread) < E.shape[1]:
        spread = np.vstack((spread, np.zeros((E.shape[1] - len(spread),) + spread.shape[1:])))
    return spread


# ---- Bernoulli mixture E-step, restated from text in f64 (the reference evaluates it with
# a float32 BLAS product whose summation order is not defined: checks are tolerance-based; golden g21 holds what
# the reference's own class computes, run through oracle/py2ref.py).
def template_loglikelihoods(data_mat, log_templates, log_invtemplates):
    """bernoulli_mixture.py:287-291 in float64."""
    X = np.asarray(data_mat, np.float64)
    return X @ np.asarray(log_templates, np.float64).T + (1 - X) @ np.asarray(log_invtemplates, np.float64).T


def mixture_e_step(data_mat, log_templates, log_invtemplates, weights):
    """bernoulli_mixture.py:263-280 (soft assignment) -> (loglikelihood, affinities f64)."""
    ll = template_loglikelihoods(data_mat, log_templates, log_invtemplates) + np.log(np.asarray(weights, np.float64))
    mx = ll.max(1)
    adj = np.exp(ll - mx[:, None]).sum(1)
    aff = np.exp(ll - (np.log(adj) + mx)[:, None])
    return float(np.log(adj).sum() + mx.sum()), aff / aff.sum(1, keepdims=True)


# ---- detection clustering (second half).  cluster_times.pyx IS native code in the reference:
# the C restatement above is pinned against the compiled module (oracle/_ref, golden g14).
def cluster_times(times, C0, C1):
    """cluster_times.pyx:133-222 -> partition uint8 [len(times) + 1]."""
    t = _c(times, np.uint16)
    out = np.empty(t.shape[0] + 1, np.uint8)
    lib().tsr_oracle_cluster_times(_ptr(t), _i64(t.shape[0]), C.c_float(C0), C.c_float(C1), _ptr(out))
    return out


def cluster_times_template_lengths(times, template_lengths, template_components):
    """cluster_times.pyx:221-310 -> partition uint8 [len(times) + 1]."""
    t, tl, tc = _c(times, np.uint16), _c(template_lengths, np.uint16), _c(template_components, np.uint16)
    out = np.empty(t.shape[0] + 1, np.uint8)
    lib().tsr_oracle_cluster_times_template_lengths(_ptr(t), _ptr(tl), _ptr(tc), _i64(t.shape[0]), _ptr(out))
    return out


def cluster_times_variable_length(times, detection_lengths):
    """cluster_times.pyx:73-131 -> partition uint8 [len(times)]."""
    t, dl = _c(times, np.int32), _c(detection_lengths, np.int32)
    out = np.empty(t.shape[0], np.uint8)
    lib().tsr_oracle_cluster_times_variable_length(_ptr(t), _ptr(dl), _i64(t.shape[0]), _ptr(out))
    return out


def get_cluster_starts_ends(potential_detect_times, cluster_partition):
    """roc_functions.py:189-210."""
    d, t, tuples = len(potential_detect_times), 0, []
    while t < d:
        start = potential_detect_times[t]
        t += 1
        while cluster_partition[t] != 2:
            t += 1
        tuples.append((start, potential_detect_times[t - 1] + 1))
    return tuple(tuples)


def get_detect_clusters_single_threshold(threshold, detection_array, detection_lengths, C0, C1, _cluster_times=None):
    """roc_functions.py:356-383: per utterance, peaks above the threshold -> clustered (start, end) tuples."""
    ct = _cluster_times or cluster_times
    out = []
    for vals, length in zip(detection_array, detection_lengths):
        times = get_max_above_threshold(vals, int(length), threshold, C0)
        out.append(get_cluster_starts_ends(times, ct(times, C0, C1)))
    return tuple(out)


def code_parts_mag_aux(X_edges, X_mags, X_aux, log_parts, log_invparts, log_mag_parts, log_mag_invparts,
                       log_aux_parts, log_aux_invparts, X_thresh_mask):
    """code_parts.pyx:138-242 (Python-2 integer division); pinned against the reference compiled with
    language_level=2 (oracle/_ref/py2, golden g15)."""
    Xe, Xm, Xa = _c(X_edges, np.uint8), _c(X_mags, np.uint8), _c(X_aux, np.uint8)
    mk = _c(X_thresh_mask, np.uint8)
    arrs = [_c(a, np.float64) for a in (log_parts, log_invparts, log_mag_parts, log_mag_invparts,
                                        log_aux_parts, log_aux_invparts)]
    K, P0, P1, E = arrs[0].shape
    X0, X1 = Xe.shape[0], Xe.shape[1]
    MD, AD = Xm.shape[1], Xa.shape[1]
    assert arrs[2].shape == (K, P0 + MD) and arrs[4].shape == (K, P0 * AD) and mk.shape == (X0, X1)
    out = np.empty((X0 - P0 + 1, X1 - P1 + 1, K + 1), np.float64)
    lib().tsr_oracle_code_parts_mag_aux(_ptr(Xe), _ptr(Xm), _ptr(Xa), _i64(X0), _i64(X1), _i64(E), _i64(MD), _i64(AD),
                                        *[_ptr(a) for a in arrs], _i64(K), _i64(P0), _i64(P1), _ptr(mk), _ptr(out))
    return out
