"""Per-utterance glue of the reference's get_train_data.py that sits directly on
the native kernels, re-hosted on the GPU path:

    reorg_part_for_fast_filtering / reorg_parts_for_fast_filtering   :41-57
    collapse_to_grid                                                  :72-81
    get_waliji_feature_map (from the edge map on)                     :182-191
    get_edge_features (default branch)                                :2489-2557
    get_classify_windows / _compute_classification_E                  :710-773
    get_isolated_classify_windows / _compute_isolated_classification_E :781-925
    _compute_detection_E (scoring + template ids)                     :927-1012
    get_detection_scores_mixture_named_params (the corpus loop)       :1238-1339
    get_part_features (edge-only branch)                              :2576-2631

Waveform -> spectrogram -> edge map (edge_signal_proc.py) is upstream of the
path and stays with the caller: every function here starts from the binary
edge map E.
"""
import hashlib

import numpy as np

from tsr_hip import api
from . import code_parts as cp
from . import compute_likelihood_linear_filter
from . import spread_waliji_patches as swp
from . import edge_signal_proc as esp


def reorg_part_for_fast_filtering(part, feature_types=8):
    """[feature_types*H, T] (edge types stacked vertically, features x time) ->
    [T, H, feature_types] (time x frequency x edge type), the kernels' layout."""
    part = np.asarray(part)
    H = part.shape[0] // feature_types
    return part[:H * feature_types].reshape(feature_types, H, part.shape[1]).transpose(2, 1, 0)


def reorg_parts_for_fast_filtering(parts, feature_types=8, min_prob=.01):
    filters = np.clip(np.array([reorg_part_for_fast_filtering(p, feature_types=feature_types)
                                for p in parts]), min_prob, 1 - min_prob)
    return np.log(filters).astype(np.float32), np.log(1 - filters).astype(np.float32)


def collapse_to_grid(E_coded, grid_time, grid_frequency):
    return E_coded[::grid_time, ::grid_frequency]


_parts_cache = {}


def _digest(a):
    a = np.ascontiguousarray(a)
    return hashlib.blake2b(a.tobytes(), digest_size=16).digest(), a.shape, a.dtype.str


def _device_parts(log_parts, log_invparts):
    # keyed by CONTENT: part re-estimation loops update logParts / logInvParts in place
    key = (_digest(log_parts), _digest(log_invparts))
    hit = _parts_cache.get(key)
    if hit is None:
        if len(_parts_cache) > 8:
            _parts_cache.clear()
        hit = api.Parts(np.asarray(log_parts, np.float64), np.asarray(log_invparts, np.float64))
        _parts_cache[key] = hit
    return hit


def get_part_features(E, parameters, verbose=False):
    """Edge map E [T, F, 8] -> spread part map uint8 [T-P0+1, F-P1+1, numParts]:
    code_parts -> argmax -> spread_patches -> optional part-graph closure, fused on
    the GPU.  `parameters` is duck-typed like get_train_data.PartsParameters
    (logParts, logInvParts, bernsteinEdgeThreshold, spreadRadiusX, spreadRadiusY,
    numParts, partGraph; :2203-2473)."""
    if getattr(parameters, "mag_data", False) and getattr(parameters, "auxiliary_data", False):
        return _get_part_features_mag_aux(E, parameters)
    parts = _device_parts(parameters.logParts, parameters.logInvParts)
    if int(parameters.numParts) != parts.K:
        raise ValueError("numParts=%d but logParts holds %d parts" % (parameters.numParts, parts.K))
    return parts.features(np.asarray(E).astype(np.uint8), parameters.bernsteinEdgeThreshold,
                          spread=(parameters.spreadRadiusX, parameters.spreadRadiusY),
                          graph=getattr(parameters, "partGraph", None))


gaussian5by5 = np.array([[0.19785663, 0.36326131, 0.44481078, 0.36326131, 0.19785663],
                         [0.36326131, 0.66694134, 0.81666476, 0.66694134, 0.36326131],
                         [0.44481078, 0.81666476, 1., 0.81666476, 0.44481078],
                         [0.36326131, 0.66694134, 0.81666476, 0.66694134, 0.36326131],
                         [0.19785663, 0.36326131, 0.44481078, 0.36326131, 0.19785663]], dtype=np.float32)


def _gaussian_edge_count(E_edges):
    """The reference's `convolve(E_edges, tile(gaussian5by5, 8 edge types), mode='same')[:, :, 4]` (:2592-2598):
    the same 5x5 weights apply to every edge type, so this is the Gaussian-weighted count of the edges around
    (t, f), zeros outside (float64 sums of 25 shifted planes)."""
    cnt = np.asarray(E_edges, np.float64).sum(2)
    T, F = cnt.shape
    Z = np.zeros((T + 4, F + 4))
    Z[2:2 + T, 2:2 + F] = cnt
    g = gaussian5by5.astype(np.float64)
    out = np.zeros((T, F))
    for i in range(5):
        for j in range(5):
            out += g[i, j] * Z[4 - i:4 - i + T, 4 - j:4 - j + F]
    return out


def _get_part_features_mag_aux(E, parameters):
    """The mag_data + auxiliary_data branch of get_part_features (:2577-2614): E [T, 8 F + mag channels +
    auxiliary channels] -> Gaussian edge-count mask (> 12) -> code_parts_mag_aux -> argmax -> spread_patches,
    zero rows appended up to T frames."""
    E = np.asarray(E).T.astype(np.uint8)
    nm, na = int(parameters.num_mag_channels), int(parameters.num_auxiliary_data)
    E_edges, E_mags, E_aux = E[:-nm - na].T, E[-nm - na:-na].T, E[-na:].T
    num_edges = 8
    F = E_edges.shape[1] // num_edges
    E_edges = np.ascontiguousarray(E_edges).reshape(len(E_edges), F, num_edges)
    if tuple(np.shape(parameters.logParts)[1:3]) != (5, 5):
        raise NotImplementedError("get_part_features: the reference's non-5x5 Gaussian uses an undefined name (part_freq)")
    mask = (_gaussian_edge_count(E_edges) > 12).astype(np.uint8)
    out = cp.code_parts_mag_aux(E_edges, np.ascontiguousarray(E_mags), np.ascontiguousarray(E_aux), parameters.logParts,
                                parameters.logInvParts, parameters.logMagFeatures, parameters.logInvMagFeatures,
                                parameters.logAuxiliaryFeatures, parameters.logInvAuxiliaryFeatures, mask)
    max_responses = np.argmax(out, -1)
    spread = cp.spread_patches(max_responses, parameters.spreadRadiusX, parameters.spreadRadiusY, parameters.numParts)
    missing = E.shape[1] - len(spread)
    if missing > 0:
        spread = np.vstack((spread, np.zeros((missing,) + spread.shape[1:])))
    return spread


def get_waliji_feature_map_from_edges(E2, log_part_blocks, log_invpart_blocks):
    """The waliji map from the reorganised edge map E2 [T, F, 8] on (:183-191):
    code_parts_fast(threshold 10) -> argmax -> spread over the patch size ->
    one cell per patch-sized grid step."""
    parts = _device_parts(log_part_blocks, log_invpart_blocks)
    return parts.features(np.asarray(E2).astype(np.uint8), 10, spread=(parts.P0, parts.P1),
                          grid=(parts.P0, parts.P1), fast=True)


def phns_syllable_matches(phns, syllable):
    """:66-70: indices where the phone sequence `syllable` starts in `phns`."""
    syllable_len = len(syllable)
    return np.array([phn_id for phn_id in range(len(phns) - syllable_len + 1)
                     if np.all(phns[phn_id:phn_id + syllable_len] == np.array(syllable))])


def map_array_to_coarse_coordinates(xs, coarsen_factor):
    """:83-85."""
    return ((xs + coarsen_factor / 2.) / coarsen_factor).astype(int)


def get_examples_from_phns_ftls(syllable, phns, flts, log_part_blocks, coarse_length, verbose=False, phn_mapping=None):
    """:87-131: (starts, ends) of the occurrences of `syllable`, or (None, None)."""
    phn_matches = phns_syllable_matches(phns, syllable)
    if phn_matches.shape[0] == 0:
        return None, None
    syllable_starts = flts[phn_matches]
    syllable_ends = flts[phn_matches + len(syllable)]
    if log_part_blocks is not None:
        coarse_factor = log_part_blocks.shape[1]
        syllable_starts = map_array_to_coarse_coordinates(syllable_starts, coarse_factor)
        syllable_ends = np.clip(map_array_to_coarse_coordinates(syllable_ends, coarse_factor),
                                syllable_starts + 1, coarse_length)
    return syllable_starts, syllable_ends


def _compute_detection_E(E, phns, E_flts, detection_array, detection_template_ids, detect_lengths,
                         linear_filters_cs, syllable, example_start_end_times, phn_mapping=None, verbose=False,
                         flags=api.TSR_EXACT, bank=None):
    """:927-1012 with the reference's signature: score one utterance against the bank into the next row of
    detection_array / detection_template_ids (row = len(detect_lengths)), append the (start, end) times of the
    occurrences of `syllable` to example_start_end_times and the detect length to detect_lengths (0 for a
    single-filter bank, as the reference leaves it).  `flags` / `bank` (a prepared tsr_hip.Bank) are extensions."""
    example_starts, example_ends = get_examples_from_phns_ftls(syllable, phns, E_flts, None, E.shape[0], verbose=verbose)
    E8 = np.asarray(E).astype(np.uint8)
    bank = bank or api.Bank(linear_filters_cs)
    scores, ids = bank.score(E8, flags=flags)
    row = len(detect_lengths)
    detection_array[row, :E.shape[0]] = scores
    if detection_template_ids is not None and len(linear_filters_cs) > 1:
        detection_template_ids[row, :E.shape[0]] = ids
    if example_starts is not None:
        example_start_end_times.append(list(zip(example_starts, example_ends)))
    else:
        example_start_end_times.append([])
    detect_lengths.append(E.shape[0] if len(linear_filters_cs) > 1 else 0)


def get_detection_scores_E(utterances, linear_filters_cs, P_config=None, flags=api.TSR_EXACT, ctx=None):
    """(Named ..._E, like get_classify_scores_E, because the reference's get_detection_scores is a different,
    file-driven function.)  The corpus loop of get_detection_scores_mixture_named_params (:1238-1339) from
    the edge maps on: `utterances` is a sequence of binary maps E_n [T_n, F, K]
    (edge maps, or part maps when P_config is None).  Returns
    (detection_array f32 [N, maxT+2], detection_template_ids i32 [N, maxT+2],
    lengths [N]) -- the arrays main_multiprocessing.py:1484-1486 allocates -- with all
    utterances scored in one batched launch."""
    maps = [get_part_features(E, P_config) if P_config is not None else np.asarray(E).astype(np.uint8)
            for E in utterances]
    lengths = np.array([m.shape[0] for m in maps], np.int64)
    offsets = np.concatenate(([0], np.cumsum(lengths)))
    bank = api.Bank(linear_filters_cs, ctx=ctx)
    feats = np.concatenate([m.reshape(m.shape[0], -1) for m in maps]) if maps else np.zeros((0, bank.D), np.uint8)
    scores, ids = bank.score(feats, offsets, flags=flags)
    maxT = int(lengths.max()) if len(lengths) else 0
    detection_array = np.zeros((len(maps), maxT + 2), np.float32)
    detection_template_ids = np.zeros((len(maps), maxT + 2), np.int32)
    for n in range(len(maps)):
        detection_array[n, :lengths[n]] = scores[offsets[n]:offsets[n + 1]]
        detection_template_ids[n, :lengths[n]] = ids[offsets[n]:offsets[n + 1]]
    return detection_array, detection_template_ids, lengths


# ---- classification over phone windows (:710-925) ---------------------------------------------
def get_classify_windows(phns, flts):
    """:710-716 -- per phone the frame range in which its detection is looked for: a third of the
    previous phone before its start to a third of its own length after it."""
    flts = np.asarray(flts)
    window_starts = np.maximum(0, flts[:-1] + (flts[:-1] - flts[1:]) / 3.).astype(np.int16)
    window_ends = np.minimum(flts[-1], flts[:-1] + (flts[1:] - flts[:-1]) / 3.).astype(np.int16)
    return window_starts, window_ends


def _store_classification(utt_id, score, loc, tid, linear_filters_cs, classify_array, classify_template_ids,
                          classify_template_lengths, classify_locs):
    n = len(score)
    classify_array[utt_id, :n] = score
    classify_locs[utt_id, :n] = loc
    classify_template_ids[utt_id, :n] = tid
    classify_template_lengths[utt_id, :n] = [len(linear_filters_cs[t][0]) for t in tid]


def _compute_classification_E(E, phns, E_flts, utt_id, classify_array, classify_template_ids,
                              classify_template_lengths, classify_locs, linear_filters_cs,
                              phn_mapping=None, verbose=False, flags=api.TSR_EXACT, bank=None):
    """:718-773 -- for every phone of the utterance: the best detection score inside its window, where
    (frame) and by which template (first maximum per template, earlier template on ties).  One scoring
    pass + one windowed arg-max launch instead of a Python loop per (template, phone)."""
    window_starts, window_ends = get_classify_windows(phns, E_flts)
    E = np.asarray(E).astype(np.uint8)
    bank = bank or api.Bank(linear_filters_cs)
    corpus = api.Corpus(E.reshape(E.shape[0], -1), [0, E.shape[0]], ctx=bank.ctx)
    score, loc, tid = corpus.classify_windows(bank, window_starts, window_ends, flags=flags)
    _store_classification(utt_id, score, loc, tid, linear_filters_cs, classify_array, classify_template_ids,
                          classify_template_lengths, classify_locs)


def get_isolated_classify_windows(E, phns, flts, bgd, linear_filters_cs, max_length=None):
    """:781-835 -- one background-padded f32 feature window per phone.  Like the reference this shifts
    `flts` IN PLACE by the front padding; returns (E_windows, window_starts, window_ends, flts[0])."""
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


def _isolated_virtual_windows(T, flts, max_filter_length, frame0=0):
    """The windows of get_isolated_classify_windows described instead of built: for every phone the
    stream frame of its first row (frame0 = first frame of the utterance; rows before it are front
    padding), the range of real frames, the row count and the number of searched positions -- what
    tsr_corpus_classify_isolated consumes.  Also returns (window_starts, front pad) for the locations."""
    flts = np.asarray(flts, dtype=np.int64)
    max_phn_length_third = int(np.ceil(np.max(flts[1:] - flts[:-1]) / 3.))
    pad = max_filter_length + max_phn_length_third
    len_padded = pad + T + max(0, int(flts[-2]) + max_phn_length_third + max_filter_length - int(flts[-1]))
    f = flts + pad
    ws = (f[:-1] + (f[:-1] - f[1:]) / 3.).astype(np.int16).astype(np.int64)
    we = (f[:-1] + (f[1:] - f[:-1]) / 3.).astype(np.int16).astype(np.int64)
    pse = (f[1:] + (f[1:] - f[:-1]) / 3.).astype(np.int16).astype(np.int64)
    avail_end = np.minimum(pse, len_padded)                     # E[ws:pse] clips at the padded length
    rows = (avail_end - ws) + np.maximum(0, we + max_filter_length - pse)
    first = frame0 + ws - pad
    real_hi = frame0 + np.clip(avail_end - pad, 0, T)
    real_lo = np.full_like(first, frame0)
    return first, real_lo, real_hi, rows, we - ws, ws, int(f[0]), pad


def _isolated_virtual_windows_batch(Ts, flts_list, max_filter_length, frame0s):
    """_isolated_virtual_windows for a whole corpus in one pass of array operations (the per-utterance Python
    loop took 17 ms for 1000 utterances -- six times the kernel it feeds).  Returns the concatenated
    (first, real_lo, real_hi, rows, n_valid, window_starts) and per utterance (phone count, f[0])."""
    n_ph = np.array([len(f) - 1 for f in flts_list], dtype=np.int64)
    if (n_ph < 1).any():
        raise ValueError("every utterance needs at least one phone (two boundaries)")
    N = len(flts_list)
    F = np.concatenate([np.asarray(f, dtype=np.int64) for f in flts_list])
    Ts = np.asarray(Ts, dtype=np.int64)
    frame0s = np.asarray(frame0s, dtype=np.int64)
    b0 = np.concatenate(([0], np.cumsum(n_ph + 1)))[:-1]          # index in F of each utterance's first boundary
    p0 = np.concatenate(([0], np.cumsum(n_ph)))                    # index of each utterance's first phone
    utt = np.repeat(np.arange(N), n_ph)
    left = b0[utt] + (np.arange(p0[-1]) - p0[:-1][utt])
    fl, fr = F[left], F[left + 1]
    third = np.ceil(np.maximum.reduceat(fr - fl, p0[:-1]) / 3.).astype(np.int64)
    pad = max_filter_length + third
    last, second_last = F[b0 + n_ph], F[b0 + n_ph - 1]
    len_padded = pad + Ts + np.maximum(0, second_last + third + max_filter_length - last)
    gl, gr = fl + pad[utt], fr + pad[utt]

    def i16(x):                                                    # the reference stores these as int16
        return x.astype(np.int16).astype(np.int64)

    ws, we, pse = i16(gl + (gl - gr) / 3.), i16(gl + (gr - gl) / 3.), i16(gr + (gr - gl) / 3.)
    avail_end = np.minimum(pse, len_padded[utt])
    rows = (avail_end - ws) + np.maximum(0, we + max_filter_length - pse)
    first = frame0s[utt] + ws - pad[utt]
    real_hi = frame0s[utt] + np.clip(avail_end - pad[utt], 0, Ts[utt])
    return first, frame0s[utt].copy(), real_hi, rows, we - ws, ws, n_ph, F[b0] + pad


def _compute_isolated_classification_E(E, phns, E_flts, utt_id, classify_array, classify_template_ids,
                                       classify_template_lengths, classify_locs, linear_filters_cs, bgd,
                                       phn_mapping=None, verbose=False, svm_classifiers=None,
                                       svm_constants=None, svm_classify_array=None, bank=None):
    """:838-925 -- every phone is scored in isolation on its own background-padded window
    (detect_float / detect_float2d + c).  The windows are not built: the kernel reads the binary map
    and substitutes the background vector for every padding row, all phones and templates of the
    utterance in one launch.  Like the reference this shifts `E_flts` IN PLACE by the front padding.
    The optional linear-SVM re-scoring of the winning window (:913-921) is a NumPy expression in the
    reference and stays one here."""
    bgd = np.asarray(bgd)
    E = np.asarray(E)
    if bgd.dtype != np.float32:   # the f32 windows would be promoted and detect_float's buffer check raise
        compute_likelihood_linear_filter._check_float_features(np.zeros((1,) + bgd.shape, bgd.dtype))
    max_filter_length = max(len(lfc[0]) for lfc in linear_filters_cs)
    first, lo, hi, rows, n_valid, ws, flt_front_pad, pad = _isolated_virtual_windows(E.shape[0], E_flts, max_filter_length)
    if svm_classifiers is not None and svm_classify_array is not None:
        E_windows = get_isolated_classify_windows(E, phns, np.array(E_flts), bgd, linear_filters_cs)[0]
    E_flts += pad
    bank = bank or api.Bank(linear_filters_cs)
    E8 = E.astype(np.uint8)
    corpus = api.Corpus(E8.reshape(E8.shape[0], -1), [0, E8.shape[0]], ctx=bank.ctx)
    score, pos, tid = corpus.classify_isolated(bank, bgd, first, lo, hi, rows, n_valid)
    _store_classification(utt_id, score, ws + pos - flt_front_pad, tid, linear_filters_cs, classify_array,
                          classify_template_ids, classify_template_lengths, classify_locs)
    if svm_classifiers is not None and svm_classify_array is not None:
        for phn_id, (E_window, p) in enumerate(zip(E_windows, pos)):
            t, n = int(tid[phn_id]), len(linear_filters_cs[int(tid[phn_id])][0])
            svm_classify_array[utt_id, phn_id] = (E_window[p:p + n] * svm_classifiers[t]).sum() + svm_constants[t]


def get_classify_scores_E(utterances, flts_list, linear_filters_cs, bgd=None, isolated=False,
                          flags=api.TSR_EXACT, ctx=None):
    """The corpus loop of get_classify_scores (:1341-1441) from the feature maps on, batched: every
    phone window of every utterance in one launch.  `utterances[n]` is a binary map [T_n, ...],
    `flts_list[n]` its phone boundaries (frames, len = phones + 1; not modified).  isolated=False is
    _compute_classification_E, isolated=True _compute_isolated_classification_E (needs the f32 `bgd`).
    Returns (classify_array f32, classify_locs i16, classify_template_lengths u16,
    classify_template_ids u16), each [N, max phones], as the reference allocates them."""
    N = len(utterances)
    nph = max((len(f) - 1 for f in flts_list), default=0)
    classify_array = np.zeros((N, nph), np.float32)
    classify_template_ids = np.zeros((N, nph), np.uint16)
    classify_locs = np.zeros((N, nph), np.int16)
    classify_template_lengths = np.zeros((N, nph), np.uint16)
    bank = api.Bank(linear_filters_cs, ctx=ctx)
    if not isolated:
        maps = [np.asarray(E).astype(np.uint8) for E in utterances]
        offs = np.concatenate(([0], np.cumsum([m.shape[0] for m in maps]))).astype(np.int64)
        corpus = api.Corpus(np.concatenate([m.reshape(m.shape[0], -1) for m in maps]), offs, ctx=bank.ctx)
        wins = [get_classify_windows(None, f) for f in flts_list]
        starts = np.concatenate([ws.astype(np.int64) + offs[n] for n, (ws, _) in enumerate(wins)])
        ends = np.concatenate([we.astype(np.int64) + offs[n] for n, (_, we) in enumerate(wins)])
        score, loc, tid = corpus.classify_windows(bank, starts, ends, flags=flags)
        k = 0
        for n, (ws, _) in enumerate(wins):
            m = len(ws)
            _store_classification(n, score[k:k + m], loc[k:k + m] - offs[n], tid[k:k + m], linear_filters_cs,
                                  classify_array, classify_template_ids, classify_template_lengths, classify_locs)
            k += m
        return classify_array, classify_locs, classify_template_lengths, classify_template_ids
    bgd = np.asarray(bgd)
    if bgd.dtype != np.float32:
        compute_likelihood_linear_filter._check_float_features(np.zeros((1,) + bgd.shape, bgd.dtype))
    maps = [np.asarray(E).astype(np.uint8) for E in utterances]
    offs = np.concatenate(([0], np.cumsum([m.shape[0] for m in maps]))).astype(np.int64)
    corpus = api.Corpus(np.concatenate([m.reshape(m.shape[0], -1) for m in maps]), offs, ctx=bank.ctx)
    Lmax = max(len(lfc[0]) for lfc in linear_filters_cs)
    first, lo, hi, rows, n_valid, ws, n_ph, f0 = _isolated_virtual_windows_batch(
        [m.shape[0] for m in maps], flts_list, Lmax, offs[:-1])
    score, pos, tid = corpus.classify_isolated(bank, bgd, first, lo, hi, rows, n_valid)
    # results into the [N, max phones] arrays in one scatter (what _store_classification does per utterance)
    utt = np.repeat(np.arange(N), n_ph)
    col = np.arange(len(utt)) - np.repeat(np.cumsum(n_ph) - n_ph, n_ph)
    lengths = np.array([len(lfc[0]) for lfc in linear_filters_cs])
    classify_array[utt, col] = score
    classify_template_ids[utt, col] = tid
    classify_template_lengths[utt, col] = lengths[tid]
    classify_locs[utt, col] = ws + pos - np.repeat(f0, n_ph)
    return classify_array, classify_locs, classify_template_lengths, classify_template_ids


# ---- edge front-end (:2489-2557) ----------------------------------------------------------------
def _edge_parameters(parameters, use_abst=True):
    kw = dict(block_length=parameters.block_length, spread_length=parameters.spread_length, threshold=parameters.threshold)
    if use_abst and getattr(parameters, "abst_threshold", None) is not None:
        kw["abst_threshold"] = parameters.abst_threshold
    return kw


def get_edge_features(S, parameters, verbose=False):
    """Spectrogram S [H, T] (features x time, as the reference passes `S.T`) -> edge features, with the
    reference's branches (get_train_data.py:2489-2557); `parameters` is duck-typed like EdgemapParameters.
      default                      thresholded, spread edge map [T, H, 8] (float64 zeros and ones)
      magnitude_and_edge_features  [T, 8 H + H'] = the edge map in the reference's orientation-major column
                                   order beside the whole-block magnitude map (+ 3 auxiliary columns)
      magnitude_thresholded_edges  the edge strengths masked by the (time-reversed, as in the reference)
                                   magnitude map before the thresholding, [T, H, 8]
      magnitude_features           a TypeError, as in the reference (its call lacks an argument, :2494-2498)"""
    S = np.asarray(S, np.float64)
    aux = bool(getattr(parameters, "auxiliary_data", False))
    if aux:
        A, S = S[-3:], S[:-3]
    H, T = S.shape
    if getattr(parameters, "magnitude_features", False):
        raise TypeError("magnitude_features() takes at least 6 arguments (5 given)")
    if getattr(parameters, "magnitude_and_edge_features", False):
        E2 = esp.magnitude_features_whole_block(S, parameters.magnitude_block_length, parameters.magnitude_spread,
                                                parameters.magnitude_threshold, parameters.mag_smooth_freq,
                                                parameters.mag_downsample_freq)
        corpus = api.Corpus.from_spectrograms(np.ascontiguousarray(S.T), [0, T], **_edge_parameters(parameters, use_abst=False))
        # [T, H, 8] -> the reference's E.T: column e * H + h
        ET = corpus.download_feats().reshape(T, H, 8).transpose(0, 2, 1).reshape(T, 8 * H).astype(np.float64)
        if not aux:
            return np.hstack((ET, E2))
        AB = esp.magnitude_features(A, parameters.magnitude_block_length, 0, parameters.auxiliary_threshold, 0, 0,
                                    mag_spread_length=1)
        return np.hstack((ET, E2, AB))
    if getattr(parameters, "magnitude_thresholded_edges", False):
        E2 = esp.magnitude_features_whole_block(S, parameters.magnitude_block_length, parameters.magnitude_spread,
                                                parameters.magnitude_threshold, parameters.mag_smooth_freq,
                                                parameters.mag_downsample_freq)
        if E2.shape[1] != H:
            raise ValueError("operands could not be broadcast together: the magnitude map has %d bins, the edge map %d"
                             % (E2.shape[1], H))
        mask = np.ascontiguousarray(E2[::-1])        # `E2[::-1].T`: the reference flips the time axis here (:2537)
        corpus = api.Corpus.from_spectrograms(np.ascontiguousarray(S.T), [0, T], mask=mask, **_edge_parameters(parameters))
        return corpus.download_feats().reshape(T, H, 8).astype(np.float64)
    return get_edge_corpus([S.T], parameters).download_feats().reshape(T, H, 8).astype(np.float64)


def get_edge_corpus(spectrograms, parameters, ctx=None):
    """The same for a list of time-major spectrograms [T_n, H], all utterances in one launch; the result
    stays on the device as a corpus (feed it to Corpus.part_features / Corpus.score)."""
    specs = [np.ascontiguousarray(S, dtype=np.float64) for S in spectrograms]
    offs = np.concatenate(([0], np.cumsum([S.shape[0] for S in specs]))).astype(np.int64)
    kw = {}
    if getattr(parameters, "abst_threshold", None) is not None:
        kw["abst_threshold"] = parameters.abst_threshold
    return api.Corpus.from_spectrograms(np.concatenate(specs), offs, block_length=parameters.block_length,
                                        spread_length=parameters.spread_length, threshold=parameters.threshold,
                                        ctx=ctx, **kw)
