"""Per-utterance glue of the reference's get_train_data.py that sits directly on
the native kernels, re-hosted on the GPU path:

    reorg_part_for_fast_filtering / reorg_parts_for_fast_filtering   :41-57
    collapse_to_grid                                                  :72-81
    get_waliji_feature_map (from the edge map on)                     :182-191
    _compute_detection_E (scoring + template ids)                     :927-1012
    get_detection_scores_mixture_named_params (the corpus loop)       :1238-1339
    get_part_features (edge-only branch)                              :2576-2631

Waveform -> spectrogram -> edge map (edge_signal_proc.py) is upstream of the
path and stays with the caller: every function here starts from the binary
edge map E.
"""
import numpy as np

from tsr_hip import api
from . import code_parts as cp
from . import compute_likelihood_linear_filter
from . import spread_waliji_patches as swp


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


def _device_parts(log_parts, log_invparts):
    key = (id(log_parts), id(log_invparts))
    hit = _parts_cache.get(key)
    if hit is None or hit[0] is not log_parts or hit[1] is not log_invparts:
        if len(_parts_cache) > 8:
            _parts_cache.clear()
        hit = (log_parts, log_invparts, api.Parts(np.asarray(log_parts, np.float64),
                                                  np.asarray(log_invparts, np.float64)))
        _parts_cache[key] = hit
    return hit[2]


def get_part_features(E, parameters, verbose=False):
    """Edge map E [T, F, 8] -> spread part map uint8 [T-P0+1, F-P1+1, numParts]:
    code_parts -> argmax -> spread_patches -> optional part-graph closure, fused on
    the GPU.  `parameters` is duck-typed like get_train_data.PartsParameters
    (logParts, logInvParts, bernsteinEdgeThreshold, spreadRadiusX, spreadRadiusY,
    numParts, partGraph; :2203-2473)."""
    if getattr(parameters, "mag_data", False) and getattr(parameters, "auxiliary_data", False):
        raise NotImplementedError("the magnitude/auxiliary-channel coder (code_parts_mag_aux) is "
                                  "not part of the accelerated path")
    parts = _device_parts(parameters.logParts, parameters.logInvParts)
    if int(parameters.numParts) != parts.K:
        raise ValueError("numParts=%d but logParts holds %d parts" % (parameters.numParts, parts.K))
    return parts.features(np.asarray(E).astype(np.uint8), parameters.bernsteinEdgeThreshold,
                          spread=(parameters.spreadRadiusX, parameters.spreadRadiusY),
                          graph=getattr(parameters, "partGraph", None))


def get_waliji_feature_map_from_edges(E2, log_part_blocks, log_invpart_blocks):
    """The waliji map from the reorganised edge map E2 [T, F, 8] on (:183-191):
    code_parts_fast(threshold 10) -> argmax -> spread over the patch size ->
    one cell per patch-sized grid step."""
    parts = _device_parts(log_part_blocks, log_invpart_blocks)
    return parts.features(np.asarray(E2).astype(np.uint8), 10, spread=(parts.P0, parts.P1),
                          grid=(parts.P0, parts.P1), fast=True)


def _compute_detection_E(E, detection_array, detection_template_ids, detect_lengths,
                         linear_filters_cs, flags=api.TSR_EXACT, bank=None):
    """Score one utterance against the bank and append the result as the next row
    of detection_array / detection_template_ids, as :962-1011 does.  The phone-label
    bookkeeping of the reference (example start/end times) is the caller's."""
    E = np.asarray(E).astype(np.uint8)
    bank = bank or api.Bank(linear_filters_cs)
    scores, ids = bank.score(E, flags=flags)
    row = len(detect_lengths)
    detection_array[row, :E.shape[0]] = scores
    if detection_template_ids is not None:
        detection_template_ids[row, :E.shape[0]] = ids
    detect_lengths.append(E.shape[0] if len(linear_filters_cs) > 1 else 0)
    return scores, ids


def get_detection_scores(utterances, linear_filters_cs, P_config=None, flags=api.TSR_EXACT, ctx=None):
    """The corpus loop of get_detection_scores_mixture_named_params (:1238-1339) from
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
