"""The per-window scorer the reference's Classifier wraps
(test_template.py:25-41), served by the GPU scoring path.

The reference recomputes the log-odds filter from (template, bgd) on every call
and evaluates sum(E * log W) and C in float64 NumPy.  Here the filter is built
once per (template, bgd) pair, cached, and the window sum runs through the same
kernel as compute_likelihood_linear_filter.detect (float32, reference order),
which agrees with the float64 value to ~1e-6 relative."""
import hashlib

import numpy as np

from . import compute_likelihood_linear_filter as cllf

_filter_cache = {}


def _window_filter(template, bgd, front_bgd_pad, back_bgd_pad):
    # keyed by CONTENT (a digest of the bytes), not by object identity: callers re-estimate templates
    # in place between calls
    template = np.asarray(template)
    key = (hashlib.blake2b(np.ascontiguousarray(template).tobytes(), digest_size=16).digest(), template.shape,
           template.dtype.str, np.asarray(bgd).tobytes(), front_bgd_pad, back_bgd_pad)
    hit = _filter_cache.get(key)
    if hit is not None:
        return hit
    bgd = np.asarray(bgd, dtype=np.float64)
    col = bgd.reshape(-1, 1)
    full = np.hstack((np.tile(col, (1, front_bgd_pad)), np.asarray(template, np.float64),
                      np.tile(col, (1, back_bgd_pad))))
    # time-major [L, D]; no clipping here: the reference's scorer does not clip
    Tm = np.ascontiguousarray(full.T)
    neg = (1. - Tm) / (1. - bgd)
    W = np.log((Tm / bgd) / neg).astype(np.float32)
    C = float(np.log(neg).sum())
    if len(_filter_cache) > 64:
        _filter_cache.clear()
    _filter_cache[key] = (W, C)
    return W, C


def score_template_background_section(template, bgd, E, front_bgd_pad=0, back_bgd_pad=0):
    """template [D, L], bgd [D], E [D, L + pads] (features x time) ->
    (sum E*log W, C); callers add the two (classification.py:44-149)."""
    E = np.asarray(E)
    assert E.shape[1] == template.shape[1]
    W, C = _window_filter(template, bgd, front_bgd_pad, back_bgd_pad)
    Et = np.ascontiguousarray(E.T)
    D = Et.shape[1]
    if Et.dtype == np.bool_ or np.all((Et == 0) | (Et == 1)):
        s = cllf.detect(Et.astype(np.uint8).reshape(-1, D, 1), W.reshape(-1, D, 1))[0]
    else:
        s = cllf.detect_float2d(Et.astype(np.float32), W)[0]
    return float(s), C
