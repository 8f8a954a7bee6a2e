"""The `Classifier` scoring surface of the reference's classification.py:26-160 on
the GPU path.

Same constructor and attributes (`score`, `score_no_bg`, `score_register`,
`coarse_score_like`, `coarse_score_like_no_bg`, `coarse_score_count`, `window`,
`coarse_length`, `type`), each accepting one [D, L] window of binary edge
features (features x time) like the reference -- plus `score_utterance`, which
scores EVERY window of a whole [D, T] utterance in one launch (what the ROC
harnesses do with a Python loop over `d`, classification.py:459-461, 601-603).
The ROC harnesses themselves are outside the accelerated path.
"""
import hashlib

import numpy as np

from tsr_hip import api
from . import test_template as tt


def map_coarse_frame_to_range(frame_id, coarse_factor, length):
    """Frames [lo, hi) of the base segment that coarse frame `frame_id` covers (:194-241)."""
    return coarse_factor * frame_id, min(coarse_factor * (frame_id + 1), length)


def get_coarse_segment(segment, coarse_type="max", coarse_factor=2):
    """Shorter-in-time version of a template ("avg") or window ("max"), :166-192."""
    height, length = segment.shape
    coarse_length = int(length / coarse_factor)
    out = np.zeros((height, coarse_length))
    reduce_ = np.mean if coarse_type == "avg" else np.max
    for f in range(coarse_length):
        lo, hi = map_coarse_frame_to_range(f, coarse_factor, length)
        out[:, f] = reduce_(segment[:, lo:hi], axis=1)
    return out


def _stretch_template(T, template_height, template_length):
    """Nearest-frame time warp of a template to `template_length` frames (:11-23)."""
    example_length = T.shape[1]
    out = np.zeros((template_height, template_length))
    if example_length < template_length:
        for f in range(template_length):
            out[:, f] = T[:, int(f / float(template_length) * example_length)]
    else:
        for f in range(example_length):
            g = int(f / float(example_length) * template_length)
            out[:, g] = np.maximum(out[:, g], T[:, f])
    return out


def _score(template, bg, E_window):
    return sum(tt.score_template_background_section(template, bg, E_window))


class Classifier:
    def __init__(self, base_object, coarse_factor=2, coarse_template_threshold=.5, bg=None):
        self.coarse_factor = coarse_factor
        self.bg = bg
        cf = coarse_factor
        coarse_E = lambda E_window: get_coarse_segment(E_window, coarse_type="max", coarse_factor=cf)
        if isinstance(base_object, list):
            height = base_object[0].shape[0]
            length = max(t.shape[1] for t in base_object)
            self.type = "collection"
            self.window = np.array((height, length))
            self.template = base_object
            self.pad_template = lambda tmplt, bg: np.hstack(
                (tmplt, np.tile(np.asarray(bg).reshape(self.window[0], 1), (1, self.window[1] - tmplt.shape[1]))))
            self._members = lambda bg: [self.pad_template(t, bg) for t in self.template]
            self.score = lambda E_window, bg: max(_score(t, bg, E_window) for t in self._members(bg))
            self.score_register = lambda E_window, bg: max(
                _score(_stretch_template(t, E_window.shape[0], E_window.shape[1]), bg, E_window)
                for t in self.template)
            self.score_no_bg = lambda E_window: self.score(E_window, self.bg)
            self.coarse_template = [get_coarse_segment(t, coarse_type="avg", coarse_factor=cf)
                                    for t in self.template]
            self.coarse_length = self.coarse_template[0].shape[1]
            self.coarse_template_mask = [T > .7 for T in self.coarse_template]
            self.coarse_score_like = lambda E_window, bg: max(
                _score(T, bg, coarse_E(E_window)) for T in self.coarse_template)
            self.coarse_score_like_no_bg = lambda E_window: self.coarse_score_like(E_window, self.bg)
            self.coarse_score_count = lambda E_window: max(
                np.sum(coarse_E(E_window)[m]) for m in self.coarse_template_mask)
        elif isinstance(base_object, np.ndarray):
            height, length = base_object.shape
            self.type = "template"
            self.window = np.array((height, length))
            self.template = base_object
            self._members = lambda bg: [self.template]
            self.score = lambda E_window, bg: _score(self.template, bg, E_window)
            self.score_register = lambda E_window, bg: _score(
                _stretch_template(self.template, E_window.shape[0], E_window.shape[1]), bg, E_window)
            self.score_no_bg = lambda E_window: self.score(E_window, self.bg)
            self.coarse_template = get_coarse_segment(self.template, coarse_type="avg", coarse_factor=cf)
            self.coarse_length = self.coarse_template.shape[1]
            self.coarse_template_mask = self.coarse_template > .7
            self.coarse_score_like = lambda E_window, bg: _score(self.coarse_template, bg, coarse_E(E_window))
            self.coarse_score_like_no_bg = lambda E_window: self.coarse_score_like(E_window, self.bg)
            self.coarse_score_count = lambda E_window: np.sum(coarse_E(E_window)[self.coarse_template_mask])
        elif hasattr(base_object, "def_templates") and hasattr(base_object, "base_template"):
            # parts_model.TwoPartModel (duck-typed): max over the deformed templates (:120-149)
            height, length = base_object.bg.shape[0], base_object.length_range[1]
            self.bg = bg if bg is not None else .4 * np.ones(height)
            self.type = "TwoPartModel"
            self.window = np.array((height, length))
            self.template = base_object
            self._members = lambda bg: list(self.template.def_templates)
            self._score_sub_no_bg = lambda E_window, t_id: _score(
                self.template.def_templates[t_id], self.template.bg, E_window)
            self.score_no_bg = lambda E_window: max(
                self._score_sub_no_bg(E_window, t_id) for t_id in range(self.template.def_templates.shape[0]))
            self.coarse_template = get_coarse_segment(self.template.base_template, coarse_type="avg",
                                                      coarse_factor=cf)
            self.coarse_length = self.coarse_template.shape[1]
            self.coarse_template_mask = self.coarse_template > .7
            self.coarse_score_like_no_bg = lambda E_window: _score(
                self.coarse_template, self.bg, coarse_E(E_window)[:, :self.coarse_length])
            self.coarse_score_count = lambda E_window: np.sum(
                coarse_E(E_window)[:, :self.coarse_length][self.coarse_template_mask])
        else:
            raise TypeError("Classifier needs a template (2-d ndarray), a list of templates or a "
                            "TwoPartModel-like object")
        self._bank_cache = None

    # ---- whole-utterance scoring (one launch for all windows) --------------------------
    def _bank(self, bg):
        bgd = np.asarray(self.template.bg if self.type == "TwoPartModel" else bg, np.float64)
        members = [np.ascontiguousarray(np.asarray(t, np.float64)) for t in self._members(bgd)]
        h = hashlib.blake2b(bgd.tobytes(), digest_size=16)  # content key: templates may be updated in place
        for t in members:
            h.update(repr(t.shape).encode())
            h.update(t.tobytes())
        key = h.digest()
        if self._bank_cache is None or self._bank_cache[0] != key:
            filters = []
            for tmpl in members:
                Tm = np.ascontiguousarray(np.asarray(tmpl, np.float64).T)  # [L, D]
                neg = (1. - Tm) / (1. - bgd)
                filters.append((np.log((Tm / bgd) / neg).astype(np.float32), float(np.log(neg).sum())))
            self._bank_cache = (key, api.Bank(filters))
        return self._bank_cache[1]

    def score_utterance(self, E, bg=None, flags=api.TSR_EXACT, return_ids=False):
        """E [D, T] binary (features x time) -> scores float32 [T - L + 1], where
        scores[d] == self.score(E[:, d:d+L], bg) for the window length L = self.window[1]."""
        bg = self.bg if bg is None else bg
        if bg is None:
            raise ValueError("a background vector is needed (bg=... here or at construction)")
        E = np.asarray(E)
        n = E.shape[1] - int(self.window[1]) + 1
        if n <= 0:
            raise ValueError("utterance shorter than the classifier window")
        scores, ids = self._bank(bg).score(np.ascontiguousarray(E.T).astype(np.uint8), flags=flags)
        return (scores[:n], ids[:n]) if return_ids else scores[:n]
