"""Batched, device-resident face of libtsr_hip.so: Context, Bank, Corpus, Parts.

This is the host-side counterpart of the reference's per-utterance glue
(get_train_data.py:927-1012 `_compute_detection_E`, :2576-2631
`get_part_features`, :138-201 `get_waliji_feature_map`), re-shaped so that a
whole corpus is scored by one call with inputs and outputs resident in HBM.
NumPy + ctypes only.
"""
import ctypes as C
import os

import numpy as np

from ._lib import TSR_BF16, TSR_EXACT, TSR_FAST, TSR_MX, TSR_W24, check, lib, ptr  # noqa: F401

_default_ctx = None


class Context:
    """One GPU, one HIP stream, scratch buffers (tsr_ctx).  Not thread-safe."""

    def __init__(self, device=None):
        if device is None:
            device = int(os.environ.get("TSR_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        h = C.c_void_p()
        check(lib().tsr_ctx_create(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            lib().tsr_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(lib().tsr_ctx_sync(self._h))

    def timer_start(self):
        check(lib().tsr_timer_start(self._h))

    def timer_stop(self):
        ms = C.c_float()
        check(lib().tsr_timer_stop(self._h, C.byref(ms)))
        return float(ms.value)

    def info(self):
        name = C.create_string_buffer(128)
        cus, hbm = C.c_int(), C.c_int64()
        check(lib().tsr_device_info(self._h, name, 128, C.byref(cus), C.byref(hbm)))
        return dict(name=name.value.decode(), cus=int(cus.value), hbm_bytes=int(hbm.value))

    # ---- kernel-choice options (tsr_ctx_set_option): same results, different kernels ------
    def set_option(self, option, value):
        check(lib().tsr_ctx_set_option(self._h, int(option), int(value)))

    def get_option(self, option):
        v = C.c_int()
        check(lib().tsr_ctx_get_option(self._h, int(option), C.byref(v)))
        return int(v.value)

    def option(self, option, value):
        """`with ctx.option(TSR_OPT_FAST_KERNEL, TSR_FAST_KERNEL_INTERLEAVED): ...` -- set, run, restore."""
        ctx = self

        class _Scoped:
            def __enter__(self):
                self.old = ctx.get_option(option)
                ctx.set_option(option, value)
                return ctx

            def __exit__(self, *exc):
                ctx.set_option(option, self.old)
                return False
        return _Scoped()

    # ---- multi-GPU (one process per GPU; see tsr_hip.dist) -------------------------
    def dist_init(self, unique_id, nranks, rank):
        check(lib().tsr_dist_init(self._h, unique_id, int(nranks), int(rank)))

    def dist_finalize(self):
        check(lib().tsr_dist_finalize(self._h))


def default_ctx():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context()
    return _default_ctx


def pool_trim():
    """Give the idle device blocks of destroyed corpora back to the driver (tsr_pool_trim)."""
    check(lib().tsr_pool_trim())


def device_count():
    n = C.c_int()
    check(lib().tsr_device_count(C.byref(n)))
    return int(n.value)


def unique_id():
    buf = C.create_string_buffer(128)
    check(lib().tsr_dist_unique_id(buf))
    return buf.raw


def pack_filters(linear_filters_cs):
    """((W_m [L_m, ...], c_m), ...) -> (W_cat f32, L i32 [M], c f64 [M], D)."""
    Ws = [np.ascontiguousarray(w, dtype=np.float32) for w, _ in linear_filters_cs]
    if not Ws:
        raise ValueError("empty template bank")
    D = int(np.prod(Ws[0].shape[1:], dtype=np.int64))
    for w in Ws:
        if w.ndim < 2 or int(np.prod(w.shape[1:], dtype=np.int64)) != D:
            raise ValueError("all templates must share the feature shape %r" % (Ws[0].shape[1:],))
    L = np.array([w.shape[0] for w in Ws], np.int32)
    W_cat = np.concatenate([w.reshape(-1) for w in Ws])
    c = np.array([float(np.asarray(cc).reshape(-1)[-1]) if np.ndim(cc) else float(cc)
                  for _, cc in linear_filters_cs], np.float64)
    return W_cat, L, c, D


class Bank:
    """Device-resident template bank (tsr_bank): the tuple `linear_filters_cs` of
    get_train_data.py:927-992, i.e. ((W_m f32 [L_m, F, K], c_m), ...)."""

    def __init__(self, linear_filters_cs, ctx=None):
        self.ctx = ctx or default_ctx()
        W_cat, L, c, D = pack_filters(linear_filters_cs)
        self.M, self.D, self.L, self.c = len(L), D, L, c
        h = C.c_void_p()
        check(lib().tsr_bank_create(self.ctx._h, ptr(W_cat), ptr(L), ptr(c), self.M, D, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().tsr_bank_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def broadcast(self, root=0):
        """RCCL broadcast of the bank from `root` to every rank (ctx.dist_init first)."""
        check(lib().tsr_bank_broadcast(self.ctx._h, self._h, int(root)))

    def classify_float_windows(self, E_windows, n_valid):
        """detect_float + c on each background-padded f32 window, arg-max over its first
        n_valid[w] positions, best template (get_train_data.py:838-925) ->
        (score f32 [W], position inside the window i32 [W], template i32 [W])."""
        wins = [np.ascontiguousarray(w, dtype=np.float32).reshape(len(w), -1) for w in E_windows]
        for w in wins:
            if w.shape[1] != self.D:
                raise ValueError("window rows have %d features, the bank %d" % (w.shape[1], self.D))
        W = len(wins)
        offs = np.concatenate(([0], np.cumsum([w.shape[0] for w in wins]))).astype(np.int64)
        cat = np.concatenate(wins) if W else np.zeros((0, self.D), np.float32)
        nv = np.ascontiguousarray(n_valid, dtype=np.int32)
        if nv.shape != (W,):
            raise ValueError("n_valid must have one entry per window")
        score, loc, tid = np.empty(W, np.float32), np.empty(W, np.int32), np.empty(W, np.int32)
        check(lib().tsr_classify_float_windows(self.ctx._h, self._h, ptr(cat), ptr(offs), ptr(nv), W,
                                               ptr(score), ptr(loc), ptr(tid)))
        return score, loc, tid

    # ---- scoring ---------------------------------------------------------------------
    def _offsets(self, feats, frame_offsets):
        feats = np.ascontiguousarray(feats, dtype=np.uint8)
        if frame_offsets is None:
            frame_offsets = [0, feats.shape[0]]
        offs = np.ascontiguousarray(frame_offsets, dtype=np.int64)
        P = int(offs[-1])
        if feats.size != P * self.D:
            raise ValueError("feats has %d bytes, expected %d frames x D=%d" % (feats.size, P, self.D))
        return feats, offs, P

    def score(self, feats, frame_offsets=None, flags=TSR_EXACT):
        """max_m(detect(E_n, W_m) + c_m) and the winning template id at every frame of
        every utterance -> (scores f32 [sum T], ids i32 [sum T])."""
        feats, offs, P = self._offsets(feats, frame_offsets)
        scores = np.empty(P, np.float32)
        ids = np.empty(P, np.int32)
        check(lib().tsr_score_bank_u8(self.ctx._h, self._h, ptr(feats), ptr(offs), len(offs) - 1,
                                      int(flags), ptr(scores), ptr(ids)))
        return scores, ids

    def score_all(self, feats, frame_offsets=None, flags=TSR_EXACT):
        """detect(E_n, W_m) + c_m for every template -> f32 [M, sum T]."""
        feats, offs, P = self._offsets(feats, frame_offsets)
        out = np.empty((self.M, P), np.float32)
        check(lib().tsr_score_bank_all_u8(self.ctx._h, self._h, ptr(feats), ptr(offs), len(offs) - 1,
                                          int(flags), ptr(out)))
        return out


class Corpus:
    """Utterances resident in HBM together with their score/id arrays (tsr_corpus)."""

    def __init__(self, feats, frame_offsets, D=None, ctx=None):
        self.ctx = ctx or default_ctx()
        feats = np.ascontiguousarray(feats, dtype=np.uint8)
        offs = np.ascontiguousarray(frame_offsets, dtype=np.int64)
        P = int(offs[-1])
        if D is None:
            D = feats.size // max(P, 1)
        if feats.size != P * D:
            raise ValueError("feats has %d bytes, expected %d frames x D=%d" % (feats.size, P, D))
        self.P, self.D, self.N = P, int(D), len(offs) - 1
        self.offsets = offs
        h = C.c_void_p()
        check(lib().tsr_corpus_create(self.ctx._h, ptr(feats), ptr(offs), self.N, self.D, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().tsr_corpus_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @classmethod
    def from_spectrograms(cls, S_cat, frame_offsets, block_length=40, spread_length=1, threshold=.7,
                          abst_threshold=(0.025, 0.025, 0.015, 0.015, 0.02, 0.02, 0.02, 0.02), ctx=None, mask=None):
        """Edge front-end on the device (get_edge_features, get_train_data.py:2547-2557): spectrogram
        stream f64 [sum T, H] (time-major) -> resident binary edge corpus with D = 8 H."""
        ctx = ctx or default_ctx()
        S = np.ascontiguousarray(S_cat, dtype=np.float64)
        offs = np.ascontiguousarray(frame_offsets, dtype=np.int64)
        if S.ndim != 2 or S.shape[0] != int(offs[-1]):
            raise ValueError("S_cat must be [frame_offsets[-1], H]")
        ab = np.ascontiguousarray(abst_threshold, dtype=np.float64)
        if ab.shape != (8,):
            raise ValueError("abst_threshold needs 8 values")
        h = C.c_void_p()
        if mask is not None:   # u8 [sum T, H]: edge strengths are zeroed where it is 0 (magnitude_thresholded_edges)
            mk = np.ascontiguousarray(mask, dtype=np.uint8)
            if mk.shape != S.shape:
                raise ValueError("mask must have the spectrogram stream's shape %r" % (S.shape,))
            check(lib().tsr_edge_corpus_create_masked(ctx._h, ptr(S), ptr(mk), ptr(offs), len(offs) - 1, S.shape[1],
                                                      int(block_length), int(spread_length), float(threshold), ptr(ab),
                                                      C.byref(h)))
            return cls._adopt(h, ctx)
        check(lib().tsr_edge_corpus_create(ctx._h, ptr(S), ptr(offs), len(offs) - 1, S.shape[1], int(block_length),
                                           int(spread_length), float(threshold), ptr(ab), C.byref(h)))
        return cls._adopt(h, ctx)

    @classmethod
    def _adopt(cls, handle, ctx):
        """Wrap a tsr_corpus* produced on the device (no upload)."""
        self = cls.__new__(cls)
        self.ctx, self._h = ctx, handle
        n, d = C.c_int64(), C.c_int64()
        check(lib().tsr_corpus_shape(handle, C.byref(n), C.byref(d), None))
        self.N, self.D = int(n.value), int(d.value)
        self.offsets = np.empty(self.N + 1, np.int64)
        check(lib().tsr_corpus_shape(handle, None, None, ptr(self.offsets)))
        self.P = int(self.offsets[-1])
        return self

    def part_features(self, parts, threshold, spread=(0, 0), graph=None, grid=(1, 1), fast=False):
        """Edge-map corpus -> part-map corpus on the device (code_parts -> argmax -> spread ->
        [graph] -> [::g0, ::g1] per utterance; get_train_data.py:2619-2628 / :183-191)."""
        g = None if graph is None else np.ascontiguousarray(graph, dtype=np.uint8)
        h = C.c_void_p()
        check(lib().tsr_corpus_part_features(self.ctx._h, parts._h, self._h, int(threshold), int(bool(fast)),
                                             int(spread[0]), int(spread[1]), ptr(g), int(grid[0]),
                                             int(grid[1]), C.byref(h)))
        return Corpus._adopt(h, self.ctx)

    def layout(self):
        """What is resident: {"bytes": bool, "bits": bool, "feature_bytes": device bytes of those forms}."""
        lay, nb = C.c_int(), C.c_int64()
        check(lib().tsr_corpus_layout(self._h, C.byref(lay), C.byref(nb)))
        return {"bytes": bool(lay.value & 1), "bits": bool(lay.value & 2), "feature_bytes": int(nb.value)}

    def download_feats(self):
        """The resident feature bytes as u8 [sum T, D] (0/1)."""
        out = np.empty((self.P, self.D), np.uint8)
        check(lib().tsr_corpus_download_feats(self.ctx._h, self._h, ptr(out)))
        return out

    def score(self, bank, flags=TSR_EXACT):
        """Enqueue one scoring pass on the ctx stream (asynchronous)."""
        check(lib().tsr_corpus_score(self.ctx._h, bank._h, self._h, int(flags)))

    def classify_windows(self, bank, starts, ends, flags=TSR_EXACT):
        """Best (score, stream frame, template) per phone window [start, end) of the frame stream
        (get_train_data.py:718-773) -> (f32 [W], i64 [W], i32 [W])."""
        starts = np.ascontiguousarray(starts, dtype=np.int64)
        ends = np.ascontiguousarray(ends, dtype=np.int64)
        if starts.shape != ends.shape or starts.ndim != 1:
            raise ValueError("starts/ends must be equal-length 1-d arrays")
        W = starts.shape[0]
        score, loc, tid = np.empty(W, np.float32), np.empty(W, np.int64), np.empty(W, np.int32)
        check(lib().tsr_corpus_classify_windows(self.ctx._h, bank._h, self._h, ptr(starts), ptr(ends), W,
                                                int(flags), ptr(score), ptr(loc), ptr(tid)))
        return score, loc, tid

    def classify_isolated(self, bank, bgd, first, real_lo, real_hi, rows, n_valid):
        """Isolated-phone classification (get_train_data.py:781-925) with virtual windows: see
        tsr_corpus_classify_isolated -> (score f32 [W], position inside the window i32 [W], template i32 [W])."""
        bgd = np.ascontiguousarray(bgd, dtype=np.float32).reshape(-1)
        if bgd.size != self.D:
            raise ValueError("bgd has %d values, the corpus D=%d" % (bgd.size, self.D))
        a = [np.ascontiguousarray(x, dtype=np.int64) for x in (first, real_lo, real_hi)]
        b = [np.ascontiguousarray(x, dtype=np.int32) for x in (rows, n_valid)]
        W = a[0].shape[0]
        if any(x.shape != (W,) for x in a + b):
            raise ValueError("per-window arrays must be 1-d and equally long")
        score, loc, tid = np.empty(W, np.float32), np.empty(W, np.int32), np.empty(W, np.int32)
        check(lib().tsr_corpus_classify_isolated(self.ctx._h, bank._h, self._h, ptr(bgd), ptr(a[0]), ptr(a[1]), ptr(a[2]),
                                                 ptr(b[0]), ptr(b[1]), W, ptr(score), ptr(loc), ptr(tid)))
        return score, loc, tid

    def score_all(self, bank, flags=TSR_EXACT):
        """detect + c of every template at every frame of the resident corpus -> f32 [M, P]."""
        out = np.empty((bank.M, self.P), np.float32)
        check(lib().tsr_corpus_score_all(self.ctx._h, bank._h, self._h, int(flags), ptr(out)))
        return out

    def weighted_sums(self, weights):
        """sum_p weights[p, k] * x[p, :] for every column k of `weights` f32 [P, K] -> f32 [K, D]
        (the M-step product of the Bernoulli mixture, bernoulli_mixture.py:179)."""
        w = np.ascontiguousarray(weights, dtype=np.float32)
        if w.ndim != 2 or w.shape[0] != self.P:
            raise ValueError("weights must be [P, K]")
        out = np.empty((w.shape[1], self.D), np.float32)
        check(lib().tsr_corpus_weighted_sums(self.ctx._h, self._h, ptr(w), w.shape[1], ptr(out)))
        return out

    def download(self):
        scores = np.empty(self.P, np.float32)
        ids = np.empty(self.P, np.int32)
        check(lib().tsr_corpus_download(self.ctx._h, self._h, ptr(scores), ptr(ids)))
        return scores, ids


def _rows(vals, offsets):
    vals = np.ascontiguousarray(vals, dtype=np.float32).reshape(-1)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    if offsets.ndim != 1 or offsets.size < 1 or vals.size != int(offsets[-1]):
        raise ValueError("vals must hold offsets[-1] scores")
    return vals, offsets


def peaks(threshold, C0, corpus=None, vals=None, offsets=None, ctx=None):
    """get_max_above_threshold (roc_functions.py:177-187) for every utterance, on the scores resident
    in `corpus` or on host rows (vals, offsets) -> list of ascending index arrays (int32)."""
    if corpus is not None:
        c, h, v, o, N = corpus.ctx, corpus._h, None, corpus.offsets, corpus.N
    else:
        c = ctx or default_ctx()
        v, o = _rows(vals, offsets)
        h, N = None, len(o) - 1
    counts = np.zeros(N, np.int64)
    times = np.empty(int(o[-1]), np.int32)
    check(lib().tsr_peaks(c._h, h, ptr(v), ptr(o), N, float(threshold), int(C0), ptr(counts), ptr(times)))
    return [times[o[n]:o[n] + counts[n]].copy() for n in range(N)]


def detect_clusters(threshold, C0, C1, corpus=None, vals=None, offsets=None, ctx=None, return_peaks=False):
    """Per utterance the clustered detections above `threshold` (roc_functions.py:356-383 with
    cluster_times.pyx:133-222) -> tuple over utterances of tuples of (start, end)."""
    if corpus is not None:
        c, h, v, o, N = corpus.ctx, corpus._h, None, corpus.offsets, corpus.N
    else:
        c = ctx or default_ctx()
        v, o = _rows(vals, offsets)
        h, N = None, len(o) - 1
    nc = np.zeros(N, np.int64)
    total = int(o[-1])
    starts, ends = np.empty(total, np.int32), np.empty(total, np.int32)
    pc = np.zeros(N, np.int64) if return_peaks else None
    pt = np.empty(total, np.int32) if return_peaks else None
    check(lib().tsr_detect_clusters(c._h, h, ptr(v), ptr(o), N, float(threshold), float(C0), float(C1), ptr(nc),
                                    ptr(starts), ptr(ends), ptr(pc), ptr(pt)))
    clusters = tuple(tuple((int(starts[o[n] + k]), int(ends[o[n] + k])) for k in range(int(nc[n]))) for n in range(N))
    if return_peaks:
        return clusters, [pt[o[n]:o[n] + pc[n]].copy() for n in range(N)]
    return clusters


def window_peak_max(utt, starts, ends, corpus=None, vals=None, offsets=None, ctx=None):
    """Per-window largest local maximum (roc_functions.py:107-116) -> f32 [W] (-inf where none)."""
    if corpus is not None:
        c, h, v, o, N = corpus.ctx, corpus._h, None, corpus.offsets, corpus.N
    else:
        c = ctx or default_ctx()
        v, o = _rows(vals, offsets)
        h, N = None, len(o) - 1
    a = [np.ascontiguousarray(x, dtype=np.int64) for x in (utt, starts, ends)]
    W = a[0].shape[0]
    out = np.empty(W, np.float32)
    check(lib().tsr_window_peak_max(c._h, h, ptr(v), ptr(o), N, ptr(a[0]), ptr(a[1]), ptr(a[2]), W, ptr(out)))
    return out


class Parts:
    """Device-resident part models of the patch coder (tsr_parts): logParts /
    logInvParts of get_train_data.PartsParameters, f64 [K, P0, P1, E]."""

    def __init__(self, log_parts, log_invparts, ctx=None):
        self.ctx = ctx or default_ctx()
        lp = np.ascontiguousarray(log_parts, dtype=np.float64)
        lip = np.ascontiguousarray(log_invparts, dtype=np.float64)
        if lp.ndim != 4 or lip.shape != lp.shape:
            raise ValueError("log_parts/log_invparts must be [K, P0, P1, E] and equal-shaped")
        self.K, self.P0, self.P1, self.E = lp.shape
        h = C.c_void_p()
        check(lib().tsr_parts_create(self.ctx._h, ptr(lp), ptr(lip), *lp.shape, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().tsr_parts_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_mode(self, mode):
        """Coder of Corpus.part_features: TSR_PARTS_AUTO (MFMA when the tables fit), _VALU, _MFMA."""
        check(lib().tsr_parts_set_mode(self._h, int(mode)))

    def info(self):
        """MFMA coder availability and the f64 re-evaluation count of the latest corpus call."""
        ok, n, r = C.c_int(), C.c_int64(), C.c_int64()
        check(lib().tsr_parts_info(self._h, C.byref(ok), C.byref(n), C.byref(r)))
        return dict(mfma_available=bool(ok.value), locations=int(n.value), rechecked=int(r.value))

    def features(self, X, threshold, spread=(0, 0), graph=None, grid=(1, 1), fast=False,
                 return_argmax=False):
        """code_parts[_fast] -> argmax -> spread -> [graph] -> [::g0, ::g1] on the GPU
        (get_train_data.py:2619-2628 / :183-191) -> u8 [O0, O1, K]."""
        X = np.ascontiguousarray(X, dtype=np.uint8)
        if X.ndim != 3 or X.shape[2] != self.E:
            raise ValueError("X must be [X0, X1, %d]" % self.E)
        N0, N1 = X.shape[0] - self.P0 + 1, X.shape[1] - self.P1 + 1
        g0, g1 = int(grid[0]), int(grid[1])
        out = np.empty((-(-N0 // g0), -(-N1 // g1), self.K), np.uint8)
        am = np.empty((N0, N1), np.int64) if return_argmax else None
        g = None if graph is None else np.ascontiguousarray(graph, dtype=np.uint8)
        if g is not None and g.shape != (self.K, self.K):
            raise ValueError("part graph must be [K, K]")
        check(lib().tsr_part_features(self.ctx._h, self._h, ptr(X), X.shape[0], X.shape[1],
                                      int(threshold), int(bool(fast)), int(spread[0]), int(spread[1]),
                                      ptr(g), g0, g1, ptr(out), ptr(am)))
        return (out, am) if return_argmax else out
