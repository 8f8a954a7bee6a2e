"""Seeded synthetic workloads for the template-scoring path (NumPy only).

Generators follow SURVEY.md section 8(d): binary edge/part feature maps drawn
per feature from Bernoulli(q_d) with q_d ~ U(0.05, 0.40) (the background range
the reference clips to, test_template.py:131-135), Bernoulli templates
P ~ U(0.05, 0.95) turned into log-odds filters exactly as
estimate_template.py:230-251 does, and part models P ~ U(0.01, 0.99) for the
5x5x8 patch coder (code_parts.pyx:70-135).  The same code runs in this
container and on the GPU box, so tests, golden fixtures and bench agree.
"""
import numpy as np


def background(rng, shape):
    """Per-feature background edge probability q_d ~ U(0.05, 0.40)."""
    return rng.uniform(0.05, 0.40, size=shape)


def features(rng, T, q):
    """u8 [T, *q.shape] with X[t, d] ~ Bernoulli(q_d)."""
    return (rng.random((T,) + q.shape) < q).astype(np.uint8)


def linear_filter(P, bgd, min_prob=0.01):
    """Log-odds filter of one Bernoulli template against a background
    (same arithmetic as estimate_template.py:230-251): W f32, c f64."""
    P = np.clip(np.asarray(P, np.float64), min_prob, 1 - min_prob)
    B = np.broadcast_to(np.clip(np.asarray(bgd, np.float64), min_prob, 1 - min_prob), P.shape)
    inv_ratio = (1.0 - P) / (1.0 - B)
    W = np.log((P / B) / inv_ratio).astype(np.float32)
    return W, float(np.log(inv_ratio).sum())


def bank(rng, q, lengths):
    """Template bank ((W_m f32 [L_m, *q.shape], c_m f64), ...) for background q."""
    out = []
    for L in lengths:
        P = rng.uniform(0.05, 0.95, size=(int(L),) + q.shape)
        out.append(linear_filter(P, q))
    return tuple(out)


def parts(rng, K, patch=(5, 5, 8)):
    """Part models: (log_parts, log_invparts) f64 [K, 5, 5, 8]."""
    P = rng.uniform(0.01, 0.99, size=(K,) + tuple(patch))
    return np.log(P), np.log(1.0 - P)


def corpus(seed, N, T, F=45, E=8, M=8, L=32):
    """One BASELINE config: returns dict(feats u8 [sum T, F, E], offsets i64 [N+1],
    bank, q).  T and L may be ints or per-item sequences."""
    rng = np.random.default_rng(seed)
    q = background(rng, (F, E))
    Ts = np.full(N, T, np.int64) if np.isscalar(T) else np.asarray(T, np.int64)
    Ls = [L] * M if np.isscalar(L) else list(L)
    bk = bank(rng, q, Ls)
    offsets = np.concatenate(([0], np.cumsum(Ts))).astype(np.int64)
    feats = (rng.random((int(offsets[-1]),) + q.shape) < q).astype(np.uint8)
    return dict(feats=feats, offsets=offsets, bank=bk, q=q)
