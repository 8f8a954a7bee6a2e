"""Bank preparation (host side, NumPy): the part of the reference's
estimate_template.py that turns Bernoulli templates into the (W, c) pairs the
scoring kernels consume.  construct_linear_filter: estimate_template.py:230-251;
construct_linear_filters: :156-172 (Bernoulli branch; the Gaussian/"spectral"
branch is outside the template-scoring path)."""
import numpy as np


def construct_linear_filter(T, bgd, min_prob=.01, all_cs=False, use_spectral=False, sigma=None):
    """Log-odds filter of a Bernoulli template T [L, ...] against the background
    bgd [...]:  W = log( (T/B) / ((1-T)/(1-B)) ) as float32 and the constant
    c = sum log((1-T)/(1-B)) in float64 (with all_cs: its cumulative sum over
    template frames, one entry per template length)."""
    if use_spectral:
        raise NotImplementedError("spectral (Gaussian) filters are outside the template-scoring path")
    T = np.clip(np.asarray(T, dtype=np.float64), min_prob, 1 - min_prob)
    B = np.broadcast_to(np.asarray(bgd, dtype=np.float64), T.shape)
    neg = (1. - T) / (1. - B)
    log_neg = np.log(neg)
    if all_cs:
        c = np.cumsum(log_neg.reshape(len(log_neg), -1).sum(1))
    else:
        c = log_neg.sum()
    return np.log((T / B) / neg).astype(np.float32), c


def construct_linear_filters(Ts, bgd, all_cs=False, use_spectral=False, T_sigmas=None,
                             bgd_sigma=None, min_prob=.01):
    """One (W, c) pair per template; templates and background are clipped to
    [min_prob, 1-min_prob] first."""
    if use_spectral:
        raise NotImplementedError("spectral (Gaussian) filters are outside the template-scoring path")
    b = np.clip(bgd, min_prob, 1 - min_prob)
    return tuple(construct_linear_filter(np.clip(T, min_prob, 1 - min_prob), b, all_cs=all_cs) for T in Ts)
