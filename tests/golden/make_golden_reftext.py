#!/usr/bin/env python3
"""Golden vectors from the reference's own PYTHON code (not a restatement): g18 = edge front-end
(edge_signal_proc._edge_map_no_threshold + _edge_map_threshold_segments + reorg_part_for_fast_filtering,
get_train_data.py:2547-2557), g19 = roc_functions.get_max_above_threshold /
get_max_detection_in_syllable_windows / _get_detect_clusters_single_threshold over the compiled cluster_times.pyx.
The Python-2 sources are executed where they lie through oracle/py2ref.py (lib2to3 + classic division + the
boolean unary minus of the NumPy of that time).  Build container only:
    python tests/golden/make_golden_reftext.py
"""
import itertools
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import build_ref  # noqa: E402
import py2ref  # noqa: E402

assert build_ref.build(), "oracle/_ref could not be built (no /root/reference?)"
ABST = np.array([0.025, 0.025, 0.015, 0.015, 0.02, 0.02, 0.02, 0.02])

esp = py2ref.load("edge_signal_proc", ["_edge_map_no_threshold", "_edge_map_threshold_segments"])
gtd = py2ref.load("get_train_data", ["reorg_part_for_fast_filtering"])
rng = np.random.default_rng(1800)
d = {}
cases = [(45, 200, 40, 1, .7), (20, 93, 40, 3, .7), (12, 57, 25, 0, .5), (5, 41, 41, 2, .9), (30, 64, 16, 1, .3), (9, 4, 4, 1, .7)]
for i, (H, T, bl, sl, q) in enumerate(cases):
    S = rng.standard_normal((H, T)) * (0.05 if i == 2 else 1.0)
    if i in (1, 4):
        S = np.round(S * 4) / 4                       # ties at the quantile
    E, breaks, orient = esp["_edge_map_no_threshold"](S.copy())
    esp["_edge_map_threshold_segments"](E, bl, sl, threshold=q, edge_orientations=orient.astype(int),
                                        edge_feature_row_breaks=breaks.astype(int), abst_threshold=ABST)
    d["S%d" % i], d["p%d" % i] = S, np.array([bl, sl, q])
    d["out%d" % i] = np.asarray(gtd["reorg_part_for_fast_filtering"](E)).astype(np.uint8)
d["n"] = np.int64(len(cases))
path = os.path.join(HERE, "g18_edge_features_reftext.npz")
np.savez_compressed(path, **d)
print("g18_edge_features_reftext %.1f KB" % (os.path.getsize(path) / 1024))

roc = py2ref.load("roc_functions", ["get_max_above_threshold", "get_max_detection_in_syllable_windows",
                                    "_get_detect_clusters_single_threshold"],
                  env={"itertools": itertools, "cluster_times": build_ref.load_extra("cluster_times")})
rng = np.random.default_rng(1900)
d = {}
n_cases = 12
for i in range(n_cases):
    N, T = int(rng.integers(1, 6)), int(rng.integers(20, 160))
    arr = rng.standard_normal((N, T)).astype(np.float32)
    if i % 3 == 0:
        arr = np.round(arr * 2) / 2
    C0, C1 = int(rng.integers(2, 9)), int(rng.integers(3, 40))
    lengths = rng.integers(C0 + 4, T + 1, size=N)
    thr = float(rng.choice([-1.0, 0.0, 0.5, 1.5]))
    clusters = roc["_get_detect_clusters_single_threshold"](thr, arr, lengths, C0, C1)
    windows = [[(int(s), int(s) + int(rng.integers(1, 30))) for s in rng.integers(0, T, size=int(rng.integers(0, 4)))]
               for _ in range(N)]
    ws, we = -int(rng.integers(0, 10)), int(rng.integers(3, 25))
    vals, order, utt = roc["get_max_detection_in_syllable_windows"](arr, windows, lengths, ws, we, return_argsort_idx=True)
    d["a%d" % i], d["l%d" % i], d["p%d" % i] = arr, lengths, np.array([thr, C0, C1, ws, we])
    for n in range(N):
        d["pk%d_%d" % (i, n)] = roc["get_max_above_threshold"](arr[n], int(lengths[n]), thr, C0)
        d["cl%d_%d" % (i, n)] = np.array(clusters[n], np.int64).reshape(-1, 2)
        d["w%d_%d" % (i, n)] = np.array(windows[n], np.int64).reshape(-1, 2)
    d["wv%d" % i], d["wo%d" % i], d["wu%d" % i] = vals, order, utt
d["n"] = np.int64(n_cases)
path = os.path.join(HERE, "g19_roc_functions_reftext.npz")
np.savez_compressed(path, **d)
print("g19_roc_functions_reftext %.1f KB" % (os.path.getsize(path) / 1024))

# g20: classification.Classifier (classification.py:26-149, template and collection types) as the reference runs it
tt_f = py2ref.load("test_template", ["score_template_background_section"])


class _NS(object):
    pass
tt = _NS()
tt.score_template_background_section = tt_f["score_template_background_section"]
pm = _NS()
pm.TwoPartModel = type("TwoPartModel", (), {})
cls = py2ref.load("classification", ["Classifier"], env={"tt": tt, "pm": pm})
rng = np.random.default_rng(2000)
d = {}
n_cases = 6
METHODS = ("score", "score_register", "score_no_bg", "coarse_score_like", "coarse_score_like_no_bg", "coarse_score_count")
for i in range(n_cases):
    D, L = int(rng.integers(6, 40)), int(rng.integers(6, 24)) * 2
    bg = rng.uniform(.05, .4, size=D)
    tmpl = rng.uniform(.02, .98, size=(D, L))
    if i % 2:   # a collection; every other one with templates of different lengths (the coarse likelihoods of the
        # reference assert equal lengths, so those cases record NaN for them)
        cut = (2, 4) if i % 4 == 1 else (0, 0)
        base = [tmpl, rng.uniform(.02, .98, size=(D, L - cut[0])), rng.uniform(.02, .98, size=(D, L - cut[1]))]
    else:
        base = tmpl
    clf = cls["Classifier"](base, coarse_factor=2, bg=bg)
    W = (rng.random((4, D, L)) < .3).astype(np.uint8)
    d["bg%d" % i], d["W%d" % i], d["nt%d" % i] = bg, W, np.int64(len(base) if i % 2 else 0)
    for k, t in enumerate(base if i % 2 else [base]):
        d["t%d_%d" % (i, k)] = t
    for m in METHODS:
        try:
            d["%s%d" % (m, i)] = np.array([float(getattr(clf, m)(w, bg) if m in ("score", "score_register", "coarse_score_like")
                                                 else getattr(clf, m)(w)) for w in W])
        except (AssertionError, IndexError):
            d["%s%d" % (m, i)] = np.full(len(W), np.nan)
    d["meta%d" % i] = np.array([clf.coarse_length, clf.window[0], clf.window[1]])
d["n"] = np.int64(n_cases)
path = os.path.join(HERE, "g20_classifier_reftext.npz")
np.savez_compressed(path, **d)
print("g20_classifier_reftext %.1f KB" % (os.path.getsize(path) / 1024))

# g21: bernoulli_mixture.BernoulliMixture (bernoulli_mixture.py:9-300) as the reference runs it: planted components,
# init, first E-step, 12 EM iterations (fixed count: tol = 0 with max_iter = min_iter = 12)
import contextlib  # noqa: E402
import io  # noqa: E402
import random  # noqa: E402

bmc = py2ref.load("bernoulli_mixture", ["BernoulliMixture"], env={"random": random})
rng = np.random.default_rng(2100)
d = {}
n_cases = 3
for i in range(n_cases):
    N, shape, K = int(rng.integers(150, 400)), (int(rng.integers(3, 7)), int(rng.integers(2, 6)), 8), int(rng.integers(2, 5))
    protos = rng.uniform(0.05, 0.95, size=(K,) + shape)
    data = (rng.random((N,) + shape) < protos[rng.integers(K, size=N)]).astype(np.uint8)
    with contextlib.redirect_stdout(io.StringIO()):                    # (the class prints its progress)
        mix = bmc["BernoulliMixture"](K, data, min_probability=0.05)
        d["init_logt%d" % i] = np.asarray(mix.log_templates, np.float64)
        d["tll%d" % i] = np.asarray(mix.get_template_loglikelihoods(), np.float64)
        mix.hard_assignment = False                                       # (run_EM sets it; the E-step reads it)
        ll0 = mix._compute_loglikelihoods()
        d["aff0_%d" % i] = np.asarray(mix.affinities, np.float64)
        mix.run_EM(0.0, min_probability=0.05, max_iter=12, min_iter=12)
    d["data%d" % i], d["K%d" % i] = data, np.int64(K)
    d["templates%d" % i], d["weights%d" % i] = np.asarray(mix.templates, np.float64), np.asarray(mix.weights, np.float64)
    mix.reset_templates()                  # (run_EM leaves the log tables in data shape; the E-step wants them flat)
    d["ll0_%d" % i], d["ll_%d" % i] = np.float64(ll0), np.float64(mix._compute_loglikelihoods())
    d["iters%d" % i] = np.int64(mix.iterations)
d["n"] = np.int64(n_cases)
path = os.path.join(HERE, "g21_bernoulli_mixture_reftext.npz")
np.savez_compressed(path, **d)
print("g21_bernoulli_mixture_reftext %.1f KB" % (os.path.getsize(path) / 1024))

# g22: get_train_data._compute_detection_E (:927-1012) with its own signature, over the compiled detect: a few
# utterances appended one after the other (rows, template ids, example times, detect lengths)
cllf, _, _ = build_ref.load()
gtd2 = py2ref.load("get_train_data", ["_compute_detection_E"],
                   env={"itertools": itertools, "compute_likelihood_linear_filter": cllf})
rng = np.random.default_rng(2200)
d = {}
n_cases = 4
for i in range(n_cases):
    F, K = int(rng.integers(3, 9)), int(rng.integers(2, 9))
    Ls = [int(x) for x in rng.integers(4, 20, size=(1 if i == 0 else int(rng.integers(2, 7))))]
    bank = tuple(((rng.standard_normal((L, F, K)) * .7).astype(np.float32), float(rng.standard_normal() * 3)) for L in Ls)
    nutt = 3
    Tmax = 90
    det, ids = np.zeros((nutt, Tmax + 2), np.float32), np.zeros((nutt, Tmax + 2), np.int32)
    lengths, examples = [], []
    syl = ("aa", "b")
    with contextlib.redirect_stdout(io.StringIO()):
        for u in range(nutt):
            nph = int(rng.integers(3, 9))
            phns = rng.choice(np.array(["aa", "b", "k"]), size=nph)
            flts = np.concatenate(([0], np.cumsum(rng.integers(5, 11, size=nph)))).astype(np.int64)
            T = int(flts[-1])
            E = (rng.random((T, F, K)) < .3).astype(np.uint8)
            gtd2["_compute_detection_E"](E, phns, flts, det, ids, lengths, bank, syl, examples)
            d["E%d_%d" % (i, u)], d["phns%d_%d" % (i, u)], d["flts%d_%d" % (i, u)] = E, phns, flts
            d["ex%d_%d" % (i, u)] = np.array(list(examples[-1]), np.int64).reshape(-1, 2)
    for m, (W, c) in enumerate(bank):
        d["W%d_%d" % (i, m)], d["c%d_%d" % (i, m)] = W, np.float64(c)
    d["M%d" % i], d["det%d" % i], d["ids%d" % i], d["len%d" % i] = np.int64(len(bank)), det, ids, np.array(lengths, np.int64)
d["n"] = np.int64(n_cases)
path = os.path.join(HERE, "g22_compute_detection_E_reftext.npz")
np.savez_compressed(path, **d)
print("g22_compute_detection_E_reftext %.1f KB" % (os.path.getsize(path) / 1024))
