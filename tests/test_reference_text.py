"""The CPU restatements of the reference's Python-2 glue against the reference's OWN code.

Runs only where /root/reference is mounted (this container): oracle/py2ref.py reads the reference's modules where
they lie, rewrites the Python-2-only syntax with lib2to3, emulates classic `/` and the boolean unary minus of the
NumPy of that time, and executes just the functions named here (their Cython dependencies come from the compiled
reference in oracle/_ref).  Everything below therefore compares a restatement in oracle/oracle.py -- which is what
the GPU parity tests and the committed golden vectors rest on -- with the reference's text itself on fresh random
inputs.  On the GPU box (no reference) the tests skip; the golden vectors made the same way
(tests/golden/make_golden_reftext.py) travel instead.
"""
import itertools
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import py2ref  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.isdir(py2ref.REFERENCE), reason="the reference is not mounted here")


@pytest.fixture(scope="module")
def oracle():
    import oracle as o
    o.build()
    return o


@pytest.fixture(scope="module")
def ref_modules():
    import build_ref
    if not build_ref.build():
        pytest.skip("oracle/_ref could not be built")
    cllf, cp, swp = build_ref.load()
    return {"compute_likelihood_linear_filter": cllf, "code_parts": cp, "spread_waliji_patches": swp,
            "cluster_times": build_ref.load_extra("cluster_times")}


def test_edge_front_end_restatement_is_the_references_code(oracle):
    """edge_signal_proc._edge_map_no_threshold / _edge_map_threshold_segments + reorg_part_for_fast_filtering
    (get_train_data.py:2547-2557) vs oracle.get_edge_features: identical edge maps."""
    esp = py2ref.load("edge_signal_proc", ["_edge_map_no_threshold", "_edge_map_threshold_segments"])
    gtd = py2ref.load("get_train_data", ["reorg_part_for_fast_filtering"])
    rng = np.random.default_rng(1801)
    for case in range(60):
        H, T = int(rng.integers(4, 50)), int(rng.integers(4, 260))
        S = rng.standard_normal((H, T)) * rng.choice([1.0, 0.01, 5.0])
        if case % 5 == 0:
            S = np.round(S * 4) / 4                                   # ties at the quantile
        bl, sl, q = min(int(rng.integers(5, 60)), T), int(rng.integers(0, 4)), float(rng.choice([.3, .5, .7, .9]))
        E, breaks, orient = esp["_edge_map_no_threshold"](S.copy())
        assert np.array_equal(E, oracle.edge_map_no_threshold(S))
        # (the NumPy of the reference's time took these integral floats as slice bounds)
        esp["_edge_map_threshold_segments"](E, bl, sl, threshold=q, edge_orientations=orient.astype(int),
                                            edge_feature_row_breaks=breaks.astype(int),
                                            abst_threshold=oracle.ABST_THRESHOLD)
        want = np.asarray(gtd["reorg_part_for_fast_filtering"](E)).astype(np.uint8)
        assert np.array_equal(want, oracle.get_edge_features(S, bl, sl, q)), case


def test_peak_picking_and_clustering_restatements_are_the_references_code(oracle, ref_modules):
    """roc_functions.get_max_above_threshold (:177-187), get_max_detection_in_syllable_windows (:88-139),
    get_cluster_starts_ends (:189-210), _get_detect_clusters_single_threshold (:356-383) over the compiled
    cluster_times.pyx."""
    roc = py2ref.load("roc_functions", ["get_max_above_threshold", "get_max_detection_in_syllable_windows",
                                        "_get_detect_clusters_single_threshold"],
                      env={"itertools": itertools, "cluster_times": ref_modules["cluster_times"]})
    rng = np.random.default_rng(1802)
    for case in range(60):
        N, T = int(rng.integers(1, 7)), int(rng.integers(12, 200))
        arr = rng.standard_normal((N, T)).astype(np.float32)
        if case % 4 == 0:
            arr = np.round(arr * 2) / 2                               # plateaus
        lengths = rng.integers(10, T + 1, size=N)
        C0, C1 = int(rng.integers(2, 9)), int(rng.integers(3, 40))
        thr = float(rng.choice([-1.0, 0.0, 0.5, 1.5]))
        for n in range(N):
            if lengths[n] - C0 + 1 >= 3:
                assert np.array_equal(roc["get_max_above_threshold"](arr[n], int(lengths[n]), thr, C0),
                                      oracle.get_max_above_threshold(arr[n], int(lengths[n]), thr, C0))
        ok = [n for n in range(N) if lengths[n] - C0 + 1 >= 3]
        if ok:
            want = roc["_get_detect_clusters_single_threshold"](thr, arr[ok], lengths[ok], C0, C1)
            got = oracle.get_detect_clusters_single_threshold(thr, arr[ok], lengths[ok], C0, C1)
            assert want == got, case
        windows = tuple(tuple((int(s), int(s) + int(rng.integers(1, 30))) for s in rng.integers(0, T, size=int(rng.integers(0, 4))))
                        for _ in range(N))
        ws, we = -int(rng.integers(0, 10)), int(rng.integers(3, 25))
        want = roc["get_max_detection_in_syllable_windows"](arr, windows, lengths, ws, we, return_argsort_idx=True)
        got = oracle.get_max_detection_in_syllable_windows(arr, windows, lengths, ws, we, return_argsort_idx=True)
        for a, b in zip(want, got):
            assert np.array_equal(a, b), case


def _bank(rng, M, F, K, Ls):
    """(W f32, c) pairs; c as Python floats: with the NumPy of the reference's time an f64 SCALAR added to an f32
    array left it f32 (value-based casting); NumPy 2 keeps that only for Python scalars."""
    return tuple(((rng.standard_normal((L, F, K)) * .7).astype(np.float32), float(rng.standard_normal() * 3)) for L in Ls)


def test_bank_epilogue_restatement_is_the_references_code(oracle, ref_modules):
    """get_train_data._compute_detection_E (:927-992), the hot path's bank epilogue, over the compiled detect."""
    gtd = py2ref.load("get_train_data", ["_compute_detection_E"],
                      env={"itertools": itertools,
                           "compute_likelihood_linear_filter": ref_modules["compute_likelihood_linear_filter"]})
    rng = np.random.default_rng(1803)
    for case in range(40):
        T, F, K = int(rng.integers(50, 200)), int(rng.integers(2, 10)), int(rng.integers(1, 9))
        Ls = [int(x) for x in rng.integers(5, 49, size=int(rng.integers(1, 9)))]
        E = (rng.random((T, F, K)) < rng.uniform(.05, .5)).astype(np.uint8)
        bank = _bank(rng, len(Ls), F, K, Ls)
        if case % 3 == 0 and len(bank) > 2:
            bank = bank[:2] + (bank[0],) + bank[2:]                     # a duplicated template: ties
        det = np.zeros((1, T + 7), np.float32)
        ids = np.zeros((1, T + 7), np.int32)
        lengths = []
        phns = np.array(["a", "b", "c"])
        flts = np.array([0, T // 3, 2 * T // 3, T])
        gtd["_compute_detection_E"](E, phns, flts, det, ids, lengths, bank, ("zz",), [])
        row, rid = oracle.compute_detection_E(E, bank)
        assert np.array_equal(det[0, :T], row) and np.array_equal(ids[0, :T], rid), case


def test_classification_glue_restatement_is_the_references_code(oracle, ref_modules):
    """get_train_data._compute_classification_E (:718-773) and _compute_isolated_classification_E (:836-925)
    over the compiled detect / detect_float."""
    env = {"itertools": itertools, "compute_likelihood_linear_filter": ref_modules["compute_likelihood_linear_filter"]}
    gtd = py2ref.load("get_train_data", ["_compute_classification_E", "_compute_isolated_classification_E"], env=env)
    rng = np.random.default_rng(1804)
    for case in range(30):
        F, K = int(rng.integers(2, 8)), int(rng.integers(1, 6))
        nph = int(rng.integers(2, 7))
        durs = rng.integers(9, 30, size=nph)
        flts = np.concatenate(([0], np.cumsum(durs))).astype(np.int64)
        T = int(flts[-1])
        E = (rng.random((T, F, K)) < .3).astype(np.uint8)
        Ls = [int(x) for x in rng.integers(3, 9, size=int(rng.integers(1, 5)))]
        bank = _bank(rng, len(Ls), F, K, Ls)
        phns = np.array(["p%d" % i for i in range(nph)])
        want = [np.zeros((2, nph), dt) for dt in (np.float32, np.int32, np.int32, np.int32)]
        got = [a.copy() for a in want]
        gtd["_compute_classification_E"](E, phns, flts.copy(), 1, want[0], want[1], want[2], want[3], bank)
        oracle.compute_classification_E(E, phns, flts.copy(), 1, got[0], got[1], got[2], got[3], bank)
        for a, b in zip(want, got):
            assert np.array_equal(a, b), case
        bgd = rng.uniform(.05, .4, size=(F, K)).astype(np.float32)
        want = [np.zeros((2, nph), dt) for dt in (np.float32, np.int32, np.int32, np.int32)]
        got = [a.copy() for a in want]
        gtd["_compute_isolated_classification_E"](E, phns, flts.copy(), 1, want[0], want[1], want[2], want[3], bank, bgd)
        oracle.compute_isolated_classification_E(E, phns, flts.copy(), 1, got[0], got[1], got[2], got[3], bank, bgd)
        for a, b in zip(want, got):
            assert np.array_equal(a, b), case


class _Params(object):
    mag_data = False
    auxiliary_data = False


def test_part_feature_glue_restatement_is_the_references_code(oracle, ref_modules):
    """get_train_data.get_part_features (:2576-2631, edge branch) over the compiled code_parts / spread_patches /
    spread_graph_patches, and reorg / collapse_to_grid."""
    gtd = py2ref.load("get_train_data", ["get_part_features", "collapse_to_grid", "reorg_part_for_fast_filtering"],
                      env={"cp": ref_modules["code_parts"]})
    rng = np.random.default_rng(1805)
    for case in range(25):
        T, F, K = int(rng.integers(8, 60)), int(rng.integers(6, 30)), int(rng.integers(2, 20))
        E = (rng.random((T, F, 8)) < rng.uniform(.1, .4)).astype(np.uint8)
        P = rng.uniform(.05, .95, size=(K, 5, 5, 8))
        par = _Params()
        par.logParts, par.logInvParts = np.log(P), np.log(1 - P)
        par.bernsteinEdgeThreshold = int(rng.integers(0, 30))
        par.spreadRadiusX, par.spreadRadiusY, par.numParts = int(rng.integers(0, 4)), int(rng.integers(0, 4)), K
        par.partGraph = (rng.random((K, K)) < .1).astype(np.uint8) if case % 2 else None
        want = gtd["get_part_features"](E, par)
        got = oracle.get_part_features(E, par.logParts, par.logInvParts, par.bernsteinEdgeThreshold, par.spreadRadiusX,
                                       par.spreadRadiusY, K, par.partGraph)
        assert np.array_equal(want, got), case
        g0, g1 = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        assert np.array_equal(gtd["collapse_to_grid"](want, g0, g1), oracle.collapse_to_grid(got, g0, g1))


def test_filter_construction_and_window_score_restatements_are_the_references_code(oracle):
    """estimate_template.construct_linear_filter(s) (:156-172, :230-251) and
    test_template.score_template_background_section (:25-41)."""
    et = py2ref.load("estimate_template", ["construct_linear_filters", "construct_linear_filter"])
    tt = py2ref.load("test_template", ["score_template_background_section"])
    rng = np.random.default_rng(1806)
    for case in range(30):
        L, F, K = int(rng.integers(2, 30)), int(rng.integers(2, 12)), int(rng.integers(1, 9))
        Ts = tuple(rng.uniform(0, 1, size=(L + i, F, K)) for i in range(3))
        bgd = rng.uniform(0, .6, size=(F, K))
        for all_cs in (False, True):
            want = et["construct_linear_filters"](Ts, bgd, all_cs=all_cs)
            got = oracle.construct_linear_filters(Ts, bgd, all_cs=all_cs)
            for (w0, c0), (w1, c1) in zip(want, got):
                assert w0.dtype == np.float32 and np.array_equal(w0, w1) and np.array_equal(c0, c1), case
        D = F * K
        tmpl, b, Ew = rng.uniform(.02, .98, size=(D, L)), rng.uniform(.05, .4, size=D), (rng.random((D, L)) < .3).astype(np.uint8)
        assert tt["score_template_background_section"](tmpl, b, Ew) == oracle.score_template_background_section(tmpl, b, Ew)


def test_waliji_feature_map_restatement_is_the_references_code(oracle, ref_modules):
    """get_train_data.get_waliji_feature_map (:136-200) behind its spectrogram: edges -> code_parts_fast(thr 10) ->
    argmax -> spread_waliji_patches -> collapse_to_grid, every native piece the compiled reference's."""
    esp_f = py2ref.load("edge_signal_proc", ["_edge_map_no_threshold", "_edge_map_threshold_segments"])

    class Esp(object):                      # the module object the function calls into; S is handed in
        def __init__(self, S):
            self.S = S

        def get_spectrogram_features(self, *a, **k):
            return self.S

        def _edge_map_no_threshold(self, S):
            E, breaks, orient = esp_f["_edge_map_no_threshold"](S)
            return E, breaks.astype(int), orient.astype(int)      # (integral floats were valid slice bounds then)

        def _edge_map_threshold_segments(self, *a, **k):
            return esp_f["_edge_map_threshold_segments"](*a, **k)

    rng = np.random.default_rng(1807)
    for case in range(12):
        H, T = int(rng.integers(12, 40)), int(rng.integers(45, 130))
        S = rng.standard_normal((H, T))
        K, p0, p1 = int(rng.integers(2, 12)), int(rng.integers(2, 6)), int(rng.integers(2, 6))
        P = rng.uniform(.05, .95, size=(K, p0, p1, 8))
        gtd = py2ref.load("get_train_data", ["get_waliji_feature_map"],
                          env={"esp": Esp(S), "cp": ref_modules["code_parts"], "swp": ref_modules["spread_waliji_patches"]})
        want = gtd["get_waliji_feature_map"](None, np.log(P), np.log(1 - P))
        got = oracle.waliji_feature_map(oracle.get_edge_features(S, 40, 1, .7), np.log(P), np.log(1 - P))
        assert np.array_equal(want, got), case


def test_dropin_host_logic_is_the_references_code():
    """The host-side NumPy helpers of the PRODUCT package (no GPU involved) against the reference's own functions."""
    sys.path.insert(0, os.path.join(ROOT, "template-speech-recognition_amd"))
    import template_speech_rec.classification as cl
    import template_speech_rec.estimate_template as et
    import template_speech_rec.get_train_data as gtd
    import template_speech_rec.roc_functions as rf
    r_gtd = py2ref.load("get_train_data", ["reorg_part_for_fast_filtering", "reorg_parts_for_fast_filtering",
                                           "collapse_to_grid", "get_classify_windows"])
    r_et = py2ref.load("estimate_template", ["construct_linear_filters", "construct_linear_filter"])
    r_cl = py2ref.load("classification", ["map_coarse_frame_to_range", "get_coarse_segment", "_stretch_template"])
    r_rf = py2ref.load("roc_functions", ["get_cluster_starts_ends"])
    import template_speech_rec.edge_signal_proc as esp
    r_esp = py2ref.load("edge_signal_proc", ["compute_edge_feature_row_breaks"])
    for base in (1, 7, 45):
        assert np.array_equal(esp.compute_edge_feature_row_breaks(base, esp.EDGE_ORIENTATIONS),
                              r_esp["compute_edge_feature_row_breaks"](base, esp.EDGE_ORIENTATIONS))
    rng = np.random.default_rng(1808)
    for case in range(25):
        H, T, K = int(rng.integers(2, 20)), int(rng.integers(3, 60)), int(rng.integers(1, 6))
        part = rng.random((8 * H, T))
        assert np.array_equal(gtd.reorg_part_for_fast_filtering(part), r_gtd["reorg_part_for_fast_filtering"](part))
        parts = rng.uniform(0, 1, size=(K, 8 * H, T))
        assert np.array_equal(gtd.reorg_parts_for_fast_filtering(parts), r_gtd["reorg_parts_for_fast_filtering"](parts))
        coded = rng.integers(0, 2, size=(T, H, K))
        g0, g1 = int(rng.integers(1, 5)), int(rng.integers(1, 5))
        assert np.array_equal(gtd.collapse_to_grid(coded, g0, g1), r_gtd["collapse_to_grid"](coded, g0, g1))
        flts = np.concatenate(([0], np.cumsum(rng.integers(3, 40, size=int(rng.integers(2, 9)))))).astype(np.int64)
        for a, b in zip(gtd.get_classify_windows(None, flts), r_gtd["get_classify_windows"](None, flts)):
            assert a.dtype == b.dtype and np.array_equal(a, b)
        L, F, E = int(rng.integers(2, 20)), int(rng.integers(2, 9)), int(rng.integers(1, 9))
        Ts = tuple(rng.uniform(0, 1, size=(L + i, F, E)) for i in range(2))
        bgd = rng.uniform(0, .6, size=(F, E))
        for all_cs in (False, True):
            for (w0, c0), (w1, c1) in zip(et.construct_linear_filters(Ts, bgd, all_cs=all_cs),
                                          r_et["construct_linear_filters"](Ts, bgd, all_cs=all_cs)):
                assert w0.dtype == w1.dtype == np.float32 and np.array_equal(w0, w1) and np.array_equal(c0, c1)
        seg = rng.random((F, T))
        for ctype in ("max", "avg"):
            for cf in (2, 3):
                assert np.array_equal(cl.get_coarse_segment(seg, coarse_type=ctype, coarse_factor=cf),
                                      r_cl["get_coarse_segment"](seg, coarse_type=ctype, coarse_factor=cf))
        for tl in (max(2, T // 2), T, T + 7):
            assert np.array_equal(cl._stretch_template(seg, F, tl), r_cl["_stretch_template"](seg, F, tl))
        n = int(rng.integers(1, 12))
        times = np.sort(rng.choice(np.arange(1, 300), size=n, replace=False)).astype(np.uint16)
        part = np.zeros(n + 1, np.uint8)
        part[0] = part[n] = 2
        part[1:n][rng.random(n - 1) < .4] = 2
        assert rf.get_cluster_starts_ends(times, part) == r_rf["get_cluster_starts_ends"](times, part)


def test_dropin_example_bookkeeping_is_the_references_code():
    """get_examples_from_phns_ftls / phns_syllable_matches / map_array_to_coarse_coordinates of the product package."""
    sys.path.insert(0, os.path.join(ROOT, "template-speech-recognition_amd"))
    import template_speech_rec.get_train_data as gtd
    r = py2ref.load("get_train_data", ["get_examples_from_phns_ftls"])
    rng = np.random.default_rng(1809)
    for case in range(40):
        n = int(rng.integers(3, 20))
        phns = rng.choice(np.array(["aa", "b", "k", "s"]), size=n)
        flts = np.concatenate(([0], np.cumsum(rng.integers(2, 30, size=n)))).astype(np.int64)
        syl = tuple(rng.choice(np.array(["aa", "b", "k", "s"]), size=int(rng.integers(1, 3))))
        lpb = None if case % 2 else np.zeros((3, int(rng.integers(2, 6)), 4, 8))
        want = r["get_examples_from_phns_ftls"](syl, phns, flts, lpb, int(flts[-1]) // 2)
        got = gtd.get_examples_from_phns_ftls(syl, phns, flts, lpb, int(flts[-1]) // 2)
        for a, b in zip(want, got):
            assert (a is None and b is None) or np.array_equal(a, b), case


def test_edge_threshold_spread_and_magnitude_restatements_are_the_references_code(oracle):
    """threshold_edgemap / spread_edgemap / magnitude_features / magnitude_features_whole_block and the branches
    of get_edge_features / get_part_features (mag + aux): oracle/oracle.py against the reference's own code
    (SciPy supplying maximum_filter / correlate / convolve as in the reference's imports)."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden_edges2 as mg
    esp, gtd = mg.reference_functions()
    rng = np.random.default_rng(77)
    for trial in range(40):
        H, T = int(rng.integers(3, 24)), int(rng.integers(5, 100))
        E = rng.standard_normal((8 * H, T)) * rng.choice([0.02, 1.0])
        if trial % 3 == 0:
            E = np.round(E * 4) / 4
        if trial % 7 == 0:
            E[2 * H:3 * H] = -np.abs(E[2 * H:3 * H])
        q, binary, breaks = float(rng.choice([0.0, .3, .7, .95])), bool(trial % 2), np.arange(9) * H
        A, B = E.copy(), E.copy()
        allpos = (A > 0).reshape(8, -1).any(1).all()
        r = esp["threshold_edgemap"](A, q, breaks, report_level=bool(allpos), abst_threshold=oracle.ABST_THRESHOLD,
                                     convert_to_binary=binary)
        taus = oracle.threshold_edgemap(B, q, breaks, oracle.ABST_THRESHOLD, binary)
        assert np.array_equal(A, B) and (r is None or np.array_equal(r[1], taus))
        sl = int(rng.integers(1, 4))
        esp["spread_edgemap"](A, breaks, oracle.EDGE_ORIENTATIONS, spread_length=sl)
        oracle.spread_edgemap(B, breaks, oracle.EDGE_ORIENTATIONS, sl)
        assert np.array_equal(A, B)
    done = 0
    for trial in range(80):
        F, T = int(rng.integers(3, 40)), int(rng.integers(30, 200))
        S = rng.standard_normal((F, T))
        if trial % 4 == 0:
            S = np.round(S * 3) / 3
        bl, q = int(rng.choice([8, 20, 40, 41])), float(rng.choice([.5, .7, .9]))
        sm, dn, msl, sr = int(rng.integers(0, 5)), int(rng.integers(0, 3)), int(rng.integers(0, 3)), int(rng.integers(0, 8))
        try:
            a = esp["magnitude_features"](S.copy(), bl, sr, q, sm, dn, mag_spread_length=msl)
            a2 = esp["magnitude_features_whole_block"](S.copy(), bl, sr, q, sm, dn)
        except IndexError:           # a rank past a short window: the reference fails there
            continue
        done += 1
        assert np.array_equal(a, oracle.magnitude_features(S.copy(), bl, sr, q, sm, dn, msl))
        assert np.array_equal(a2, oracle.magnitude_features_whole_block(S.copy(), bl, sr, q, sm, dn))
    assert done > 40
    for trial in range(6):
        H, T = int(rng.integers(6, 24)), int(rng.integers(50, 130))
        S = rng.standard_normal((H + 3, T))
        for kw in (dict(magnitude_and_edge_features=True, magnitude_spread=int(rng.integers(0, 3))),
                   dict(magnitude_and_edge_features=True, auxiliary_data=True, magnitude_block_length=20),
                   dict(magnitude_thresholded_edges=True, magnitude_threshold=.4),
                   dict(magnitude_thresholded_edges=True, auxiliary_data=True, spread_length=2, block_length=25), dict()):
            p = mg.edge_params(**kw)
            Sin = S if p.auxiliary_data else S[:H]
            assert np.array_equal(np.asarray(gtd["get_edge_features"](Sin.copy(), p)),
                                  oracle.get_edge_features_with_parameters(Sin.copy(), p)), kw
    flips = 0
    for trial in range(5):
        T, F, K = int(rng.integers(12, 50)), int(rng.integers(6, 16)), int(rng.integers(2, 7))
        E, p = mg.mag_aux_case(rng, T, F, K, int(rng.integers(1, 5)), int(rng.integers(1, 4)))
        want = np.asarray(gtd["get_part_features"](E.copy(), p))
        got = oracle.get_part_features_mag_aux(E.copy(), p)
        assert got.shape == want.shape
        flips += int((got != want).sum())       # (only a Gaussian count within float32 rounding of 12 could differ)
    assert flips == 0
