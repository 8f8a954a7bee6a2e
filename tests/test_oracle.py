"""Pin the CPU oracle (oracle/tsr_oracle.c) to the reference.

Two anchors: (1) the committed golden vectors in tests/golden/ (outputs of the
reference's own compiled Cython kernels, see tests/golden/make_golden.py);
(2) where oracle/_ref is present, the compiled reference itself on fresh
random inputs.  Everything here is bit-exact (np.array_equal) because the
oracle restates the reference's loops in the same summation order.
"""
import numpy as np
import pytest

from helpers import golden, golden_bank, unpack
from tsr_hip import synth


def test_g01_detect_u8(oracle):
    g = golden("g01_detect_u8")
    for i in range(int(g["n"])):
        X = unpack(g["c%d_X_bits" % i], g["c%d_X_shape" % i])
        out = oracle.detect(X, g["c%d_W" % i])
        assert out.dtype == np.float32 and np.array_equal(out, g["c%d_out" % i]), i
        L = g["c%d_W" % i].shape[0]
        assert np.all(out[X.shape[0] - L + 1:] == 0)  # zero tail (quirk 1)


def test_g02_detect_float(oracle):
    g = golden("g02_detect_float")
    for i in range(int(g["n"])):
        W = g["c%d_W" % i]
        for tag in ("bin", "pad", "real"):
            key = "c%d_%s" % (i, tag)
            X = g[key + "_X"]
            out = oracle.detect_float(X, W)
            assert np.array_equal(out, g[key + "_out3d"])
            assert np.array_equal(out, g[key + "_out2d"])
            assert np.array_equal(out, g[key + "_outsp"])
        # detect_float on a binary map == detect on the same map
        Xb = g["c%d_bin_X" % i]
        assert np.array_equal(oracle.detect(Xb.astype(np.uint8), W), g["c%d_bin_out3d" % i])


def test_g03_detect_float_max(oracle):
    g = golden("g03_detect_float_max")
    for i in range(int(g["n"])):
        s, a = oracle.detect_float_max(g["c%d_X" % i], g["c%d_LF" % i], g["c%d_cs" % i])
        assert np.array_equal(s, g["c%d_out" % i])
        assert a.dtype == np.uint16 and np.array_equal(a, g["c%d_ids" % i])
        M = g["c%d_LF" % i].shape[0]
        assert not np.any(a == M - 1)  # duplicate of template 0 never wins (strict '>')


def test_g04_bank_epilogue(oracle):
    g = golden("g04_bank_epilogue")
    for i in range(int(g["n"])):
        X = unpack(g["c%d_X_bits" % i], g["c%d_X_shape" % i])
        bank = golden_bank(g, "c%d_" % i, g["c%d_M" % i])
        s, a = oracle.score_bank(X, bank)
        assert np.array_equal(s, g["c%d_scores" % i]) and np.array_equal(a, g["c%d_ids" % i])
        s2, a2 = oracle.compute_detection_E(X, bank)
        assert np.array_equal(s, s2) and np.array_equal(a, a2)
        allm = oracle.score_bank_all(X, bank)
        assert np.array_equal(allm.max(0), s) and np.array_equal(allm.argmax(0).astype(np.int32), a)


def test_g05_max_detect_examples(oracle):
    g = golden("g05_max_detect_examples")
    X = unpack(g["X_bits"], g["X_shape"])
    assert np.array_equal(oracle.get_max_detect_examples(X, g["LF"], g["lengths"]), g["out"])


def test_g06_code_parts(oracle):
    g = golden("g06_code_parts")
    for i in range(int(g["n"])):
        X = unpack(g["c%d_X_bits" % i], g["c%d_X_shape" % i])
        lp, lip = g["c%d_lp" % i], g["c%d_lip" % i]
        for j, thr in enumerate(g["c%d_thr" % i]):
            o = oracle.code_parts(X, lp, lip, int(thr))
            of = oracle.code_parts_fast(X, lp, lip, int(thr))
            n = g["c%d_t%d_out" % (i, j)].shape[0]
            assert np.array_equal(o[:n], g["c%d_t%d_out" % (i, j)])
            assert np.array_equal(of[:n], g["c%d_t%d_outfast" % (i, j)])
            am = oracle.argmax_last(o)
            assert np.array_equal(am, g["c%d_t%d_argmax" % (i, j)])
            assert np.array_equal(oracle.argmax_last(of), g["c%d_t%d_argmaxfast" % (i, j)])
            assert np.array_equal(am, np.argmax(o, -1))
            if int(thr) < 0 or int(thr) >= 10 ** 6:
                assert not am.any()  # quirk 4: negative threshold => all background


def test_g07_spread(oracle):
    g = golden("g07_spread")
    for i in range(int(g["n"])):
        A, K = g["c%d_A" % i], int(g["c%d_K" % i])
        for j, (r0, r1) in enumerate(g["radii"]):
            want = unpack(g["c%d_r%d_bits" % (i, j)], g["c%d_r%d_shape" % (i, j)])
            assert np.array_equal(oracle.spread_patches(A, int(r0), int(r1), K), want)


def test_g08_spread_graph(oracle):
    g = golden("g08_spread_graph")
    S = unpack(g["S_bits"], g["S_shape"])
    for nm in ("chain", "back", "rand"):
        Y = S.copy()
        assert oracle.spread_graph_patches(Y, g[nm + "_graph"]) is None
        assert np.array_equal(Y, unpack(g[nm + "_bits"], g[nm + "_shape"]))


def test_g09_end_to_end(oracle):
    g = golden("g09_end_to_end")
    X = unpack(g["X_bits"], g["X_shape"])
    K = g["lp"].shape[0]
    sp = oracle.get_part_features(X, g["lp"], g["lip"], int(g["thr"]), 2, 2, K)
    assert np.array_equal(sp, unpack(g["spread_bits"], g["spread_shape"]))
    spg = oracle.get_part_features(X, g["lp"], g["lip"], int(g["thr"]), 2, 2, K, g["graph"])
    assert np.array_equal(spg, unpack(g["spreadg_bits"], g["spreadg_shape"]))
    bank = golden_bank(g, "", g["M"])
    s, a = oracle.score_bank(sp, bank)
    assert np.array_equal(s, g["scores"]) and np.array_equal(a, g["ids"])
    s, a = oracle.score_bank(spg, bank)
    assert np.array_equal(s, g["scores_graph"]) and np.array_equal(a, g["ids_graph"])


def test_g10_waliji(oracle):
    g = golden("g10_waliji")
    X = unpack(g["X_bits"], g["X_shape"])
    G = oracle.waliji_feature_map(X, g["lp"], g["lip"])
    assert np.array_equal(G, unpack(g["grid_bits"], g["grid_shape"]))
    assert np.array_equal(oracle.detect(np.ascontiguousarray(G), g["W"]), g["detect"])


def test_g11_classifier_identity(oracle):
    """sum(score_template_background_section(window)) == detect + c at every t (A8)."""
    g = golden("g11_classifier_identity")
    E = unpack(g["E_bits"], g["E_shape"])
    tmpl, bg = g["template"], g["bg"]
    L = tmpl.shape[1]
    E3 = np.ascontiguousarray(oracle.reorg_part_for_fast_filtering(E))
    T3 = oracle.reorg_part_for_fast_filtering(tmpl)
    B3 = np.array([bg[i * (len(bg) // 8):(i + 1) * (len(bg) // 8)] for i in range(8)]).T
    W, c = oracle.construct_linear_filter(T3, B3)
    assert np.array_equal(W, g["W"]) and abs(c - float(g["c"])) <= 1e-9 * abs(c)
    det = oracle.detect(E3, W)
    assert np.array_equal(det, g["detect"])
    for t in range(E.shape[1] - L + 1):
        s = sum(oracle.score_template_background_section(tmpl, bg, E[:, t:t + L]))
        assert abs(s - (float(det[t]) + c)) <= 1e-5 * max(abs(s), 1.0)


# ---- against the compiled reference itself (present in the build container) --------
def test_oracle_vs_compiled_reference_random(oracle, ref):
    if ref is None:
        pytest.skip("oracle/_ref not built")
    cllf, cp, swp = ref
    rng = np.random.default_rng(7)
    for trial in range(6):
        T, F, K, L = int(rng.integers(20, 90)), int(rng.integers(2, 20)), int(rng.integers(1, 9)), int(rng.integers(1, 20))
        q = synth.background(rng, (F, K))
        X = synth.features(rng, T, q)
        W, c = synth.bank(rng, q, [L])[0]
        assert np.array_equal(oracle.detect(X, W), cllf.detect(X, W))
        Xf = (X * rng.random(X.shape)).astype(np.float32)
        assert np.array_equal(oracle.detect_float(Xf, W), cllf.detect_float(Xf, W))
        P0, P1 = int(rng.integers(1, 5)), int(rng.integers(1, 5))
        Kp = int(rng.integers(1, 7))
        X8 = synth.features(rng, T, rng.uniform(0.05, 0.5, size=(F + 4, 8)))
        lp, lip = synth.parts(rng, Kp, (P0, P1, 8))
        thr = int(rng.integers(0, 2 * P0 * P1))
        o_ref = cp.code_parts(X8, lp, lip, thr)
        assert np.array_equal(oracle.code_parts(X8, lp, lip, thr), o_ref)
        assert np.array_equal(oracle.code_parts_fast(X8, lp, lip, thr), cp.code_parts_fast(X8, lp, lip, thr))
        A = np.argmax(o_ref, -1)
        r0, r1 = int(rng.integers(0, 4)), int(rng.integers(0, 4))
        assert np.array_equal(oracle.spread_patches(A, r0, r1, Kp), swp.spread_waliji_patches(A, r0, r1, Kp))


def test_oracle_handles_T_less_than_L(oracle):
    """Documented deviation: the reference segfaults for T < L (uint16 wrap,
    compute_likelihood_linear_filter.pyx:157); the bank oracle treats every frame
    as tail (raw score 0), so the score is f32(c_m)."""
    rng = np.random.default_rng(3)
    q = synth.background(rng, (4, 3))
    X = synth.features(rng, 5, q)
    bank = synth.bank(rng, q, [3, 9])
    s, a = oracle.score_bank(X, bank)
    allm = oracle.score_bank_all(X, bank)
    assert np.all(allm[1] == np.float32(bank[1][1]))
    assert np.array_equal(s, allm.max(0))


@pytest.mark.parametrize("isolated", [False, True])
def test_classification_restatement_matches_golden(oracle, isolated):
    """g12 / g13 were produced by the reference's compiled detect / detect_float under the restated
    glue; the oracle's own C restatements must give the same arrays bit for bit."""
    g = golden("g13_isolated_classification_E" if isolated else "g12_classification_E")
    bank = [(g["W%d" % m], float(g["c%d" % m])) for m in range(int(g["M"]))]
    want = [g[k] for k in ("classify_array", "classify_template_ids", "classify_template_lengths", "classify_locs")]
    got = [np.zeros_like(w) for w in want]
    for u in range(int(g["N"])):
        shape = tuple(int(x) for x in g["Eshape%d" % u])
        E = np.unpackbits(g["E%d" % u])[:int(np.prod(shape))].reshape(shape)
        flts = g["flts%d" % u].copy()
        if isolated:
            oracle.compute_isolated_classification_E(E, None, flts, u, got[0], got[1], got[2], got[3], bank, g["q"])
            assert flts[0] > 0          # shifted in place, like the reference
        else:
            oracle.compute_classification_E(E, None, flts, u, got[0], got[1], got[2], got[3], bank)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)


def test_g14_cluster_times(oracle):
    """cluster_times.pyx is native code in the reference: g14 holds the compiled module's partitions."""
    g = golden("g14_cluster_times")
    for i in range(int(g["n"])):
        C0, C1 = (float(x) for x in g["C%d" % i])
        assert np.array_equal(oracle.cluster_times(g["t%d" % i], C0, C1), g["p%d" % i]), i


def test_cluster_times_restatement_vs_compiled_reference(oracle):
    """When oracle/_ref holds the compiled cluster_times.pyx: the C restatement on fresh random peak lists."""
    import build_ref
    build_ref.build()
    ct = build_ref.load_extra("cluster_times")
    if ct is None:
        pytest.skip("oracle/_ref/cluster_times not built here")
    rng = np.random.default_rng(77)
    for _ in range(500):
        n = int(rng.integers(0, 40))
        T = int(rng.integers(n + 2, 400))
        times = np.sort(rng.choice(np.arange(1, T), size=n, replace=False)).astype(np.uint16)
        C0, C1 = float(rng.integers(1, 40)), float(rng.integers(1, 80))
        assert np.array_equal(oracle.cluster_times(times, C0, C1), ct.cluster_times(times, C0, C1))


def test_g16_cluster_times_template_lengths(oracle):
    """cluster_times_template_lengths (cluster_times.pyx:221-310) as run by the compiled reference module."""
    g = golden("g16_cluster_times_template_lengths")
    for i in range(int(g["n"])):
        got = oracle.cluster_times_template_lengths(g["t%d" % i], g["l%d" % i], g["c%d" % i])
        assert np.array_equal(got, g["p%d" % i]), i


def test_cluster_times_template_lengths_restatement_vs_compiled_reference(oracle):
    import build_ref
    build_ref.build()
    ct = build_ref.load_extra("cluster_times")
    if ct is None:
        pytest.skip("oracle/_ref/cluster_times not built here")
    rng = np.random.default_rng(78)
    for _ in range(500):
        n = int(rng.integers(0, 40))
        T = int(rng.integers(n + 2, 400))
        times = np.sort(rng.choice(np.arange(1, T), size=n, replace=False)).astype(np.uint16)
        nl = int(rng.integers(1, 6))
        lengths = rng.integers(1, 50, size=nl).astype(np.uint16)
        comps = rng.integers(0, nl, size=n).astype(np.uint16)
        assert np.array_equal(oracle.cluster_times_template_lengths(times, lengths, comps),
                              ct.cluster_times_template_lengths(times, lengths, comps))


def test_g17_cluster_times_variable_length(oracle):
    g = golden("g17_cluster_times_variable_length")
    for i in range(int(g["n"])):
        assert np.array_equal(oracle.cluster_times_variable_length(g["t%d" % i], g["l%d" % i]), g["p%d" % i]), i


def _g15_case(g, i):
    return tuple(g["c%d_%s" % (i, n)] for n in ("Xe", "Xm", "Xa", "lp", "lip", "lmp", "lmip", "lap", "laip", "mask")), g["c%d_out" % i]


def test_g15_code_parts_mag_aux(oracle):
    """g15: the reference's code_parts_mag_aux compiled with Python-2 semantics (oracle/_ref/py2)."""
    g = golden("g15_code_parts_mag_aux")
    for i in range(int(g["n"])):
        args, want = _g15_case(g, i)
        assert np.array_equal(oracle.code_parts_mag_aux(*args), want), i


def test_g23_edge_threshold_spread_magnitude_from_the_references_own_code(oracle):
    """g23: the reference's threshold_edgemap / spread_edgemap / magnitude_features(_whole_block) and the
    magnitude / auxiliary branches of get_edge_features / get_part_features (run through oracle/py2ref.py)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden_edges2 as mg
    g = golden("g23_edges2_reftext")
    for i in range(int(g["te_n"])):
        H, q, binary, sl = g["te%d_p" % i]
        E = g["te%d_E" % i].astype(np.float64)
        breaks = np.arange(9) * int(H)
        oracle.threshold_edgemap(E, float(q), breaks, oracle.ABST_THRESHOLD, bool(binary))
        assert np.array_equal(E, g["te%d_thr" % i].astype(np.float64))
        oracle.spread_edgemap(E, breaks, oracle.EDGE_ORIENTATIONS, int(sl))
        assert np.array_equal(E, g["te%d_spread" % i].astype(np.float64))
    for i in range(int(g["mg_n"])):
        bl, sr, q, sm, dn, msl = g["mg%d_p" % i]
        S = g["mg%d_S" % i].astype(np.float64)
        assert np.array_equal(oracle.magnitude_features(S, int(bl), int(sr), float(q), int(sm), int(dn), int(msl)), g["mg%d_bin" % i])
        assert np.array_equal(oracle.magnitude_features_whole_block(S, int(bl), int(sr), float(q), int(sm), int(dn)), g["mg%d_whole" % i])
    S = g["ef_S"].astype(np.float64)
    H = S.shape[0] - 3
    for name, p in mg.EDGE_CASES:
        Sin = S if p.auxiliary_data else S[:H]
        assert np.array_equal(oracle.get_edge_features_with_parameters(Sin.copy(), p), g["ef_" + name]), name
    for i in range(int(g["pf_n"])):
        nm, na, K = (int(x) for x in g["pf%d_p" % i])
        p = mg.PartP(True, True, nm, na, *[g["pf%d_%s" % (i, f)] for f in mg.PART_TABLES], 2, 1, K, None, 12)
        assert np.array_equal(oracle.get_part_features_mag_aux(g["pf%d_E" % i], p), g["pf%d_out" % i])
