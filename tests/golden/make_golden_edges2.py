#!/usr/bin/env python3
"""Golden vectors g23 from the reference's own PYTHON code (through oracle/py2ref.py, SciPy supplying the
filters it imports): edge_signal_proc.threshold_edgemap / spread_edgemap / magnitude_features /
magnitude_features_whole_block, and get_train_data.get_edge_features (magnitude branches) / get_part_features
(mag + auxiliary branch, over code_parts.pyx compiled with Python-2 semantics).  Build container only:
    python tests/golden/make_golden_edges2.py
"""
import collections
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import build_ref  # noqa: E402
import oracle  # noqa: E402
import py2ref  # noqa: E402

PART_TABLES = ("logParts", "logInvParts", "logMagFeatures", "logInvMagFeatures", "logAuxiliaryFeatures",
               "logInvAuxiliaryFeatures")


def reference_functions():
    from scipy.ndimage import correlate, maximum_filter
    from scipy.signal import convolve
    assert build_ref.build(), "oracle/_ref could not be built (no /root/reference?)"
    esp_f = py2ref.load("edge_signal_proc", ["threshold_edgemap", "spread_edgemap", "magnitude_features",
                                             "magnitude_features_whole_block", "_edge_map_no_threshold",
                                             "_edge_map_threshold_segments"],
                        env={"maximum_filter": maximum_filter, "correlate": correlate})
    raw = esp_f["_edge_map_no_threshold"]

    def no_threshold(S):   # the NumPy of the reference's time took integral floats as slice bounds; today's wants ints
        E, breaks, orientations = raw(S)
        return E, breaks.astype(int), orientations.astype(int)

    esp = types.SimpleNamespace(**dict(esp_f, _edge_map_no_threshold=no_threshold))
    cp2 = build_ref.load_extra("code_parts", py2=True)
    gtd = py2ref.load("get_train_data", ["get_edge_features", "get_part_features"],
                      env={"esp": esp, "cp": cp2, "convolve": convolve, "gaussian5by5": oracle.GAUSSIAN5BY5})
    return esp_f, gtd


EdgeP = collections.namedtuple("EdgeP", "block_length spread_length threshold abst_threshold magnitude_block_length "
                                        "magnitude_spread magnitude_threshold mag_smooth_freq mag_downsample_freq "
                                        "auxiliary_data auxiliary_threshold magnitude_features magnitude_and_edge_features "
                                        "magnitude_thresholded_edges")
PartP = collections.namedtuple("PartP", "mag_data auxiliary_data num_mag_channels num_auxiliary_data logParts logInvParts "
                                        "logMagFeatures logInvMagFeatures logAuxiliaryFeatures logInvAuxiliaryFeatures "
                                        "spreadRadiusX spreadRadiusY numParts partGraph bernsteinEdgeThreshold")


def edge_params(**kw):
    base = dict(block_length=40, spread_length=1, threshold=.7, abst_threshold=oracle.ABST_THRESHOLD,
                magnitude_block_length=40, magnitude_spread=1, magnitude_threshold=.7, mag_smooth_freq=0,
                mag_downsample_freq=0, auxiliary_data=False, auxiliary_threshold=.5, magnitude_features=False,
                magnitude_and_edge_features=False, magnitude_thresholded_edges=False)
    base.update(kw)
    return EdgeP(**base)


EDGE_CASES = (("both", edge_params(magnitude_and_edge_features=True, magnitude_spread=2)),
              ("both_aux", edge_params(magnitude_and_edge_features=True, auxiliary_data=True, magnitude_block_length=20)),
              ("masked", edge_params(magnitude_thresholded_edges=True, magnitude_threshold=.4, magnitude_spread=1)),
              ("masked_aux", edge_params(magnitude_thresholded_edges=True, auxiliary_data=True, magnitude_threshold=.5,
                                         spread_length=2, block_length=25)))


def mag_aux_case(rng, T, F, K, nm, na):
    """Random inputs of the mag + aux part coder: E [T, 8 F + nm + na] 0/1, tables of a 5x5x8 coder."""
    E = (rng.random((T, 8 * F + nm + na)) < 0.35).astype(np.uint8)
    lp = np.log(rng.uniform(.05, .95, (K, 5, 5, 8)))
    lmp = np.log(rng.uniform(.05, .95, (K, 5 + nm)))
    lap = np.log(rng.uniform(.05, .95, (K, 5 * na)))
    lip, lmip, laip = (np.log(1 - np.exp(a)) for a in (lp, lmp, lap))
    return E, PartP(True, True, nm, na, lp, lip, lmp, lmip, lap, laip, 2, 1, K, None, 12)


if __name__ == "__main__":
    esp, gtd = reference_functions()
    rng = np.random.default_rng(2300)
    d = {}
    # threshold_edgemap / spread_edgemap
    cases = [(16, 80, .3, True, 3), (12, 57, .7, False, 1), (6, 31, .0, True, 2), (10, 64, .9, True, 3)]
    for i, (H, T, q, binary, sl) in enumerate(cases):
        E = (rng.standard_normal((8 * H, T)) * (0.03 if i == 1 else 1.0)).astype(np.float32).astype(np.float64)
        if i == 3:
            E = np.round(E * 4) / 4
            E[2 * H:3 * H] = -np.abs(E[2 * H:3 * H])          # a block without positive values
        breaks = np.arange(9) * H
        A = E.copy()
        esp["threshold_edgemap"](A, q, breaks, report_level=False, abst_threshold=oracle.ABST_THRESHOLD, convert_to_binary=binary)
        B = A.copy()
        esp["spread_edgemap"](B, breaks, oracle.EDGE_ORIENTATIONS, spread_length=sl)
        d["te%d_E" % i], d["te%d_p" % i] = E.astype(np.float32), np.array([H, q, binary, sl])   # (inputs are float32-exact)
        d["te%d_thr" % i], d["te%d_spread" % i] = A.astype(np.float32), B.astype(np.float32)
    d["te_n"] = np.int64(len(cases))
    # magnitude features
    cases = [(45, 120, 40, 1, .7, 0, 0, 1), (30, 93, 20, 3, .9, 3, 2, 2), (12, 64, 8, 2, .5, 4, 0, 0), (20, 130, 41, 6, .7, 0, 3, 1)]
    for i, (F, T, bl, sr, q, sm, dn, msl) in enumerate(cases):
        S = rng.standard_normal((F, T)).astype(np.float32).astype(np.float64)
        if i == 1:
            S = np.round(S * 3) / 3
        d["mg%d_S" % i], d["mg%d_p" % i] = S.astype(np.float32), np.array([bl, sr, q, sm, dn, msl])
        d["mg%d_bin" % i] = esp["magnitude_features"](S.copy(), bl, sr, q, sm, dn, mag_spread_length=msl)
        d["mg%d_whole" % i] = esp["magnitude_features_whole_block"](S.copy(), bl, sr, q, sm, dn)
    d["mg_n"] = np.int64(len(cases))
    # get_edge_features branches
    H, T = 20, 97
    S = rng.standard_normal((H + 3, T)).astype(np.float32).astype(np.float64)
    for name, p in EDGE_CASES:
        Sin = S if p.auxiliary_data else S[:H]
        d["ef_" + name] = np.asarray(gtd["get_edge_features"](Sin.copy(), p)).astype(np.uint8)   # zeros and ones
    d["ef_S"] = S.astype(np.float32)
    # get_part_features, mag + aux branch
    for i, (T, F, K, nm, na) in enumerate([(40, 14, 6, 4, 3), (23, 9, 3, 2, 1)]):
        E, p = mag_aux_case(rng, T, F, K, nm, na)
        d["pf%d_E" % i] = E
        for f in PART_TABLES:
            d["pf%d_%s" % (i, f)] = getattr(p, f)
        d["pf%d_p" % i] = np.array([nm, na, K])
        d["pf%d_out" % i] = np.asarray(gtd["get_part_features"](E.copy(), p)).astype(np.uint8)
    d["pf_n"] = np.int64(2)
    path = os.path.join(HERE, "g23_edges2_reftext.npz")
    np.savez_compressed(path, **d)
    print("g23_edges2_reftext %.1f KB" % (os.path.getsize(path) / 1024))
