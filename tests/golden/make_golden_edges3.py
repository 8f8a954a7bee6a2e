#!/usr/bin/env python3
"""Golden vectors g24 from the reference's own PYTHON code (through oracle/py2ref.py): the two parameter branches of
edge_signal_proc.py that round 2 left out -- spread_type='all' of _edge_map_threshold_segments (:1623-1636, :1664-1675)
and critical_bands of threshold_edgemap (:1208-1250, which the reference can only execute up to a NumPy broadcast
error except in one degenerate geometry: the fixture holds the exception text and the array as it was left) -- plus
return_thresholds (the early return from inside the block loop).  Build container only:
    python tests/golden/make_golden_edges3.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import build_ref  # noqa: E402
import py2ref  # noqa: E402


def main():
    from scipy.ndimage import correlate, maximum_filter
    assert build_ref.build()
    f = py2ref.load("edge_signal_proc", ["threshold_edgemap", "_edge_map_threshold_segments", "spread_edgemap",
                                         "_edge_map_no_threshold"],
                    env={"maximum_filter": maximum_filter, "correlate": correlate})
    rng = np.random.default_rng(24)
    out = {}
    # ---- (a) spread_type='all' ------------------------------------------------------------------------------------
    cases = [dict(H=6, T=25, BL=10, s=1, binary=True), dict(H=9, T=40, BL=40, s=2, binary=True),
             dict(H=5, T=17, BL=20, s=1, binary=True), dict(H=7, T=33, BL=8, s=1, binary=False),
             dict(H=6, T=30, BL=10, s=0, binary=True), dict(H=6, T=21, BL=10, s=1, binary=True, kind="other")]
    out["a_n"] = len(cases)
    for i, c in enumerate(cases):
        S = (rng.normal(size=(c["T"], c["H"])).cumsum(0) + 0.3 * rng.normal(size=(c["T"], c["H"]))).T.copy()   # [frequency, time]
        E, brk, ori = f["_edge_map_no_threshold"](S)
        brk, ori = brk.astype(int), ori.astype(int)
        E_in = E.copy()
        r = f["_edge_map_threshold_segments"](E, c["BL"], c["s"], threshold=.6, edge_orientations=ori,
                                              edge_feature_row_breaks=brk, spread_type=c.get("kind", "all"),
                                              convert_to_binary=c["binary"])
        assert r is None
        out["a%d_E_in" % i], out["a%d_E_out" % i], out["a%d_breaks" % i], out["a%d_orient" % i] = E_in, E, brk, ori
        out["a%d_par" % i] = np.array([c["BL"], c["s"], int(c["binary"]), 0 if c.get("kind", "all") == "all" else 1])
    # ---- (b) critical_bands -----------------------------------------------------------------------------------------
    H = 6
    bands = [dict(breaks=np.arange(9) * H, cb=[[0, 6]], report=False), dict(breaks=np.arange(9) * H, cb=[[0, 3], [3, 6]], report=False),
             dict(breaks=np.arange(9) * H, cb=[[0, 2], [2, 6]], report=True), dict(breaks=np.array([0, H]), cb=[[0, 6]], report=False),
             dict(breaks=np.array([0, H]), cb=[[0, 6]], report=True), dict(breaks=np.array([0, H]), cb=[[0, 2], [2, 6]], report=False),
             dict(breaks=np.arange(9) * H, cb=[[0, 5]], report=False)]
    out["b_n"] = len(bands)
    for i, c in enumerate(bands):
        rows = int(c["breaks"][-1])
        E = rng.normal(size=(rows, 20))
        E_in = E.copy()
        err, ret = "", None
        try:
            ret = f["threshold_edgemap"](E, .7, c["breaks"], report_level=c["report"], critical_bands=np.array(c["cb"]))
        except Exception as e:
            err = "%s: %s" % (type(e).__name__, e)
        out["b%d_E_in" % i], out["b%d_E_out" % i], out["b%d_breaks" % i], out["b%d_cb" % i] = E_in, E, c["breaks"], np.array(c["cb"])
        out["b%d_report" % i], out["b%d_err" % i] = int(c["report"]), np.array(err)
        if ret is not None:
            out["b%d_ret_E" % i], out["b%d_ret_tau" % i] = ret[0], ret[1]
        print("critical_bands case", i, c["cb"], "->", err or "ok")
    # ---- (c) return_thresholds: the list comes back after the FIRST full block ------------------------------------
    S = rng.normal(size=(34, 6)).cumsum(0).T.copy()
    E, brk, ori = f["_edge_map_no_threshold"](S)
    brk, ori = brk.astype(int), ori.astype(int)
    E_in = E.copy()
    views = f["_edge_map_threshold_segments"](E, 10, 1, edge_orientations=ori, edge_feature_row_breaks=brk, return_thresholds=True)
    out["c_E_in"], out["c_E_out"], out["c_breaks"], out["c_orient"], out["c_views"] = E_in, E, brk, ori, np.array(views)
    # ... and without a full block the function falls off its end: None
    S2 = rng.normal(size=(7, 6)).cumsum(0).T.copy()
    E2, brk2, ori2 = f["_edge_map_no_threshold"](S2)
    E2_in = E2.copy()
    none = f["_edge_map_threshold_segments"](E2, 10, 1, edge_orientations=ori2.astype(int), edge_feature_row_breaks=brk2.astype(int),
                                             return_thresholds=True)
    assert none is None
    out["c2_E_in"], out["c2_E_out"], out["c2_breaks"], out["c2_orient"] = E2_in, E2, brk2.astype(int), ori2.astype(int)
    np.savez_compressed(os.path.join(HERE, "g24_edges3_reftext.npz"), **out)
    print("g24 written:", len(out), "arrays; return_thresholds gave", np.array(views).shape)


if __name__ == "__main__":
    main()
