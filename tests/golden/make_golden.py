#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE's compiled kernels.

Run in the build container only (needs oracle/_ref, built by
oracle/build_ref.py from /root/reference's own .pyx).  Each fixture holds the
seeded inputs and the reference's outputs -- data only, no reference text.
Binary feature maps are stored bit-packed (np.packbits) with their shape.

    python tests/golden/make_golden.py

Cases follow SURVEY.md section 8(c) G1-G10.  Where the reference's caller is
Python-2 glue that cannot be imported (the bank epilogue of
get_train_data.py:962-992, get_part_features :2619-2628, the waliji map
:183-191), the fixture records the outputs of the reference's compiled kernels
chained by that glue's few NumPy lines, restated here with the NumPy-1.x f32
promotion made explicit.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "template-speech-recognition_amd"))

import build_ref  # noqa: E402
from tsr_hip import synth  # noqa: E402

assert build_ref.build(), "oracle/_ref could not be built (no /root/reference?)"
cllf, cp, swp = build_ref.load()


def pack(a):
    a = np.ascontiguousarray(a, np.uint8)
    assert a.max(initial=0) <= 1
    return np.packbits(a.reshape(-1)), np.array(a.shape, np.int64)


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("%-28s %8.1f KB" % (name, os.path.getsize(path) / 1024))


def bank_epilogue(E, bank):
    """get_train_data.py:962-992 on top of the reference's detect()."""
    E = E.astype(np.uint8)
    row = cllf.detect(E, bank[0][0]) + np.float32(bank[0][1])
    ids = np.zeros(E.shape[0], np.int32)
    for m, (W, c) in enumerate(bank[1:], start=1):
        v = cllf.detect(E, W) + np.float32(c)
        ids[v > row] = m
        row = np.maximum(v, row)
    return row.astype(np.float32), ids


# ---- G1: detect (u8) ---------------------------------------------------------
d = {}
cases = [(50, 6, 5, 9, None), (400, 45, 8, 32, None), (120, 41, 50, 20, None), (32, 45, 8, 32, None),
         (64, 6, 5, 9, 0.05), (64, 6, 5, 9, 0.15), (64, 6, 5, 9, 0.40), (40, 6, 5, 9, 0.0),
         (40, 6, 5, 9, 1.0)]
for i, (T, F, K, L, dens) in enumerate(cases):
    rng = np.random.default_rng(100 + i)
    q = synth.background(rng, (F, K)) if dens is None else np.full((F, K), dens)
    X = synth.features(rng, T, q)
    W, c = synth.bank(rng, synth.background(rng, (F, K)), [L])[0]
    d["c%d_X_bits" % i], d["c%d_X_shape" % i] = pack(X)
    d["c%d_W" % i] = W
    d["c%d_out" % i] = cllf.detect(X, W)
d["n"] = np.int64(len(cases))
save("g01_detect_u8", **d)

# ---- G2: detect_float / detect_float2d / detect_spectral ----------------------
d = {}
rng = np.random.default_rng(200)
for i, (T, F, K, L) in enumerate([(60, 6, 5, 9), (90, 45, 8, 32)]):
    q = synth.background(rng, (F, K))
    Xb = synth.features(rng, T, q).astype(np.float32)
    # background-padded window as get_isolated_classify_windows builds (:781-833)
    Xf = Xb.copy()
    Xf[: T // 5] = q.astype(np.float32)
    Xf[-(T // 4):] = q.astype(np.float32)
    Xr = rng.standard_normal((T, F, K)).astype(np.float32)
    W = synth.bank(rng, q, [L])[0][0]
    for tag, X in (("bin", Xb), ("pad", Xf), ("real", Xr)):
        key = "c%d_%s" % (i, tag)
        d[key + "_X"] = X
        d[key + "_out3d"] = cllf.detect_float(X, W)
        d[key + "_out2d"] = cllf.detect_float2d(X.reshape(T, -1), W.reshape(L, -1))
        d[key + "_outsp"] = cllf.detect_spectral(X.reshape(T, -1), W.reshape(L, -1))
    d["c%d_W" % i] = W
d["n"] = np.int64(2)
save("g02_detect_float", **d)

# ---- G3: detect_float_max ------------------------------------------------------
d = {}
rng = np.random.default_rng(300)
for i, (T, F, K, L, M) in enumerate([(70, 6, 5, 9, 3), (80, 12, 8, 16, 8)]):
    q = synth.background(rng, (F, K))
    X = synth.features(rng, T, q).astype(np.float32)
    bk = synth.bank(rng, q, [L] * M)
    LF = np.stack([w for w, _ in bk]).astype(np.float32)
    LF[M - 1] = LF[0]  # exact tie between template 0 and the last one
    cs = np.array([c for _, c in bk], np.float32)
    cs[M - 1] = cs[0]
    s, a = cllf.detect_float_max(X, LF, cs)
    d["c%d_X" % i], d["c%d_LF" % i], d["c%d_cs" % i] = X, LF, cs
    d["c%d_out" % i], d["c%d_ids" % i] = s, a
d["n"] = np.int64(2)
save("g03_detect_float_max", **d)

# ---- G4: bank epilogue, variable L_m, forced ties ------------------------------
d = {}
rng = np.random.default_rng(400)
for i, (T, F, K, Ls) in enumerate([(90, 6, 5, [20, 9, 33, 48, 20]), (130, 45, 8, [24, 32, 40, 48, 20, 28])]):
    q = synth.background(rng, (F, K))
    X = synth.features(rng, T, q)
    bk = list(synth.bank(rng, q, Ls))
    bk[-1] = (bk[0][0].copy(), bk[0][1])  # duplicate template 0 as the last one: tie -> id 0
    s, a = bank_epilogue(X, bk)
    d["c%d_X_bits" % i], d["c%d_X_shape" % i] = pack(X)
    d["c%d_M" % i] = np.int64(len(bk))
    for m, (W, c) in enumerate(bk):
        d["c%d_W%d" % (i, m)] = W
        d["c%d_c%d" % (i, m)] = np.float64(c)
    d["c%d_scores" % i], d["c%d_ids" % i] = s, a
d["n"] = np.int64(2)
save("g04_bank_epilogue", **d)

# ---- G5: get_max_detect_examples ------------------------------------------------
rng = np.random.default_rng(500)
N, T, F, K, L = 7, 40, 6, 5, 9
q = synth.background(rng, (F, K))
X = np.stack([synth.features(rng, T, q) for _ in range(N)])
W64 = synth.bank(rng, q, [L])[0][0].astype(np.float64)
lengths = np.array([40, 1, 17, 32, 36, 5, 40], np.int64)
xb, xs = pack(X)
save("g05_max_detect_examples", X_bits=xb, X_shape=xs, LF=W64, lengths=lengths,
     out=cllf.get_max_detect_examples(X, W64, lengths))

# ---- G6: code_parts / code_parts_fast --------------------------------------------
d = {}
rng = np.random.default_rng(600)
cfgs = [(30, 12, 6, [0, 10, 12, 10 ** 6, -1]), (24, 45, 50, [12, 40])]
for i, (T, F, K, thrs) in enumerate(cfgs):
    q = rng.uniform(0.02, 0.45, size=(F, 8))
    X = synth.features(rng, T, q)
    X[: T // 3, : F // 2] = 0  # a sparse corner so that both branches occur
    lp, lip = synth.parts(rng, K)
    lp[K - 1], lip[K - 1] = lp[1], lip[1]  # duplicate part: exact tie -> first max
    d["c%d_X_bits" % i], d["c%d_X_shape" % i] = pack(X)
    d["c%d_lp" % i], d["c%d_lip" % i] = lp, lip
    d["c%d_thr" % i] = np.array(thrs, np.int64)
    for j, thr in enumerate(thrs):
        o = cp.code_parts(X, lp, lip, thr)
        of = cp.code_parts_fast(X, lp, lip, thr)
        if K <= 6:
            d["c%d_t%d_out" % (i, j)] = o
            d["c%d_t%d_outfast" % (i, j)] = of
        else:  # keep the big case small: a slab of the map + both argmax maps
            d["c%d_t%d_out" % (i, j)] = o[:6]
            d["c%d_t%d_outfast" % (i, j)] = of[:6]
        d["c%d_t%d_argmax" % (i, j)] = np.argmax(o, -1)
        d["c%d_t%d_argmaxfast" % (i, j)] = np.argmax(of, -1)
d["n"] = np.int64(len(cfgs))
save("g06_code_parts", **d)

# ---- G7: spread_patches == spread_waliji_patches ---------------------------------
d = {}
rng = np.random.default_rng(700)
for i, (X0, X1, K, zero_frac) in enumerate([(26, 8, 6, 0.5), (40, 41, 50, 0.7), (9, 5, 3, 0.0)]):
    A = rng.integers(1, K + 1, size=(X0, X1)).astype(np.int64)
    A[rng.random((X0, X1)) < zero_frac] = 0
    d["c%d_A" % i] = A
    d["c%d_K" % i] = np.int64(K)
    for j, (r0, r1) in enumerate([(0, 0), (2, 2), (5, 5), (1, 3)]):
        o = cp.spread_patches(A, r0, r1, K)
        assert np.array_equal(o, swp.spread_waliji_patches(A, r0, r1, K))
        d["c%d_r%d_bits" % (i, j)], d["c%d_r%d_shape" % (i, j)] = pack(o)
d["radii"] = np.array([(0, 0), (2, 2), (5, 5), (1, 3)], np.int64)
d["n"] = np.int64(3)
save("g07_spread", **d)

# ---- G8: spread_graph_patches (order dependence: chain graph) ---------------------
rng = np.random.default_rng(800)
X0, X1, K = 12, 7, 9
S = (rng.random((X0, X1, K)) < 0.12).astype(np.uint8)
chain = np.zeros((K, K), np.uint8)
for a in range(K - 1):
    chain[a, a + 1] = 1  # a -> a+1: one forward pass closes the whole chain
back = np.zeros((K, K), np.uint8)
for a in range(1, K):
    back[a, a - 1] = 1  # a -> a-1: only one step propagates per pass
rand = (rng.random((K, K)) < 0.2).astype(np.uint8)
d = {"S_bits": pack(S)[0], "S_shape": pack(S)[1]}
for nm, g in (("chain", chain), ("back", back), ("rand", rand)):
    Y = S.copy()
    cp.spread_graph_patches(Y, g)
    d[nm + "_graph"] = g
    d[nm + "_bits"], d[nm + "_shape"] = pack(Y)
save("g08_spread_graph", **d)

# ---- G9: end to end: code_parts -> argmax -> spread -> detect -> +c -> max/ids -----
rng = np.random.default_rng(900)
T, F, K, M = 64, 20, 12, 3
q = rng.uniform(0.05, 0.45, size=(F, 8))
X = synth.features(rng, T, q)
lp, lip = synth.parts(rng, K)
graph = (rng.random((K, K)) < 0.08).astype(np.uint8)
out = cp.code_parts(X, lp, lip, 12)
mr = np.argmax(out, -1)
sp = cp.spread_patches(mr, 2, 2, K)
spg = sp.copy()
cp.spread_graph_patches(spg, graph)
qp = synth.background(rng, sp.shape[1:])
bk = synth.bank(rng, qp, [16, 9, 24])
s, a = bank_epilogue(sp, bk)
sg, ag = bank_epilogue(spg, bk)
d = {"X_bits": pack(X)[0], "X_shape": pack(X)[1], "lp": lp, "lip": lip, "graph": graph,
     "thr": np.int64(12), "r": np.array([2, 2], np.int64), "argmax": mr,
     "spread_bits": pack(sp)[0], "spread_shape": pack(sp)[1],
     "spreadg_bits": pack(spg)[0], "spreadg_shape": pack(spg)[1],
     "scores": s, "ids": a, "scores_graph": sg, "ids_graph": ag, "M": np.int64(M)}
for m, (W, c) in enumerate(bk):
    d["W%d" % m], d["c%d" % m] = W, np.float64(c)
save("g09_end_to_end", **d)

# ---- G10: waliji variant: code_parts_fast(10) -> argmax -> spread(5,5) -> [::5,::5] --
rng = np.random.default_rng(1000)
T, F, K = 70, 30, 10
q = rng.uniform(0.05, 0.45, size=(F, 8))
X = synth.features(rng, T, q)
lp, lip = synth.parts(rng, K)
Fm = cp.code_parts_fast(X, lp, lip, 10)
Am = np.argmax(Fm, 2)
Sm = swp.spread_waliji_patches(Am, lp.shape[1], lp.shape[2], lp.shape[0])
Gm = np.ascontiguousarray(Sm[::lp.shape[1], ::lp.shape[2]])
qg = synth.background(rng, Gm.shape[1:])
W, c = synth.bank(rng, qg, [4])[0]
save("g10_waliji", X_bits=pack(X)[0], X_shape=pack(X)[1], lp=lp, lip=lip, argmax=Am,
     grid_bits=pack(Gm)[0], grid_shape=pack(Gm)[1], W=W, c=np.float64(c),
     detect=cllf.detect(Gm.astype(np.uint8), W))

# ---- G11: Classifier surface identity: sliding sum(score_template_background_section)
#           == detect + c; reference detect on the reorganised map ----
rng = np.random.default_rng(1100)
H, L, T = 10, 7, 30
tmpl = rng.uniform(0.05, 0.95, size=(8 * H, L))
bg = rng.uniform(0.05, 0.40, size=8 * H)
Ef = (rng.random((8 * H, T)) < bg[:, None]).astype(np.uint8)
# reorg_part_for_fast_filtering (get_train_data.py:41-51): [8H, T] -> [T, H, 8]
E3 = np.array([Ef[i * H:(i + 1) * H].T for i in range(8)]).swapaxes(0, 1).swapaxes(1, 2)
T3 = np.array([tmpl[i * H:(i + 1) * H].T for i in range(8)]).swapaxes(0, 1).swapaxes(1, 2)
B3 = np.array([bg[i * H:(i + 1) * H] for i in range(8)]).T
W, c = synth.linear_filter(T3, B3)
save("g11_classifier_identity", template=tmpl, bg=bg, E_bits=pack(Ef)[0], E_shape=pack(Ef)[1],
     W=W, c=np.float64(c), detect=cllf.detect(np.ascontiguousarray(E3), W))
