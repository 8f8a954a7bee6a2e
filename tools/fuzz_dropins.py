"""Random fuzz of the per-call drop-in functions (GPU box) against the oracle, exact equality."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "template-speech-recognition_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle
from tsr_hip import synth
import template_speech_rec.compute_likelihood_linear_filter as cllf
import template_speech_rec.code_parts as cp
import template_speech_rec.spread_waliji_patches as swp
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
def check(name, got, want, info):
    global bad
    ok = got.dtype == want.dtype and got.shape == want.shape and np.array_equal(got, want)
    if not ok:
        bad += 1
        print("case %s %s: %s" % (name, info, "shape/dtype" if got.shape != want.shape or got.dtype != want.dtype else "values"))
for case in range(n_cases):
    F, K = int(rng.integers(1, 50)), int(rng.choice([1, 2, 3, 5, 8, 16]))
    L = int(rng.integers(1, 50))
    T = L + int(rng.integers(0, 260))
    q = synth.background(rng, (F, K))
    X = synth.features(rng, T, q)
    if rng.integers(4) == 0:
        X = X * rng.integers(1, 255, size=X.shape).astype(np.uint8)      # any non-zero byte is an edge
    W, c = synth.bank(rng, q, [L])[0]
    info = "T=%d L=%d F=%d K=%d" % (T, L, F, K)
    check("detect", cllf.detect(X, W), oracle.detect(X, W), info)
    Xf = (X.astype(np.float32) * rng.random(X.shape).astype(np.float32))
    check("detect_float", cllf.detect_float(Xf, W), oracle.detect_float(Xf, W), info)
    check("detect_float2d", cllf.detect_float2d(Xf.reshape(T, -1), W.reshape(L, -1)), oracle.detect_float(Xf, W), info)
    M = int(rng.integers(1, 6))
    LF = np.stack([synth.bank(rng, q, [L])[0][0] for _ in range(M)])
    cs = rng.normal(size=M).astype(np.float32)
    g = cllf.detect_float_max(Xf, LF, cs)
    w = oracle.detect_float_max(Xf, LF, cs)
    check("detect_float_max.s", g[0], w[0], info); check("detect_float_max.i", g[1], w[1], info)
    Nn = int(rng.integers(1, 4))
    Fx = synth.features(rng, Nn * T, q).reshape(Nn, T, F, K)
    lens = rng.integers(1, T + 1, size=Nn).astype(np.int64)
    W64 = W.astype(np.float64)
    check("get_max_detect_examples", cllf.get_max_detect_examples(Fx, W64, lens), oracle.get_max_detect_examples(Fx, W64, lens), info)
    # parts
    P0, P1, Kp = int(rng.integers(1, 7)), int(rng.integers(1, 7)), int(rng.integers(1, 12))
    X8 = synth.features(rng, P0 + int(rng.integers(0, 40)), rng.uniform(0.05, 0.5, size=(P1 + int(rng.integers(0, 20)), 8)))
    lp, lip = synth.parts(rng, Kp, (P0, P1, 8))
    thr = int(rng.integers(-1, P0 * P1 * 4))
    pi = "X=%s patch=%dx%d K=%d thr=%d" % (X8.shape, P0, P1, Kp, thr)
    o = cp.code_parts(X8, lp, lip, thr)
    check("code_parts", o, oracle.code_parts(X8, lp, lip, thr), pi)
    check("code_parts_fast", cp.code_parts_fast(X8, lp, lip, thr), oracle.code_parts_fast(X8, lp, lip, thr), pi)
    A = np.argmax(o, -1)
    r0, r1 = int(rng.integers(0, 5)), int(rng.integers(0, 5))
    sp = swp.spread_waliji_patches(A, r0, r1, Kp)
    check("spread", sp, oracle.spread_patches(A, r0, r1, Kp), pi)
    check("spread_patches", cp.spread_patches(A, r0, r1, Kp), oracle.spread_patches(A, r0, r1, Kp), pi)
    gph = (rng.random((Kp, Kp)) < 0.2).astype(np.uint8)
    a, b = sp.copy(), sp.copy()
    cp.spread_graph_patches(a, gph); oracle.spread_graph_patches(b, gph)
    check("spread_graph", a, b, pi)
print("fuzz drop-ins: %d cases, %d bad" % (n_cases, bad))
