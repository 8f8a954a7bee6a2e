"""Latency of the per-utterance drop-in calls (host arrays in, host arrays out) on one MI355X."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "template-speech-recognition_amd"))
import numpy as np
from tsr_hip import synth
import template_speech_rec.compute_likelihood_linear_filter as cllf
import template_speech_rec.code_parts as cp
import template_speech_rec.spread_waliji_patches as swp
rng = np.random.default_rng(0)
q = synth.background(rng, (45, 8))
E = synth.features(rng, 200, q)
W, c = synth.bank(rng, q, [32])[0]
lp, lip = synth.parts(rng, 50)
def t(fn, n=20):
    fn(); t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e3
print("detect            [200,45,8] x [32,45,8]: %.3f ms   (reference: 2.94 ms, SURVEY 8(a))" % t(lambda: cllf.detect(E, W)))
Ef = E.astype(np.float32)
print("detect_float      same shapes:            %.3f ms   (reference: 1.84 ms)" % t(lambda: cllf.detect_float(Ef, W)))
print("code_parts        [200,45,8], K=50:       %.3f ms   (reference: 44.4 ms)" % t(lambda: cp.code_parts(E, lp, lip, 12), 5))
A = np.argmax(cp.code_parts(E, lp, lip, 12), -1)
print("spread_patches    [196,41] r=2:           %.3f ms   (reference: 0.20 ms)" % t(lambda: swp.spread_waliji_patches(A, 2, 2, 50)))
