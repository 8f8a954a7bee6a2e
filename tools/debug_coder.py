"""Development aid (GPU box): where do the MFMA coder's ids differ from the VALU kernel's?"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "template-speech-recognition_amd"))
import numpy as np
import tsr_hip
from tsr_hip import synth
for (P0, P1, F, K, Ts, thr) in [(1, 4, 15, 50, [69, 13, 75], 1), (4, 1, 1, 5, [43, 6, 3], 11), (1, 5, 11, 5, [29, 28, 28, 36], 1), (5, 5, 45, 50, [200, 100], 10), (2, 3, 9, 17, [300], 3)]:
    rng = np.random.default_rng(1)
    q = rng.uniform(0.1, 0.4, size=(F, 8))
    edges = [synth.features(rng, T, q) for T in Ts]
    Pm = rng.uniform(0.01, 0.99, size=(K, P0, P1, 8))
    lp, lip = np.log(Pm), np.log(1 - Pm)
    offs = np.concatenate(([0], np.cumsum(Ts))).astype(np.int64)
    corpus = tsr_hip.Corpus(np.concatenate(edges).reshape(int(offs[-1]), -1), offs)
    parts = tsr_hip.Parts(lp, lip)
    ids = {}
    for mode in (tsr_hip.TSR_PARTS_MFMA, tsr_hip.TSR_PARTS_VALU):
        parts.set_mode(mode)
        pc = corpus.part_features(parts, thr, spread=(0, 0))
        f = pc.download_feats()
        N1 = F - P1 + 1
        f = f.reshape(f.shape[0], N1, K)
        ids[mode] = np.where(f.any(axis=2), f.argmax(axis=2) + 1, 0)
    a, b = ids[tsr_hip.TSR_PARTS_MFMA], ids[tsr_hip.TSR_PARTS_VALU]
    d = np.argwhere(a != b)
    print("patch %dx%d F=%d K=%d Ts=%s: %d of %d ids differ; info %s" % (P0, P1, F, K, Ts, len(d), a.size, parts.info()))
    for r, c in d[:12]:
        print("   row %d bin %d: mfma %d valu %d" % (r, c, a[r, c], b[r, c]))
