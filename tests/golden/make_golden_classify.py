#!/usr/bin/env python3
"""Golden vectors for the phone-window classification glue:
g12 = _compute_classification_E (get_train_data.py:718-773),
g13 = _compute_isolated_classification_E (:838-925, without the SVM stage).

The two callers are Python-2 modules that cannot be imported; their few NumPy lines are restated in
oracle/oracle.py and run here ON TOP OF THE REFERENCE'S COMPILED KERNELS (oracle/_ref: detect,
detect_float), so the fixtures hold what the reference's native code produces for these inputs.
Build container only:   python tests/golden/make_golden_classify.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "template-speech-recognition_amd"))

import build_ref  # noqa: E402
import oracle  # noqa: E402
from tsr_hip import synth  # noqa: E402

assert build_ref.build(), "oracle/_ref could not be built (no /root/reference?)"
cllf, _, _ = build_ref.load()


def phone_boundaries(rng, T, lo=6, hi=30):
    b = [0]
    while b[-1] < T:
        b.append(min(T, b[-1] + int(rng.integers(lo, hi + 1))))
    if b[-1] - b[-2] < 3 and len(b) > 2:   # no degenerate last phone
        del b[-2]
    return np.array(b, np.int64)


def run(seed, isolated, F=9, Kf=8, Ls=(7, 12, 9), Ts=(90, 61)):
    rng = np.random.default_rng(seed)
    q = synth.background(rng, (F, Kf)).astype(np.float32)
    bank = list(synth.bank(rng, q, list(Ls)))
    bank.append((bank[1][0].copy(), bank[1][1]))           # a duplicate template: ties go to the earlier one
    out = {"q": q, "M": np.int64(len(bank))}
    for m, (W, c) in enumerate(bank):
        out["W%d" % m], out["c%d" % m] = W, np.float64(c)
    nphn = 0
    utts = []
    for T in Ts:
        E = synth.features(rng, T, q)
        flts = phone_boundaries(rng, T)
        utts.append((E, flts))
        nphn = max(nphn, len(flts) - 1)
    N = len(utts)
    arr = np.zeros((N, nphn), np.float32)
    ids = np.zeros((N, nphn), np.uint16)
    lens = np.zeros((N, nphn), np.uint16)
    locs = np.zeros((N, nphn), np.int16)
    for u, (E, flts) in enumerate(utts):
        out["E%d" % u] = np.packbits(E.reshape(-1))
        out["Eshape%d" % u] = np.array(E.shape, np.int64)
        out["flts%d" % u] = flts.copy()
        if isolated:
            oracle.compute_isolated_classification_E(E, None, flts.copy(), u, arr, ids, lens, locs, bank, q,
                                                     _detect_float=cllf.detect_float)
        else:
            oracle.compute_classification_E(E, None, flts.copy(), u, arr, ids, lens, locs, bank,
                                            _detect=cllf.detect)
    out.update(N=np.int64(N), classify_array=arr, classify_template_ids=ids, classify_template_lengths=lens,
               classify_locs=locs)
    return out


for name, iso in (("g12_classification_E", False), ("g13_isolated_classification_E", True)):
    d = run(1200 + iso, iso)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **d)
    print("%-34s %8.1f KB" % (name, os.path.getsize(path) / 1024))
