#!/usr/bin/env python3
"""Golden vectors g15 = code_parts.code_parts_mag_aux (code_parts.pyx:138-242), produced by the reference's
own source compiled with Cython language_level=2 (oracle/_ref/py2: the function divides unsigned ints with `/`).
Build container only:   python tests/golden/make_golden_magaux.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import build_ref  # noqa: E402

assert build_ref.build(), "oracle/_ref could not be built (no /root/reference?)"
cp2 = build_ref.load_extra("code_parts", py2=True)
rng = np.random.default_rng(1500)
d = {}
cases = [(5, 5, 8, 6, 24, 14, 4, 3), (3, 4, 8, 3, 9, 11, 2, 1), (1, 1, 4, 2, 6, 5, 5, 2), (5, 5, 8, 10, 12, 45, 7, 3)]
for i, (P0, P1, E, K, X0, X1, MD, AD) in enumerate(cases):
    Xe = (rng.random((X0, X1, E)) < 0.3).astype(np.uint8)
    Xm = (rng.random((X0, MD)) < 0.4).astype(np.uint8)
    Xa = (rng.random((X0, AD)) < 0.5).astype(np.uint8)
    lp = np.log(rng.uniform(.05, .95, (K, P0, P1, E)))
    lmp = np.log(rng.uniform(.05, .95, (K, P0 + MD)))
    lap = np.log(rng.uniform(.05, .95, (K, P0 * AD)))
    lip, lmip, laip = (np.log(1 - np.exp(a)) for a in (lp, lmp, lap))
    mask = (rng.random((X0, X1)) < 0.7).astype(np.uint8)
    out = cp2.code_parts_mag_aux(Xe, Xm, Xa, lp, lip, lmp, lmip, lap, laip, mask)
    for name, a in (("Xe", Xe), ("Xm", Xm), ("Xa", Xa), ("lp", lp), ("lip", lip), ("lmp", lmp), ("lmip", lmip),
                    ("lap", lap), ("laip", laip), ("mask", mask), ("out", out)):
        d["c%d_%s" % (i, name)] = a
d["n"] = np.int64(len(cases))
path = os.path.join(HERE, "g15_code_parts_mag_aux.npz")
np.savez_compressed(path, **d)
print("g15_code_parts_mag_aux %.1f KB" % (os.path.getsize(path) / 1024))
