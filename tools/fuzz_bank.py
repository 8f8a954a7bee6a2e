"""Random-geometry fuzz of the bank scorers (GPU box): exact vs oracle bit for bit; fast/fast24 vs exact
within 1e-4 of max(|score|, 1) AND within 1e-4 of the sum of magnitudes of the window's set taps (the
norm of include/tsr.h); ids equal wherever the top two templates are further apart than the bar."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "template-speech-recognition_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import tsr_hip, oracle
from tsr_hip import synth
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    N = int(rng.integers(1, 7))
    M = int(rng.choice([1, 2, 3, 8, 9, 17, 33]))
    F, E = int(rng.integers(1, 40)), int(rng.choice([1, 3, 8, 16, 50]))
    Ls = [int(x) for x in rng.integers(1, 70, size=M)]
    Ts = [int(x) for x in rng.integers(0, 400, size=N)]
    c = synth.corpus(int(rng.integers(1 << 30)), N=N, T=Ts, F=F, E=E, M=M, L=Ls)
    bank = tsr_hip.Bank(c["bank"])
    se, ae = bank.score(c["feats"], c["offsets"], flags=tsr_hip.TSR_EXACT)
    so, ao = oracle.score_corpus(c["feats"], c["offsets"], c["bank"])
    ok = np.array_equal(se, so) and np.array_equal(ae, ao)
    msg = "" if ok else " EXACT!=ORACLE"
    for flags in (1, 3):
        sf, af = bank.score(c["feats"], c["offsets"], flags=flags)
        err = float(np.max(np.abs(sf - se) / np.maximum(np.abs(se), 1.0))) if len(se) else 0.0
        if err > 1e-4:
            msg += " flags=%d err=%.3g" % (flags, err)
        allm = bank.score_all(c["feats"], c["offsets"], flags=0)
        if len(se):   # per template, against the sum of |W| over the window's set taps
            absbank = tsr_hip.Bank([(np.abs(W), 0.0) for W, _ in c["bank"]])
            scale = absbank.score_all(c["feats"], c["offsets"], flags=0)
            allf = bank.score_all(c["feats"], c["offsets"], flags=flags)
            e2 = float(np.max(np.abs(allf - allm) / np.maximum(scale, 1.0)))
            if e2 > 1e-4:
                msg += " flags=%d scaled err=%.3g" % (flags, e2)
        if M > 1 and len(se):
            top2 = np.sort(allm, axis=0)[-2:]
            clear = (top2[1] - top2[0]) > 1e-4 * np.maximum(np.abs(top2[1]), 1.0)
            if not np.array_equal(af[clear], ae[clear]):
                msg += " flags=%d ids differ on %d clear frames" % (flags, int((af[clear] != ae[clear]).sum()))
    if msg:
        bad += 1
        print("case %d N=%d M=%d D=%d Ls=%s Ts=%s:%s" % (case, N, M, F * E, Ls, Ts, msg))
print("fuzz: %d cases, %d bad" % (n_cases, bad))
