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
LARGE = "--large" in sys.argv   # corpora of 20 000 .. 120 000 frames: the kernels' steady-state tiles, segments cut
bad = 0                         # inside utterances, several column groups, the tap-pair variant as a third approximate path
for case in range(n_cases):
    if LARGE:
        N = int(rng.integers(30, 500))
        M = int(rng.choice([1, 3, 8, 9, 20]))
        F, E = int(rng.choice([5, 23, 45, 46])), int(rng.choice([3, 8]))
        Ls = [int(x) for x in rng.integers(1, 60, size=M)]
        Ts = [int(x) for x in rng.integers(0, 480, size=N)]
        if rng.random() < 0.3:
            Ts[int(rng.integers(N))] = int(rng.integers(3000, 40000))   # one very long utterance
    else:
        N = int(rng.integers(1, 7))
        M = int(rng.choice([1, 2, 3, 8, 9, 17, 33]))
        F, E = int(rng.integers(1, 40)), int(rng.choice([1, 3, 8, 16, 50]))
        Ls = [int(x) for x in rng.integers(1, 70, size=M)]
        Ts = [int(x) for x in rng.integers(0, 400, size=N)]
    c = synth.corpus(int(rng.integers(1 << 30)), N=N, T=Ts, F=F, E=E, M=M, L=Ls)
    bank = tsr_hip.Bank(c["bank"])
    se, ae = bank.score(c["feats"], c["offsets"], flags=tsr_hip.TSR_EXACT)
    if LARGE:   # the oracle on a few utterances; the vector kernel (TSR_EXACTK=valu) on all of them
        offs = c["offsets"]
        pick = rng.choice(N, size=min(N, 5), replace=False)
        ok = True
        for n in pick:
            sl = slice(int(offs[n]), int(offs[n + 1]))
            if sl.stop - sl.start > 2500 or sl.stop == sl.start:
                continue
            so, ao = oracle.score_bank(c["feats"][sl], c["bank"])
            ok = ok and np.array_equal(se[sl], so) and np.array_equal(ae[sl], ao)
        with tsr_hip.default_ctx().option(tsr_hip.TSR_OPT_EXACT_KERNEL, tsr_hip.TSR_EXACT_KERNEL_VALU):
            sv, av = bank.score(c["feats"], c["offsets"], flags=tsr_hip.TSR_EXACT)
        msg = ("" if ok else " EXACT!=ORACLE") + ("" if np.array_equal(se, sv) and np.array_equal(ae, av) else " EXACT(mfma)!=EXACT(valu)")
    else:
        so, ao = oracle.score_corpus(c["feats"], c["offsets"], c["bank"])
        ok = np.array_equal(se, so) and np.array_equal(ae, ao)
        msg = "" if ok else " EXACT!=ORACLE"
    for flags in ((1, 3, -1) if LARGE else (1, 3)):
        if flags == -1:   # the tap-pair kernel (falls back to the default kernel where its geometry does not fit)
            tsr_hip.default_ctx().set_option(tsr_hip.TSR_OPT_FAST_KERNEL, tsr_hip.TSR_FAST_KERNEL_TAP2)
            flags = 1
        else:
            tsr_hip.default_ctx().set_option(tsr_hip.TSR_OPT_FAST_KERNEL, tsr_hip.TSR_OPT_AUTO)
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
tsr_hip.default_ctx().set_option(tsr_hip.TSR_OPT_FAST_KERNEL, tsr_hip.TSR_OPT_AUTO)
print("fuzz%s: %d cases, %d bad" % (" (large corpora)" if LARGE else "", n_cases, bad))
