"""Random-geometry fuzz of TSR_FAST | TSR_MX (GPU box):  python tools/fuzz_mx.py [cases] [seed] [--large]
Where the MX kernel serves the call (TSR_INFO_LAST_BANK_KERNEL), its result must equal the integer emulation of its own
arithmetic (tests/test_gpu_mx.py::_emulate) to the last f32 rounding, stay within the worst-case bound of include/tsr.h
against TSR_EXACT (2^-13 max|W_m| per set tap) and give the same ids wherever the top two templates are further apart than
twice that bound; where the call falls outside the kernel's envelope the result must be plain TSR_FAST's bit for bit."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "template-speech-recognition_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import tsr_hip
from tsr_hip import synth
from test_gpu_mx import _emulate
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 0)
LARGE = "--large" in sys.argv
ctx = tsr_hip.default_ctx()
bad = served = 0
for case in range(n_cases):
    M = int(rng.choice([1, 2, 3, 5, 7, 8, 8, 8, 9, 12]))
    F, E = [(45, 8), (16, 8), (64, 8), (26, 5), (33, 3), (9, 16), (80, 8), (6, 5), (131, 8), (41, 50)][int(rng.integers(10))]   # (D > 512: feature phases)
    D = F * E
    Lmin = max(2, 4096 // D + 1) if rng.random() < 0.9 else 2
    Ls = [int(x) for x in rng.integers(Lmin, Lmin + 30, size=M)]
    if rng.random() < 0.5:
        Ls = [Ls[0]] * M
    if LARGE:
        N = int(rng.integers(40, 400))
        Ts = [int(x) for x in rng.integers(0, 420, size=N)]
        if rng.random() < 0.3:
            Ts[int(rng.integers(N))] = int(rng.integers(3000, 20000))
    else:
        N = int(rng.integers(1, 9))
        Ts = [int(x) for x in rng.integers(0, 300, size=N)]
    c = synth.corpus(int(rng.integers(1 << 30)), N=N, T=Ts, F=F, E=E, M=M, L=Ls)
    P = int(c["offsets"][-1])
    f2 = c["feats"].reshape(P, -1)
    bank = tsr_hip.Bank(c["bank"])
    s, a = bank.score(f2, c["offsets"], flags=tsr_hip.TSR_FAST | tsr_hip.TSR_MX)
    mx = P > 0 and ctx.get_option(tsr_hip.TSR_INFO_LAST_BANK_KERNEL) == tsr_hip.TSR_KERNEL_MX
    msg = ""
    if not mx:
        sf, af = bank.score(f2, c["offsets"], flags=tsr_hip.TSR_FAST)
        if not (np.array_equal(s, sf) and np.array_equal(a, af)):
            msg += " fallback != TSR_FAST"
        if P and min(Ls) * D > 4096 and (D <= 512 or max(Ls) >= 12):   # (wide rows: at least 24 K-blocks per phase)
            msg += " inside the envelope but not served by the MX kernel"
    else:
        served += 1
        sub = slice(0, int(c["offsets"][min(N, 60)]))          # the emulation on the first 60 utterances
        es, ea, vals, slack = _emulate(f2[sub], c["offsets"][:min(N, 60) + 1], c["bank"])
        if not np.all(np.abs(s[sub] - es) <= slack + 1e-30):
            msg += " != emulation (%.3g over the slack)" % float(np.max(np.abs(s[sub] - es) - slack))
        if vals.shape[0] > 1 and es.size:
            top2 = np.sort(vals, axis=0)[-2:]
            clear = (top2[1] - top2[0]) > 4 * slack
            if not np.array_equal(a[sub][clear], ea[clear]):
                msg += " ids != emulation on %d clear frames" % int((a[sub][clear] != ea[clear]).sum())
        se, ae = bank.score(f2, c["offsets"], flags=tsr_hip.TSR_EXACT)
        wmax = max(float(np.abs(W).max()) for W, _ in c["bank"])
        nset = np.zeros(P)
        for n in range(N):                                      # set taps of the longest window at each start
            lo, hi = int(c["offsets"][n]), int(c["offsets"][n + 1])
            cs = np.concatenate([[0], np.cumsum(f2[lo:hi].sum(1, dtype=np.int64))])
            for t in range(hi - lo):
                nset[lo + t] = cs[min(hi - lo, t + max(Ls))] - cs[t]
        bound = nset * wmax / 8190.0 + 1e-4 * np.maximum(np.abs(se), 1.0) * 0 + np.abs(se) * 2e-7 + 1e-6
        if not np.all(np.abs(s - se) <= bound):
            msg += " outside the worst-case bound against TSR_EXACT (%.3g)" % float(np.max(np.abs(s - se) - bound))
        err = float(np.max(np.abs(s - se) / np.maximum(np.abs(se), 1.0)))
        if err > 1e-4 and min(Ls) * D * 0.1 > 200:             # (scores of order >= 10^2: the bar of the BASELINE shapes)
            msg += " rel err %.3g" % err
    bank.close()
    if msg:
        bad += 1
        print("case %d N=%d M=%d D=%d Ls=%s Ts=%s:%s" % (case, N, M, D, Ls, Ts[:12], msg))
print("fuzz_mx%s: %d cases (%d served by the MX kernel), %d bad" % (" (large corpora)" if LARGE else "", n_cases, served, bad))
