"""A/B of two builds of the bank kernel inside ONE process, interleaved rounds (cdna_hip_programming.md
rule 24): a context option (tsr_ctx_set_option) selects the kernel per launch.
    python tools/ab_kernel.py TSR_FASTK old,tap2 [R] [rounds] [config: 2 | 2b | 3 | 4] [flags]
    python tools/ab_kernel.py TSR_EXACTK valu,mfma ...
Streams over R device-resident copies of the corpus (working set >> L3 for R >= 8)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "template-speech-recognition_amd"))
import numpy as np
import tsr_hip
from tsr_hip import synth
VAR, VALS = sys.argv[1], sys.argv[2].split(",")
OPTION = {"TSR_FASTK": (tsr_hip.TSR_OPT_FAST_KERNEL, {"old": tsr_hip.TSR_FAST_KERNEL_INTERLEAVED, "tap2": tsr_hip.TSR_FAST_KERNEL_TAP2, "auto": 0}),
          "TSR_EXACTK": (tsr_hip.TSR_OPT_EXACT_KERNEL, {"valu": tsr_hip.TSR_EXACT_KERNEL_VALU, "mfma": tsr_hip.TSR_EXACT_KERNEL_MFMA, "auto": 0}),
          "TSR_XCD_MAP": (tsr_hip.TSR_OPT_XCD_MAP, {"0": 0, "1": 1})}[VAR]
R = int(sys.argv[3]) if len(sys.argv) > 3 else 24
ROUNDS = int(sys.argv[4]) if len(sys.argv) > 4 else 7
CONF = sys.argv[5] if len(sys.argv) > 5 else "2"
FLAGS = int(sys.argv[6]) if len(sys.argv) > 6 else tsr_hip.TSR_FAST
if CONF == "2":
    c = synth.corpus(2, N=1000, T=200, F=45, E=8, M=8, L=32)
elif CONF == "2b":
    c = synth.corpus(2, N=1000, T=200, F=45, E=8, M=8, L=[20, 24, 28, 32, 36, 40, 44, 48])
elif CONF == "3":   # the part-feature bank of config 3 (D = 2050) on random part features
    c = synth.corpus(3, N=250, T=196, F=41, E=50, M=8, L=32)
else:
    c = synth.corpus(4, N=1250, T=200, F=45, E=8, M=64, L=[24 + (7 * m) % 25 for m in range(64)])
ctx = tsr_hip.Context(0)
bank = tsr_hip.Bank(c["bank"], ctx=ctx)
P = int(c["offsets"][-1])
reps = [tsr_hip.Corpus(c["feats"].reshape(P, -1), c["offsets"], ctx=ctx) for _ in range(R)]
res = {v: [] for v in VALS}
out = {}
for rnd in range(ROUNDS + 1):
    for v in VALS:
        ctx.set_option(OPTION[0], OPTION[1][v])
        ctx.sync()
        ctx.timer_start()
        for _ in range(3):
            for cp in reps:
                cp.score(bank, FLAGS)
        ms = ctx.timer_stop() / (3 * R)
        if rnd:
            res[v].append(ms * 1e3)
        out[v] = reps[-1].download()
for v, t in res.items():
    print("config %s flags %d  %s=%s: median %.2f us  min %.2f us  (%d rounds x %d launches)"
          % (CONF, FLAGS, VAR, v, np.median(t), min(t), ROUNDS, 3 * R))
ref = out[VALS[0]]
print("outputs identical across variants:", all(np.array_equal(out[v][0], ref[0]) and np.array_equal(out[v][1], ref[1]) for v in VALS))
