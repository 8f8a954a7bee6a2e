"""Time a bank pass on the config-2 corpus (device resident), no parity check:
    python tools/time_fast.py [M=8] [flags=1] [L = 32 | lo-hi: filter lengths spread evenly over lo .. hi]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "template-speech-recognition_amd"))
import numpy as np
import tsr_hip
from tsr_hip import synth
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8
FLAGS = int(sys.argv[2]) if len(sys.argv) > 2 else 1
LARG = sys.argv[3] if len(sys.argv) > 3 else "32"
if "-" in LARG:
    lo, hi = (int(x) for x in LARG.split("-"))
    L = [lo + (i * (hi - lo)) // max(M - 1, 1) for i in range(M)]
else:
    L = int(LARG)
c = synth.corpus(2, N=1000, T=200, F=45, E=8, M=M, L=L)
ctx = tsr_hip.Context(0)
bank = tsr_hip.Bank(c["bank"], ctx=ctx)
cp = tsr_hip.Corpus(c["feats"], c["offsets"], ctx=ctx)
for _ in range(5):
    cp.score(bank, FLAGS)
ctx.sync()
best = 1e9
for rep in range(5):
    ctx.timer_start()
    for _ in range(20):
        cp.score(bank, FLAGS)
    best = min(best, ctx.timer_stop() / 20)
print("M=%d flags=%d L=%s: %.1f us/launch (kernel %d)" % (M, FLAGS, LARG, best * 1e3, ctx.get_option(tsr_hip.TSR_INFO_LAST_BANK_KERNEL)))
