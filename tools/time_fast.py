"""Time the fast bank kernel on the config-2 corpus (device resident), no parity check."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "template-speech-recognition_amd"))
import numpy as np
import tsr_hip
from tsr_hip import synth
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8
FLAGS = int(sys.argv[2]) if len(sys.argv) > 2 else 1
c = synth.corpus(2, N=1000, T=200, F=45, E=8, M=M, L=32)
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
print("M=%d flags=%d: %.1f us/launch" % (M, FLAGS, best * 1e3))
