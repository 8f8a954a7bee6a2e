"""How long does a process take to reach its steady launch rate?  Config 2, R device-resident copies, consecutive steps of R
launches each, hipEvent time per step from the first launch of the process on (GPU box):
    python tools/time_steps.py [R=96] [steps=40] [flags=9]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "template-speech-recognition_amd"))
import numpy as np
import tsr_hip
from tsr_hip import synth
R = int(sys.argv[1]) if len(sys.argv) > 1 else 96
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 40
FLAGS = int(sys.argv[3]) if len(sys.argv) > 3 else 9
c = synth.corpus(2, N=1000, T=200, F=45, E=8, M=8, L=32)
ctx = tsr_hip.Context(0)
bank = tsr_hip.Bank(c["bank"], ctx=ctx)
reps = [tsr_hip.Corpus(c["feats"], c["offsets"], ctx=ctx) for _ in range(R)]
reps[0].score(bank, FLAGS)          # (builds the bank's and this corpus' derived forms)
ctx.sync()
out = []
for s in range(STEPS):
    ctx.timer_start()
    for cp in reps:
        cp.score(bank, FLAGS)
    out.append(ctx.timer_stop() / R * 1e3)
print("us per launch, step by step (%d launches each):" % R, " ".join("%.2f" % x for x in out))
t = np.cumsum(np.array(out) * R / 1e3)
print("ms since the first step began:", " ".join("%.0f" % x for x in t))
