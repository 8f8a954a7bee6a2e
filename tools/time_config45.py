"""BASELINE config 4 as one GPU of eight sees it / config 5: 1250 utterances x 200 frames, D=360, 64 templates
with L_m in 24..48 -- MFMA path (2 and 3 limbs) against the bit-exact VALU kernel on one MI355X."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "template-speech-recognition_amd"))
import numpy as np
import tsr_hip
from tsr_hip import synth
L = [24 + (7 * m) % 25 for m in range(64)]
c = synth.corpus(4, N=1250, T=200, F=45, E=8, M=64, L=L)
ctx = tsr_hip.default_ctx()
bank = tsr_hip.Bank(c["bank"])
corpus = tsr_hip.Corpus(c["feats"], c["offsets"])
frames = int(c["offsets"][-1])
flops = 2.0 * 360 * sum((200 - l + 1) * l for l in L) * 1250
for name, flags in (("fast (2 limbs)", 1), ("fast24 (3 limbs)", 3), ("exact (f32 VALU)", 0)):
    for _ in range(3):
        corpus.score(bank, flags)
    ctx.sync(); ctx.timer_start()
    for _ in range(10):
        corpus.score(bank, flags)
    ms = ctx.timer_stop() / 10
    print("%-18s %8.3f ms/pass  %7.1f M frames/s  %7.1f Tflop/s algorithmic" % (name, ms, frames / ms / 1e3, flops / ms / 1e9))
