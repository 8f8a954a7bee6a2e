"""BASELINE config 4 as one GPU of eight sees it / config 5: 1250 utterances x 200 frames, D=360, 64 templates
with L_m in 24..48, R device-resident copies scored in turn (R x 92 MB: past the Infinity Cache for R >= 4).
    python tools/time_config45.py [modes: comma list of fast,fast24,exact,bf16x2,bf16,mx] [R] [passes]
Prints one line per mode: ms per pass, frames/s, algorithmic Tflop/s, max error against TSR_EXACT."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "template-speech-recognition_amd"))
import numpy as np
import tsr_hip
from tsr_hip import synth
MODES = (sys.argv[1] if len(sys.argv) > 1 else "fast,fast24,exact").split(",")
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4
PASSES = int(sys.argv[3]) if len(sys.argv) > 3 else 5
FLAGS = {"exact": 0, "fast": 1, "fast24": 3, "bf16x2": getattr(tsr_hip, "TSR_FAST", 1) | getattr(tsr_hip, "TSR_BF16", 4) | 2,
         "bf16": getattr(tsr_hip, "TSR_FAST", 1) | getattr(tsr_hip, "TSR_BF16", 4),
         "mx": getattr(tsr_hip, "TSR_FAST", 1) | getattr(tsr_hip, "TSR_MX", 8)}   # (round 4: eight slices of eight templates)
L = [24 + (7 * m) % 25 for m in range(64)]
c = synth.corpus(4, N=1250, T=200, F=45, E=8, M=64, L=L)
ctx = tsr_hip.default_ctx()
bank = tsr_hip.Bank(c["bank"])
reps = [tsr_hip.Corpus(c["feats"], c["offsets"]) for _ in range(R)]
frames = int(c["offsets"][-1])
flops = 2.0 * 360 * sum((200 - l + 1) * l for l in L) * 1250
reps[0].score(bank, 0)
exact, _ = reps[0].download()
for name in MODES:
    fl = FLAGS[name]
    for cp in reps:
        cp.score(bank, fl)
    ctx.sync(); ctx.timer_start()
    for _ in range(PASSES):
        for cp in reps:
            cp.score(bank, fl)
    ms = ctx.timer_stop() / (PASSES * R)
    got, _ = reps[-1].download()
    err = float(np.max(np.abs(got - exact) / np.maximum(np.abs(exact), 1.0)))
    print("%-8s %8.3f ms/pass  %7.1f M frames/s  %7.1f Tflop/s algorithmic  max |a-b|/max(|b|,1) vs exact %.2e  (R=%d, %d passes)"
          % (name, ms, frames / ms / 1e3, flops / ms / 1e9, err, R, PASSES))
