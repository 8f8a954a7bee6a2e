"""Host-side cost of one tsr_corpus_score call (enqueue only) against the kernel time it has to stay ahead of (GPU box):
    python tools/time_enqueue.py [launches=2000]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "template-speech-recognition_amd"))
import numpy as np
import tsr_hip
from tsr_hip import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
c = synth.corpus(2, N=1000, T=200, F=45, E=8, M=8, L=32)
ctx = tsr_hip.Context(0)
P = int(c["offsets"][-1])
bank = tsr_hip.Bank(c["bank"], ctx=ctx)
small = synth.corpus(3, N=4, T=200, F=45, E=8, M=8, L=32)
for name, corp in (("4 utterances (kernel ~ 0: the call's own cost)", small), ("config 2", c)):
    dc = tsr_hip.Corpus(corp["feats"].reshape(int(corp["offsets"][-1]), -1), corp["offsets"], ctx=ctx)
    for flags, fname in ((tsr_hip.TSR_FAST | tsr_hip.TSR_MX, "mx"), (tsr_hip.TSR_FAST, "fast")):
        for _ in range(20):
            dc.score(bank, flags)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(n):
            dc.score(bank, flags)
        t1 = time.perf_counter()
        ctx.sync()
        t2 = time.perf_counter()
        print("%-48s %-4s enqueue loop %.2f us per call, until the GPU is done %.2f us per call" % (name, fname, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
    dc.close()
