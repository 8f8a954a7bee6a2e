import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "template-speech-recognition_amd"))
which = sys.argv[1]
if which == "torch_first":
    import torch, torch.distributed
import tsr_hip, numpy as np
from tsr_hip import synth
ctx = tsr_hip.Context(0)
uid = tsr_hip.unique_id()
ctx.dist_init(uid, 1, 0)
c = synth.corpus(1, N=2, T=20, F=3, E=2, M=2, L=4)
b = tsr_hip.Bank(c["bank"], ctx=ctx)
b.broadcast(0)
print(which, "ok", b.score(c["feats"], c["offsets"])[0][:2])
with open("/proc/self/maps") as f:
    libs = sorted({l.split()[-1] for l in f if "rccl" in l or "amdhip64" in l})
print(libs)
ctx.dist_finalize()
