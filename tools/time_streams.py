"""Independent corpora scored on S contexts (= S HIP streams) of one device, launches dealt round-robin (GPU box):
    python tools/time_streams.py [S=2] [copies=32] [flags=9] [passes=10] [utterances per corpus=1000]
Prints the time per launch over all streams against one stream alone."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "template-speech-recognition_amd"))
import numpy as np
import tsr_hip
from tsr_hip import synth
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
R = int(sys.argv[2]) if len(sys.argv) > 2 else 32
FLAGS = int(sys.argv[3]) if len(sys.argv) > 3 else 9
NU = int(sys.argv[5]) if len(sys.argv) > 5 else 1000
c = synth.corpus(2, N=NU, T=200, F=45, E=8, M=8, L=32)
P = int(c["offsets"][-1])
ctxs = [tsr_hip.Context(0) for _ in range(S)]
banks = [tsr_hip.Bank(c["bank"], ctx=x) for x in ctxs]
reps = [[tsr_hip.Corpus(c["feats"], c["offsets"], ctx=ctxs[s]) for _ in range(R)] for s in range(S)]
def run(streams, passes):
    for s in streams:
        for cp in reps[s]:
            cp.score(banks[s], FLAGS)
    for x in ctxs:
        x.sync()
    t0 = time.perf_counter()
    for _ in range(passes):
        for i in range(R):
            for s in streams:
                reps[s][i].score(banks[s], FLAGS)
    for x in ctxs:
        x.sync()
    return (time.perf_counter() - t0) / (passes * R * len(streams))
PASSES = int(sys.argv[4]) if len(sys.argv) > 4 else 10
# alternately, several times: the first timed loops of a process run slower than the later ones (clocks), whatever they are
ones, alls = [], []
for rnd in range(4):
    ones.append(run([0], PASSES))
    alls.append(run(list(range(S)), PASSES))
print("flags=%d copies per stream=%d, %d launches per loop; us per launch, one stream: %s; %d streams: %s; last round %.2fx"
      % (FLAGS, R, PASSES * R, " ".join("%.2f" % (x * 1e6) for x in ones), S, " ".join("%.2f" % (x * 1e6) for x in alls), ones[-1] / alls[-1]))
s0, a0 = reps[0][0].download(); s1, a1 = reps[S - 1][R - 1].download()
print("first and last copy score alike:", bool(np.array_equal(s0, s1) and np.array_equal(a0, a1)))
