"""The whole device-resident chain on one MI355X: spectrograms -> edge corpus -> part corpus -> bank scores ->
peaks, 1000 utterances x 200 frames x 45 bins (BASELINE config 3 with the edge front-end in front of it)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "template-speech-recognition_amd"))
import numpy as np
import tsr_hip
from tsr_hip import synth
N, T, H, K = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, 200, 45, 50
rng = np.random.default_rng(9)
S = rng.normal(size=(N * T, H)).cumsum(0) + 0.3 * rng.normal(size=(N * T, H))
offs = np.arange(N + 1, dtype=np.int64) * T
ctx = tsr_hip.default_ctx()
lp, lip = synth.parts(rng, K)
parts = tsr_hip.Parts(lp, lip)
bank = tsr_hip.Bank(synth.bank(rng, synth.background(rng, (H - 4, K)), [32] * 8))
best = None
thr = -1e30
for rep in range(4):
    ctx.sync(); t = [time.perf_counter()]
    edges = tsr_hip.Corpus.from_spectrograms(S, offs); t.append(time.perf_counter())
    pc = edges.part_features(parts, 12, spread=(2, 2)); t.append(time.perf_counter())
    pc.score(bank, tsr_hip.TSR_FAST); ctx.sync(); t.append(time.perf_counter())
    pk = tsr_hip.detect_clusters(thr, 32, 48, corpus=pc); t.append(time.perf_counter())
    d = np.diff(t) * 1e3
    if rep == 0:
        thr = float(np.quantile(pc.download()[0], 0.98))   # a detection threshold that lets 2 % of the frames through
    if rep and (best is None or d.sum() < best.sum()):
        best = d
    edges.close(); pc.close()
print("%d utterances x %d frames: spectrogram upload + edges %.2f ms | parts coder + dilation %.2f ms | D=%d bank %.2f ms | "
      "peaks + clusters %.2f ms | total %.2f ms = %.1f M frames/s (%d clustered detections)"
      % (N, T, best[0], best[1], (H - 4) * K, best[2], best[3], best.sum(), N * T / best.sum() / 1e3, sum(len(p) for p in pk)))
