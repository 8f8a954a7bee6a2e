"""Time the edge front-end: 1000 spectrograms of 200 frames x 45 bins -> edge corpus."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "template-speech-recognition_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import tsr_hip
N, T, H = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, 200, 45
rng = np.random.default_rng(5)
S = rng.normal(size=(N * T, H)).cumsum(0) + 0.3 * rng.normal(size=(N * T, H))
offs = np.arange(N + 1, dtype=np.int64) * T
ctx = tsr_hip.default_ctx()
for rep in range(4):
    ctx.sync(); t0 = time.perf_counter()
    c = tsr_hip.Corpus.from_spectrograms(S, offs)
    dt = time.perf_counter() - t0
    if rep < 3:
        c.close()
print("edge front-end: %d frames x %d bins (%.0f MB f64 uploaded) -> %d-byte edge rows: %.2f ms per call "
      "(upload + kernel + corpus allocation), edge density %.3f" % (N * T, H, S.nbytes / 1e6, c.D, dt * 1e3, c.download_feats().mean()))
if "--cpu" in sys.argv:
    import oracle
    t0 = time.perf_counter()
    for n in range(10):
        oracle.get_edge_features(S[n * T:(n + 1) * T].T)
    print("NumPy restatement, one core: %.1f ms per utterance" % ((time.perf_counter() - t0) * 100))
