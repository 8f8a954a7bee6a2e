"""Time BASELINE config 3 on the device: edge corpus -> code_parts / argmax / spread -> D = 2050 bank, with the part
corpus resident as bit rows (default) and as byte rows (round 2's layout), hipEvent times per stage.
    python tools/time_config3.py [N=1000] [bits,bytes] [reps=5] [fast|mx]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "template-speech-recognition_amd"))
import numpy as np
import tsr_hip
from tsr_hip import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
LAYOUTS = (sys.argv[2] if len(sys.argv) > 2 else "bits,bytes").split(",")
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 5
FLAGS = tsr_hip.TSR_FAST | (tsr_hip.TSR_MX if len(sys.argv) > 4 and sys.argv[4] == "mx" else 0)
OPT = {"bits": tsr_hip.TSR_PART_LAYOUT_BITS, "bytes": tsr_hip.TSR_PART_LAYOUT_BYTES, "auto": tsr_hip.TSR_OPT_AUTO}
rng = np.random.default_rng(3)
q = synth.background(rng, (45, 8))
feats = (rng.random((N * 200, 45, 8)) < q).astype(np.uint8)
offs = np.arange(N + 1, dtype=np.int64) * 200
lp, lip = synth.parts(rng, 50)
ctx = tsr_hip.Context(0)
corpus = tsr_hip.Corpus(feats.reshape(N * 200, -1), offs, ctx=ctx)
parts = tsr_hip.Parts(lp, lip, ctx=ctx)
bank = tsr_hip.Bank(synth.bank(rng, synth.background(rng, (41, 50)), [32] * 8), ctx=ctx)
ref = None
for layout in LAYOUTS:
    ctx.set_option(tsr_hip.TSR_OPT_PART_LAYOUT, OPT[layout])
    fe_wall, fe_gpu, bank_gpu = [], [], []
    for rep in range(REPS + 1):
        ctx.sync(); t0 = time.perf_counter(); ctx.timer_start()
        pc = corpus.part_features(parts, 12, spread=(2, 2))
        g = ctx.timer_stop(); w = (time.perf_counter() - t0) * 1e3
        pc.score(bank, FLAGS); ctx.sync()
        ctx.timer_start()
        for _ in range(4):
            pc.score(bank, FLAGS)
        b = ctx.timer_stop() / 4
        if rep:
            fe_wall.append(w); fe_gpu.append(g); bank_gpu.append(b)
        out = pc.download()
        lay = pc.layout()
        pc.close()
    if ref is None:
        ref = out
    same = np.array_equal(out[0], ref[0]) and np.array_equal(out[1], ref[1])
    frames = N * 200
    tot = min(fe_gpu) + min(bank_gpu)
    print("N=%d layout=%-5s front-end %.3f ms on the stream (%.2f ms wall), D=2050 bank %.3f ms, sum %.3f ms = %.1f M frames/s; "
          "part corpus %.1f MB resident (%s); scores identical to the first layout: %s"
          % (N, layout, min(fe_gpu), min(fe_wall), min(bank_gpu), tot, frames / tot / 1e3, lay["feature_bytes"] / 1e6,
             "+".join(k for k in ("bits", "bytes") if lay[k]), same))
print("coder:", parts.info(), "bank kernel:", ctx.get_option(tsr_hip.TSR_INFO_LAST_BANK_KERNEL))
