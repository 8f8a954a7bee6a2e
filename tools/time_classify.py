"""Time the phone-window classification paths on the device and against the CPU oracle."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "template-speech-recognition_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import tsr_hip
from tsr_hip import synth
import template_speech_rec.get_train_data as gtrd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
T, F, M, L = 200, 45, 8, 32
rng = np.random.default_rng(12)
q = synth.background(rng, (F, 8)).astype(np.float32)
bank_cs = synth.bank(rng, q, [L] * M)
utts = [synth.features(rng, T, q) for _ in range(N)]
flts = []
for _ in range(N):
    b = [0]
    while b[-1] < T:
        b.append(min(T, b[-1] + int(rng.integers(8, 24))))
    flts.append(np.array(b, np.int64))
nwin = sum(len(f) - 1 for f in flts)
ctx = tsr_hip.default_ctx()
bank = tsr_hip.Bank(bank_cs)
offs = np.arange(N + 1, dtype=np.int64) * T
corpus = tsr_hip.Corpus(np.concatenate(utts).reshape(N * T, -1), offs)
wins = [gtrd.get_classify_windows(None, f) for f in flts]
starts = np.concatenate([ws.astype(np.int64) + offs[n] for n, (ws, _) in enumerate(wins)])
ends = np.concatenate([we.astype(np.int64) + offs[n] for n, (_, we) in enumerate(wins)])
for flags, name in ((tsr_hip.TSR_EXACT, "exact"), (tsr_hip.TSR_FAST, "fast")):
    best = 1e9
    for _ in range(4):
        ctx.sync(); t0 = time.perf_counter()
        corpus.classify_windows(bank, starts, ends, flags)
        best = min(best, time.perf_counter() - t0)
    print("windowed arg-max over bank scores (%s): %d windows of %d utterances in %.3f ms (resident corpus; includes the %d-byte result download)"
          % (name, nwin, N, best * 1e3, nwin * 16))
# isolated windows, virtual (the kernel substitutes the background for the padding rows)
t0 = time.perf_counter()
per = [gtrd._isolated_virtual_windows(T, f, L, frame0=int(offs[n])) for n, f in enumerate(flts)]
cat = [np.concatenate([p[i] for p in per]) for i in range(5)]
t_loop = time.perf_counter() - t0
t0 = time.perf_counter()
batch = gtrd._isolated_virtual_windows_batch([T] * N, flts, L, [int(o) for o in offs[:-1]])
t_glue = time.perf_counter() - t0
assert all(np.array_equal(a, b) for a, b in zip(cat, batch[:5]))
print("isolated-window geometry on the host: %.1f ms as one pass of array operations (what get_classify_scores_E uses), "
      "%.1f ms as the per-utterance loop" % (t_glue * 1e3, t_loop * 1e3))
best = 1e9
for _ in range(4):
    ctx.sync(); t0 = time.perf_counter()
    corpus.classify_isolated(bank, q, *cat)
    best = min(best, time.perf_counter() - t0)
res_m = corpus.classify_isolated(bank, q, *cat)
print("isolated windows (virtual, resident corpus): %d windows, %d positions: window geometry on the host %.1f ms, "
      "item tables + kernels + download %.2f ms" % (len(cat[0]), int(cat[4].sum()), t_glue * 1e3, best * 1e3))
with ctx.option(tsr_hip.TSR_OPT_EXACT_KERNEL, tsr_hip.TSR_EXACT_KERNEL_VALU):   # the one-lane-per-position vector kernel
    best_v = 1e9
    for _ in range(4):
        ctx.sync(); t0 = time.perf_counter()
        res_v = corpus.classify_isolated(bank, q, *cat)
        best_v = min(best_v, time.perf_counter() - t0)
print("  the same through the vector kernel (TSR_EXACT_KERNEL_VALU): %.2f ms; results identical: %s"
      % (best_v * 1e3, all(np.array_equal(a, b) for a, b in zip(res_m, res_v))))
if "--materialised" in sys.argv:
    t0 = time.perf_counter()
    allw, nv = [], []
    for E, f in zip(utts, flts):
        ew, ws, we, pad = gtrd.get_isolated_classify_windows(E, None, f.copy(), q, bank_cs)
        allw.extend(ew); nv.extend((we.astype(np.int64) - ws.astype(np.int64)).tolist())
    t_glue = time.perf_counter() - t0
    rows = sum(len(w) for w in allw)
    ctx.sync(); t0 = time.perf_counter()
    bank.classify_float_windows(allw, nv)
    print("isolated windows (materialised f32): %d rows (%.0f MB): host glue %.1f ms, concat+upload+kernels+download %.1f ms"
          % (rows, rows * F * 8 * 4 / 1e6, t_glue * 1e3, (time.perf_counter() - t0) * 1e3))
if "--cpu" in sys.argv:
    import oracle
    n = min(N, 20)
    z = [np.zeros((n, 64), dt) for dt in (np.float32, np.uint16, np.uint16, np.int16)]
    t0 = time.perf_counter()
    for u in range(n):
        oracle.compute_classification_E(utts[u], None, flts[u].copy(), u, z[0], z[1], z[2], z[3], bank_cs)
    t1 = time.perf_counter()
    for u in range(n):
        oracle.compute_isolated_classification_E(utts[u], None, flts[u].copy(), u, z[0], z[1], z[2], z[3], bank_cs, q)
    t2 = time.perf_counter()
    print("CPU oracle, one core, %d utterances: %.1f ms/utt (windowed), %.1f ms/utt (isolated)"
          % (n, (t1 - t0) * 1e3 / n, (t2 - t1) * 1e3 / n))
