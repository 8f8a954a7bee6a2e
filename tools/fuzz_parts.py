"""Random-geometry fuzz of the corpus front-end (GPU box): MFMA coder vs f64 VALU kernel vs the oracle
chain, byte for byte, over random patch shapes, part counts, table ranges, thresholds, spreads, grids, graphs."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "template-speech-recognition_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import tsr_hip, oracle
from tsr_hip import synth
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
tot_loc = tot_re = 0
for case in range(n_cases):
    while True:
        P0, P1 = int(rng.integers(1, 9)), int(rng.integers(1, 9))
        if P0 * P1 <= 32:
            break
    K = int(rng.choice([1, 2, 5, 17, 31, 32, 33, 50, 63]))
    F = P1 + int(rng.integers(0, 30))
    N = int(rng.integers(1, 5))
    Ts = [int(x) for x in rng.integers(0, 90, size=N)]
    dens = float(rng.choice([0.02, 0.1, 0.3, 0.6]))
    q = rng.uniform(dens / 2, min(0.95, dens * 1.5), size=(F, 8))
    edges = [synth.features(rng, T, q) for T in Ts]
    kind = rng.choice(["uniform", "extreme", "flat", "dups"])
    if kind == "uniform":
        Pm = rng.uniform(0.01, 0.99, size=(K, P0, P1, 8))
    elif kind == "extreme":
        Pm = np.clip(rng.beta(0.2, 0.2, size=(K, P0, P1, 8)), 1e-6, 1 - 1e-6)
    elif kind == "flat":
        Pm = 0.5 + 1e-9 * rng.normal(size=(K, P0, P1, 8))
    else:
        Pm = rng.uniform(0.05, 0.95, size=(K, P0, P1, 8))
        for k in range(1, K, 3):
            Pm[k] = Pm[k - 1]
    lp, lip = np.log(Pm), np.log(1 - Pm)
    thr = int(rng.integers(-1, P0 * P1 * 8 // 2 + 2))
    fast = bool(rng.integers(2))
    spread = (int(rng.integers(0, 4)), int(rng.integers(0, 4)))
    grid = (int(rng.integers(1, 4)), int(rng.integers(1, 4)))
    graph = (rng.random((K, K)) < 0.05).astype(np.uint8) if rng.integers(3) == 0 else None
    offs = np.concatenate(([0], np.cumsum(Ts))).astype(np.int64)
    if offs[-1] == 0:
        continue
    corpus = tsr_hip.Corpus(np.concatenate(edges).reshape(int(offs[-1]), -1), offs)
    parts = tsr_hip.Parts(lp, lip)
    res = {}
    res_bank = ""
    for mode in (tsr_hip.TSR_PARTS_MFMA, tsr_hip.TSR_PARTS_VALU):
        if mode == tsr_hip.TSR_PARTS_MFMA and not parts.info()["mfma_available"]:
            continue
        parts.set_mode(mode)
        pc = corpus.part_features(parts, thr, spread=spread, graph=graph, grid=grid, fast=fast)
        if case % 3 == 0 and pc.P > 0:   # a random bank on the part corpus: the bit form (read as bits where the kernel can) vs byte rows
            rb = np.random.default_rng(case)
            O1 = pc.D // K
            filt = synth.bank(rb, synth.background(rb, (O1, K)), [int(x) for x in rb.integers(1, 40, size=int(rb.choice([1, 3, 8, 9])))])
            bank = tsr_hip.Bank(filt)
            pc.score(bank, tsr_hip.TSR_FAST)
            sb = pc.download() + (pc.score_all(bank, tsr_hip.TSR_FAST),)
            with tsr_hip.default_ctx().option(tsr_hip.TSR_OPT_PART_LAYOUT, tsr_hip.TSR_PART_LAYOUT_BYTES):
                pcb = corpus.part_features(parts, thr, spread=spread, graph=graph, grid=grid, fast=fast)
            pcb.score(bank, tsr_hip.TSR_FAST)
            sy = pcb.download() + (pcb.score_all(bank, tsr_hip.TSR_FAST),)
            pc.score(bank, tsr_hip.TSR_EXACT)
            se = pc.download()
            if not all(np.array_equal(x, y) for x, y in zip(sb, sy)):
                bank_bad = " bank scores differ between the bit form and byte rows"
            elif len(se[0]) and float(np.max(np.abs(sb[0] - se[0]) / np.maximum(np.abs(se[0]), 1.0))) > 1e-4:
                # (dense part features: the norm of include/tsr.h -- 1e-4 of the sum of magnitudes of the set taps)
                mags = tsr_hip.Bank([(np.abs(W), 0.0) for W, _ in filt]).score_all(pc.download_feats(), pc.offsets)
                ok2 = np.all(np.abs(sb[2].astype(np.float64) - pc.score_all(bank, tsr_hip.TSR_EXACT)) <= 1e-4 * np.maximum(mags, 1.0))
                bank_bad = "" if ok2 else " fast bank scores on the bit form off the exact ones"
            else:
                bank_bad = ""
            res_bank += bank_bad
            pcb.close(); bank.close()
        res[mode] = pc.download_feats()
        if mode == tsr_hip.TSR_PARTS_MFMA:
            i = parts.info()
            tot_loc += i["locations"]; tot_re += i["rechecked"]
    want = []
    for E in edges:
        if E.shape[0] < P0:
            continue
        fn = oracle.code_parts_fast if fast else oracle.code_parts
        sp = oracle.spread_patches(oracle.argmax_last(fn(E, lp, lip, thr)), spread[0], spread[1], K)
        if graph is not None:
            oracle.spread_graph_patches(sp, graph)
        w = np.ascontiguousarray(sp[::grid[0], ::grid[1]])
        want.append(w.reshape(w.shape[0], w.shape[1] * K))
    want = np.concatenate(want) if want else np.zeros((0, 0), np.uint8)
    msg = res_bank
    for mode, got in res.items():
        if got.shape[0] != want.shape[0] or (want.size and not np.array_equal(got, want)):
            msg += " mode %d differs from the oracle" % mode
    if len(res) == 1:
        msg += "" if parts.info()["mfma_available"] else ""
    if msg:
        bad += 1
        print("case %d K=%d patch=%dx%d F=%d Ts=%s kind=%s thr=%d fast=%s spread=%s grid=%s graph=%s:%s"
              % (case, K, P0, P1, F, Ts, kind, thr, fast, spread, grid, graph is not None, msg))
print("fuzz parts: %d cases, %d bad; MFMA coder re-evaluated %d of %d locations in f64" % (n_cases, bad, tot_re, tot_loc))
