"""Random fuzz of the SURVEY 8(f) widenings (GPU box): phone-window classification (both variants), peak
picking, edge front-end -- each against its oracle restatement, exact equality."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "template-speech-recognition_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import tsr_hip, oracle
from tsr_hip import synth
import template_speech_rec.get_train_data as gtrd
import template_speech_rec.roc_functions as rf
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    # ---- classification ----
    F, E = int(rng.integers(1, 12)), int(rng.choice([1, 4, 8]))
    M = int(rng.integers(1, 12))
    Ls = [int(x) for x in rng.integers(1, 25, size=M)]
    q = synth.background(rng, (F, E)).astype(np.float32)
    bank = synth.bank(rng, q, Ls)
    T = int(rng.integers(max(Ls) + 6, 160))
    Em = synth.features(rng, T, q)
    nb = int(rng.integers(1, max(2, T // 6)))
    cuts = np.sort(rng.choice(np.arange(3, T - 2), size=min(nb, T - 5), replace=False)) if T > 6 else np.array([], int)
    cuts = cuts[np.concatenate(([True], np.diff(cuts) >= 3))] if len(cuts) else cuts
    flts = np.concatenate(([0], cuts, [T])).astype(np.int64)
    nph = len(flts) - 1
    for iso in (False, True):
        got = [np.zeros((1, nph), dt) for dt in (np.float32, np.uint16, np.uint16, np.int16)]
        ref = [np.zeros((1, nph), dt) for dt in (np.float32, np.uint16, np.uint16, np.int16)]
        try:
            if iso:
                gtrd._compute_isolated_classification_E(Em, None, flts.copy(), 0, got[0], got[1], got[2], got[3], bank, q)
                oracle.compute_isolated_classification_E(Em, None, flts.copy(), 0, ref[0], ref[1], ref[2], ref[3], bank, q)
            else:
                gtrd._compute_classification_E(Em, None, flts.copy(), 0, got[0], got[1], got[2], got[3], bank)
                oracle.compute_classification_E(Em, None, flts.copy(), 0, ref[0], ref[1], ref[2], ref[3], bank)
            ok = all(np.array_equal(a, b) for a, b in zip(got, ref))
        except Exception as e:  # noqa: BLE001
            ok = False
            print("   exception:", repr(e)[:200])
        if not ok:
            bad += 1
            print("case %d classification iso=%s T=%d flts=%s Ls=%s D=%d" % (case, iso, T, flts.tolist(), Ls, F * E))
    # ---- peaks ----
    lens = rng.integers(0, 150, size=int(rng.integers(1, 8)))
    det = np.zeros((len(lens), int(lens.max()) + 2), np.float32)
    for n, Tn in enumerate(lens):
        det[n, :Tn] = np.round(rng.normal(size=Tn) * 3) / 3
    thr, C0 = float(rng.normal()), int(rng.integers(1, 40))
    got = rf.get_peaks_above_threshold(det, lens, thr, C0)
    if not all(np.array_equal(g, oracle.get_max_above_threshold(det[n], int(lens[n]), thr, C0)) for n, g in enumerate(got)):
        bad += 1
        print("case %d peaks lens=%s thr=%g C0=%d" % (case, lens.tolist(), thr, C0))
    ex = [[(int(s), int(s) + 5) for s in np.sort(rng.integers(0, max(1, Tn), size=int(rng.integers(0, 4))))] for Tn in lens]
    ws, we = int(rng.integers(-20, 1)), int(rng.integers(1, 30))
    g = rf.get_max_detection_in_syllable_windows(det, ex, lens, ws, we)
    w = oracle.get_max_detection_in_syllable_windows(det, ex, lens, ws, we)
    if not np.array_equal(g, np.asarray(w, np.float32)):
        bad += 1
        print("case %d window peaks" % case)
    C0c, C1c = int(rng.integers(1, 30)), int(rng.integers(1, 70))
    gc = rf._get_detect_clusters_single_threshold(thr, det, lens, C0c, C1c)
    wc = oracle.get_detect_clusters_single_threshold(thr, det, lens, C0c, C1c)
    if gc != tuple(tuple((int(a), int(b)) for a, b in u) for u in wc):
        bad += 1
        print("case %d clusters lens=%s thr=%g C0=%d C1=%d" % (case, lens.tolist(), thr, C0c, C1c))
    # ---- edges ----
    H = int(rng.integers(1, 40))
    Ts = [int(x) for x in rng.integers(0, 120, size=int(rng.integers(1, 5)))]
    BL, s, qq = int(rng.integers(1, 50)), int(rng.integers(0, 4)), float(rng.choice([0.0, 0.3, 0.7, 0.95]))
    if H * BL > 4096:
        BL = max(1, 4096 // H)
    specs = [np.round((rng.normal(size=(Tn, H)).cumsum(0)) * float(rng.choice([1, 4, 1000]))) / float(rng.choice([1, 4, 1000])) for Tn in Ts]
    offs = np.concatenate(([0], np.cumsum(Ts))).astype(np.int64)
    c = tsr_hip.Corpus.from_spectrograms(np.concatenate(specs), offs, block_length=BL, spread_length=s, threshold=qq)
    gotf = c.download_feats()
    for n, S in enumerate(specs):
        want = oracle.get_edge_features(S.T, block_length=BL, spread_length=s, threshold=qq)
        if not np.array_equal(gotf[offs[n]:offs[n + 1]].reshape(len(S), H, 8), want):
            bad += 1
            print("case %d edges H=%d T=%d BL=%d s=%d q=%g" % (case, H, len(S), BL, s, qq))
            break
print("fuzz next: %d cases, %d bad" % (n_cases, bad))
