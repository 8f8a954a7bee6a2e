"""GPU parity tests of TSR_FAST | TSR_MX (csrc/detect_mx.hip): the bank pass as a skew-free Toeplitz product on the
block-scaled matrix instruction, 13-bit weights as three fp6 digits.

Two checkers.  (1) The oracle (the reference's f32 sums, compute_likelihood_linear_filter.pyx:162-175 + the bank epilogue
of get_train_data.py:962-992): |a - b| <= 1e-4 max(|b|, 1) over every frame, ids equal wherever the top-2 margin is clear,
tail frames exactly max_m c_m.  (2) An integer emulation of the kernel's own arithmetic written out here in NumPy
(quantise with the per-template scale, exact integer window sums, one multiply, one add): the kernel's sums are exact
integers, so it must reproduce that emulation to the last f32 rounding -- which pins the operand layout, the digit split,
the block scales, the cross-wave reduction and the tile plan far more tightly than the 1e-4 bar can."""
import numpy as np
import pytest

from helpers import rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def tsr_hip():
    import tsr_hip as t
    return t


def _emulate(feats2d, offs, bank, qmax=4095):
    """The kernel's arithmetic: q = rint(W * qmax / max|W_m|), exact integer sums over the set taps, score =
    f32(sum) * f32(1 / scale) + f32(c_m), zero sum where the window leaves the utterance; max over m, lowest id on ties.
    Returns (scores f64 [P], ids, per-template values f64 [M, P], rounding slack [P])."""
    P = int(offs[-1])
    M = len(bank)
    vals = np.zeros((M, P))
    slack = np.zeros((M, P))
    X = feats2d.astype(np.float64)
    for m, (W, c) in enumerate(bank):
        L = W.shape[0]
        W2 = W.reshape(L, -1).astype(np.float64)
        mx = np.abs(W2).max()
        scale = qmax / mx if mx > 0 else 1.0
        q = np.rint(W2 * scale)
        inv = float(np.float32(1.0 / scale))
        cf = float(np.float32(c))
        G = X @ q.T                                   # [P, L] exact integers
        for n in range(len(offs) - 1):
            a, b = int(offs[n]), int(offs[n + 1])
            T = b - a
            raw = np.zeros(T)
            if T >= L:
                for tau in range(L):
                    raw[:T - L + 1] += G[a + tau:a + tau + T - L + 1, tau]
            vals[m, a:b] = raw * inv + cf
            slack[m, a:b] = (np.abs(raw * inv) + abs(cf)) * 2.0 ** -22
    ids = np.argmax(vals, axis=0)
    return vals.max(axis=0), ids, vals, slack.max(axis=0)


def _check(tsr_hip, oracle, c, with_oracle=True, tol=TOL):
    feats, offs = c["feats"], c["offsets"]
    P = int(offs[-1])
    f2 = feats.reshape(P, -1)
    bank = tsr_hip.Bank(c["bank"])
    s, a = bank.score(f2, offs, flags=tsr_hip.TSR_FAST | tsr_hip.TSR_MX)
    assert s.shape == (P,) and a.shape == (P,)
    if P:                                      # (no silent service by TSR_FAST's kernels: its 16-bit sums differ from the emulation)
        assert tsr_hip.default_ctx().get_option(tsr_hip.TSR_INFO_LAST_BANK_KERNEL) == tsr_hip.TSR_KERNEL_MX
    es, ea, vals, slack = _emulate(f2, offs, c["bank"])
    assert np.all(np.abs(s - es) <= slack + 1e-30), float(np.max(np.abs(s - es) - slack))
    if vals.shape[0] > 1 and P:
        top2 = np.sort(vals, axis=0)[-2:]
        clear = (top2[1] - top2[0]) > 4 * slack
        assert np.array_equal(a[clear], ea[clear])
    elif P:
        assert np.all(a == 0)
    if with_oracle and P:
        so, ao = oracle.score_corpus(feats, offs, c["bank"])
        assert rel_err(s, so) <= tol
    bank.close()
    return s, a


def test_mx_is_the_matrix_instruction_kernel(tsr_hip, oracle):
    """Config 1's geometry with eight templates: one workgroup, partially filled tiles."""
    from tsr_hip import synth
    c = synth.corpus(1, N=1, T=400, F=45, E=8, M=8, L=32)
    _check(tsr_hip, oracle, c)


@pytest.mark.parametrize("shape", [
    dict(N=7, T=[200, 199, 33, 32, 31, 5, 64], F=45, E=8, M=8, L=32),            # odd lengths, T = L, T < L
    dict(N=5, T=[90, 0, 61, 1, 47], F=45, E=8, M=5, L=[20, 24, 28, 32, 36]),      # an empty utterance, ragged L_m, M < 8
    dict(N=4, T=[150, 77, 300, 41], F=16, E=8, M=8, L=40),                         # D = 128: one K-block per tap row
    dict(N=6, T=[120, 64, 65, 66, 67, 230], F=64, E=8, M=3, L=[36, 33, 40]),       # D = 512: four K-blocks per row
    dict(N=3, T=[500, 45, 46], F=26, E=5, M=1, L=45),                              # one template, D = 130 off every grid
    dict(N=40, T=70, F=45, E=8, M=7, L=[32, 32, 33, 34, 35, 36, 37]),              # several workgroups' worth of short utterances
], ids=["odd_T", "empty_utt", "D128", "D512", "M1_D130", "short_utts"])
def test_mx_small_against_emulation_and_oracle(tsr_hip, oracle, shape):
    from tsr_hip import synth
    c = synth.corpus(77, **shape)
    _check(tsr_hip, oracle, c)


def test_mx_ties_go_to_the_lowest_id(tsr_hip, oracle):
    from tsr_hip import synth
    c = synth.corpus(5, N=3, T=[80, 120, 57], F=45, E=8, M=4, L=32)
    b = list(c["bank"])
    b[2] = b[0]            # templates 0 and 2 identical: every frame is a tie between them
    b[3] = b[1]
    c["bank"] = tuple(b)
    s, a = _check(tsr_hip, oracle, c)
    assert set(np.unique(a)) <= {0, 1}


def test_mx_config2_all_frames(tsr_hip, oracle):
    """BASELINE config 2 (1000 x 200 frames, D = 360, M = 8, L = 32) -- what bench.py times -- every frame against the
    oracle and against the emulation, the tail rule, invariance to batching, and the other precisions beside it."""
    from tsr_hip import synth
    c = synth.corpus(2, N=1000, T=200, F=45, E=8, M=8, L=32)
    feats, offs = c["feats"], c["offsets"]
    P = int(offs[-1])
    f2 = feats.reshape(P, -1)
    bank = tsr_hip.Bank(c["bank"])
    dc = tsr_hip.Corpus(f2, offs)
    dc.score(bank, tsr_hip.TSR_FAST | tsr_hip.TSR_MX)
    s, a = dc.download()
    dc.score(bank, tsr_hip.TSR_FAST | tsr_hip.TSR_MX)       # second pass on the cached bit rows and plan
    s1, a1 = dc.download()
    assert np.array_equal(s, s1) and np.array_equal(a, a1)
    so, ao = oracle.score_corpus(feats, offs, c["bank"])
    err = rel_err(s, so)
    assert err <= TOL, err
    assert err <= 4e-5, err                                  # (observed 2e-5: the 13-bit weights, not a layout slip)
    dc.score(bank, tsr_hip.TSR_EXACT)
    se, ae = dc.download()
    alle = dc.score_all(bank, tsr_hip.TSR_EXACT)
    assert np.array_equal(se, so) and np.array_equal(ae, ao)
    top2 = np.sort(alle, axis=0)[-2:]
    clear = (top2[1] - top2[0]) > TOL * np.maximum(np.abs(top2[1]), 1.0)
    assert np.array_equal(a[clear], ao[clear]) and clear.mean() > 0.5
    cf = np.array([np.float32(cc) for _, cc in c["bank"]])
    assert np.all(s.reshape(1000, 200)[:, 169:] == cf.max()) and np.all(a.reshape(1000, 200)[:, 169:] == int(np.argmax(cf)))
    es, ea, vals, slack = _emulate(f2[:offs[40]], offs[:41], c["bank"])      # the emulation on the first 40 utterances
    assert np.all(np.abs(s[:offs[40]] - es) <= slack)
    s2, a2 = bank.score(f2[offs[500]:], offs[500:] - offs[500], flags=tsr_hip.TSR_FAST | tsr_hip.TSR_MX)   # second half alone
    assert np.array_equal(s2, s[offs[500]:]) and np.array_equal(a2, a[offs[500]:])
    dc.close()
    bank.close()


def test_mx_ragged_large_corpus(tsr_hip, oracle):
    """P > 60 000 frames of ragged utterances (some shorter than every template, odd lengths) with variable template
    lengths: tiles straddle utterances, workgroups end where the LDS image is full."""
    from tsr_hip import synth
    rng = np.random.default_rng(606)
    T = rng.integers(1, 401, size=420)
    T[rng.choice(420, size=12, replace=False)] = rng.integers(0, 20, size=12)
    c = synth.corpus(606, N=420, T=T, F=45, E=8, M=8, L=[28, 29, 30, 32, 32, 33, 35, 36])
    assert int(c["offsets"][-1]) >= 60000
    _check(tsr_hip, oracle, c)


@pytest.mark.parametrize("shape", [
    dict(N=269, F=6, E=5, M=8, L=153),      # one chunk array per parity: its rows are capped by what one staging pass covers
    dict(N=219, F=26, E=5, M=2, L=58),      # five chunk arrays, two staged pieces per row
    dict(N=150, F=9, E=16, M=3, L=200),     # D = 144, long filters: tiles closed early where 16 pairs would outgrow the image
], ids=["D30", "D130", "D144"])
def test_mx_many_short_utterances_on_narrow_rows(tsr_hip, oracle, shape):
    """Utterances shorter than the filters still occupy rows of the LDS image (one pair each, for the tail rule), so that a
    workgroup's frames can span thousands of rows when the rows are narrow: every one of them has to be staged (found by
    tools/fuzz_mx.py --large: arrays longer than five pieces per thread cover were left partly unfilled).  Where 16 pairs of
    such utterances would span more rows than the image holds, the plan closes the tile early (records that score nothing)."""
    from tsr_hip import synth
    rng = np.random.default_rng(shape["N"])
    Ts = [int(x) for x in rng.integers(0, 420, size=shape["N"])]
    c = synth.corpus(55, N=shape["N"], T=Ts, F=shape["F"], E=shape["E"], M=shape["M"], L=shape["L"])
    _check(tsr_hip, oracle, c)


def test_mx_sums_beyond_2_24_repeat_bit_for_bit(tsr_hip, oracle):
    """128-tap filters on 512 features (one phase): the fixed-point sums pass 2^24, where the f32 accumulator rounds and the
    order of the additions shows.  The bank is then scored with its K-blocks in fixed shares: three launches return the same
    bits, the all-scores pass has the fused pass' maximum, and the result stays within the bar of the oracle (and within
    2^-18 relative of the exact-integer emulation)."""
    from tsr_hip import synth
    rng = np.random.default_rng(9)
    Ts = [int(x) for x in rng.integers(150, 420, size=24)]
    c = synth.corpus(56, N=24, T=Ts, F=64, E=8, M=3, L=128)
    # weights of one sign and similar size: the sums grow with the number of set taps (~ -4 * 10^7 here)
    c["bank"] = [((-(1.0 + 0.5 * rng.random((128, 64, 8))) * (1.0 + 0.1 * m)).astype(np.float32), float(np.float32(3000.0 + 7 * m))) for m in range(3)]
    P = int(c["offsets"][-1])
    f2 = c["feats"].reshape(P, -1)
    bank = tsr_hip.Bank(c["bank"])
    dc = tsr_hip.Corpus(f2, c["offsets"])
    runs = []
    for _ in range(3):
        dc.score(bank, tsr_hip.TSR_FAST | tsr_hip.TSR_MX)
        assert tsr_hip.default_ctx().get_option(tsr_hip.TSR_INFO_LAST_BANK_KERNEL) == tsr_hip.TSR_KERNEL_MX
        runs.append(dc.download())
    for s_, a_ in runs[1:]:
        assert np.array_equal(s_, runs[0][0]) and np.array_equal(a_, runs[0][1])
    s = runs[0][0]
    allm = dc.score_all(bank, tsr_hip.TSR_FAST | tsr_hip.TSR_MX)
    assert np.array_equal(allm.max(0), s)
    es, ea, vals, slack = _emulate(f2, c["offsets"], c["bank"])
    raw = max(float(np.abs(vals[m] - np.float32(cm)).max()) * 4095 / float(np.abs(W).max()) for m, (W, cm) in enumerate(c["bank"]))
    assert raw > 2 ** 24   # (the case is what it says: integer sums beyond the f32 mantissa)
    assert np.all(np.abs(s - es) <= 16 * slack + 1e-30)
    so, ao = oracle.score_corpus(c["feats"], c["offsets"], c["bank"])
    assert rel_err(s, so) <= TOL
    dc.close()
    bank.close()


@pytest.mark.parametrize("shape", [
    dict(N=6, T=[200, 90, 33, 140, 64, 31], F=45, E=8, M=9, L=32),                                  # nine templates: a slice of eight and a slice of one
    dict(N=5, T=[120, 61, 200, 47, 90], F=45, E=8, M=20, L=[20 + (3 * i) % 17 for i in range(20)]),   # three slices, ragged filters
    dict(N=30, T=200, F=45, E=8, M=64, L=32),                                                       # config 4's bank on a small corpus: eight slices
], ids=["M9", "M20_ragged", "M64"])
def test_mx_banks_of_more_than_eight_templates(tsr_hip, oracle, shape):
    """More than eight templates go through the kernel in slices of eight, one launch each; the later launches keep the
    better of what is there and what they found (ties: the lower id, i.e. the earlier slice).  Against the emulation and
    the oracle, ids included; the all-scores pass writes every template's row."""
    from tsr_hip import synth
    c = synth.corpus(91, **shape)
    s, a = _check(tsr_hip, oracle, c)
    P = int(c["offsets"][-1])
    f2 = c["feats"].reshape(P, -1)
    bank = tsr_hip.Bank(c["bank"])
    dc = tsr_hip.Corpus(f2, c["offsets"])
    allm = dc.score_all(bank, tsr_hip.TSR_FAST | tsr_hip.TSR_MX)
    assert tsr_hip.default_ctx().get_option(tsr_hip.TSR_INFO_LAST_BANK_KERNEL) == tsr_hip.TSR_KERNEL_MX
    es, ea, vals, slack = _emulate(f2, c["offsets"], c["bank"])
    assert allm.shape == vals.shape and np.all(np.abs(allm - vals) <= slack[None, :] + 1e-30)
    assert np.array_equal(allm.max(0), s)
    assert np.array_equal(a, np.argmax(allm, axis=0))           # (first maximum = lowest id, across the slices too)
    dc.close()
    bank.close()


def test_mx_ties_across_slices_go_to_the_lowest_id(tsr_hip, oracle):
    """Templates 1 and 9 are the same filter: wherever they are the best, the id is 1 (the later slice replaces a score only
    where it is strictly better)."""
    from tsr_hip import synth
    c = synth.corpus(92, N=4, T=[120, 64, 200, 90], F=45, E=8, M=12, L=32)
    bk = list(c["bank"])
    bk[9] = bk[1]
    c["bank"] = tuple(bk)
    s, a = _check(tsr_hip, oracle, c)
    assert not np.any(a == 9) and np.any(a == 1)


def test_mx_long_utterance(tsr_hip, oracle):
    from tsr_hip import synth
    c = synth.corpus(70, N=2, T=[30001, 777], F=45, E=8, M=6, L=[32, 41, 36, 33, 40, 38])
    _check(tsr_hip, oracle, c)


def test_mx_requests_outside_its_envelope_are_served_as_fast(tsr_hip, oracle):
    """Short filters (fewer than 24 K-blocks) fall to TSR_FAST's int8 kernels: same results as plain TSR_FAST, bit for bit."""
    from tsr_hip import synth
    for shape in (dict(N=3, T=[40, 50, 30], F=6, E=5, M=3, L=9), dict(N=2, T=[70, 90], F=16, E=8, M=9, L=20)):
        c = synth.corpus(31, **shape)
        P = int(c["offsets"][-1])
        f2 = c["feats"].reshape(P, -1)
        bank = tsr_hip.Bank(c["bank"])
        s, a = bank.score(f2, c["offsets"], flags=tsr_hip.TSR_FAST | tsr_hip.TSR_MX)
        assert tsr_hip.default_ctx().get_option(tsr_hip.TSR_INFO_LAST_BANK_KERNEL) != tsr_hip.TSR_KERNEL_MX
        sf, af = bank.score(f2, c["offsets"], flags=tsr_hip.TSR_FAST)
        assert np.array_equal(s, sf) and np.array_equal(a, af)
        bank.close()
    c = synth.corpus(32, N=3, T=[90, 140, 60], F=45, E=8, M=8, L=32)
    P = int(c["offsets"][-1])
    bank = tsr_hip.Bank(c["bank"])
    dc = tsr_hip.Corpus(c["feats"].reshape(P, -1), c["offsets"])
    with pytest.raises(ValueError):
        dc.score(bank, tsr_hip.TSR_FAST | tsr_hip.TSR_MX | tsr_hip.TSR_W24)
    dc.close()
    bank.close()


@pytest.mark.parametrize("shape", [
    dict(N=7, T=[200, 199, 33, 32, 31, 5, 64], F=45, E=8, M=8, L=32),
    dict(N=5, T=[90, 0, 61, 1, 47], F=45, E=8, M=5, L=[20, 24, 28, 32, 36]),
    dict(N=40, T=70, F=45, E=8, M=7, L=[32, 32, 33, 34, 35, 36, 37]),
], ids=["odd_T", "ragged_L", "short_utts"])
def test_mx_all_scores_pass(tsr_hip, oracle, shape):
    """The all-scores pass of the MX kernel: every template's score at every frame (tail frames: 0 + c_m) against the
    emulation, its maximum equal to the fused pass bit for bit, and within the bar of TSR_EXACT's all-scores pass."""
    from tsr_hip import synth
    c = synth.corpus(78, **shape)
    P = int(c["offsets"][-1])
    f2 = c["feats"].reshape(P, -1)
    bank = tsr_hip.Bank(c["bank"])
    dc = tsr_hip.Corpus(f2, c["offsets"])
    allm = dc.score_all(bank, tsr_hip.TSR_FAST | tsr_hip.TSR_MX)
    assert tsr_hip.default_ctx().get_option(tsr_hip.TSR_INFO_LAST_BANK_KERNEL) == tsr_hip.TSR_KERNEL_MX
    dc.score(bank, tsr_hip.TSR_FAST | tsr_hip.TSR_MX)
    s, a = dc.download()
    es, ea, vals, slack = _emulate(f2, c["offsets"], c["bank"])
    assert allm.shape == vals.shape
    assert np.all(np.abs(allm - vals) <= slack[None, :] + 1e-30)
    assert np.array_equal(allm.max(0), s)
    alle = dc.score_all(bank, tsr_hip.TSR_EXACT)
    assert rel_err(allm, alle) <= TOL
    dc.close()
    bank.close()


@pytest.mark.parametrize("case", [
    dict(K=50, F=25, Ts=[60] * 30 + [34, 200, 32, 46], thr=12, spread=(2, 2), L=[20, 24, 28, 32, 32, 30, 26, 22]),   # D = 1050: three phases of 11 dwords
    dict(K=50, F=45, Ts=[200] * 12, thr=12, spread=(2, 2), L=[32] * 8),                                              # config 3's rows: D = 2050, six phases
    dict(K=30, F=28, Ts=[64, 40, 90, 38], thr=10, spread=(1, 1), L=[32, 36, 33]),                                    # D = 720: two phases of 12 dwords
], ids=["D1050", "D2050", "D720"])
def test_mx_wide_rows_in_feature_phases(tsr_hip, oracle, case):
    """Part corpora (rows wider than 512 features, resident as bit rows): the MX kernel goes through the features in the
    phases of the corpus' own bit form with one accumulator set.  Against the integer emulation, TSR_FAST and the oracle;
    the all-scores pass too (its maximum equal to the fused pass bit for bit: sums beyond 2^24 round in f32, and the kernel
    adds the K-blocks of wide rows in a fixed order so that every launch rounds alike).  A corpus with an utterance of odd
    length, and the same frames as byte rows, go through the kernel's own phase-major copy."""
    from tsr_hip import synth
    ctx = tsr_hip.default_ctx()
    rng = np.random.default_rng(case["K"] + case["F"])
    K, F, Ts = case["K"], case["F"], case["Ts"]
    edges = [synth.features(rng, T, rng.uniform(0.05, 0.45, size=(F, 8))) for T in Ts]
    offs = np.concatenate(([0], np.cumsum(Ts))).astype(np.int64)
    lp, lip = synth.parts(rng, K)
    corpus = tsr_hip.Corpus(np.concatenate(edges).reshape(int(offs[-1]), -1), offs)
    parts = tsr_hip.Parts(lp, lip)
    pc = corpus.part_features(parts, case["thr"], spread=case["spread"])
    assert pc.layout()["bits"] and pc.D > 512
    bank_cs = synth.bank(rng, synth.background(rng, (pc.D // K, K)), case["L"])
    bank = tsr_hip.Bank(bank_cs)
    pc.score(bank, tsr_hip.TSR_FAST | tsr_hip.TSR_MX)
    assert ctx.get_option(tsr_hip.TSR_INFO_LAST_BANK_KERNEL) == tsr_hip.TSR_KERNEL_MX
    s, a = pc.download()
    allm = pc.score_all(bank, tsr_hip.TSR_FAST | tsr_hip.TSR_MX)
    assert ctx.get_option(tsr_hip.TSR_INFO_LAST_BANK_KERNEL) == tsr_hip.TSR_KERNEL_MX
    for _ in range(3):
        pc.score(bank, tsr_hip.TSR_FAST | tsr_hip.TSR_MX)
        s_again, a_again = pc.download()
        assert np.array_equal(s, s_again) and np.array_equal(a, a_again)
    pc.score(bank, tsr_hip.TSR_FAST)
    sf, af = pc.download()
    feats = pc.download_feats()
    po = pc.offsets
    es, ea, vals, slack = _emulate(feats, po, bank_cs)
    assert np.all(np.abs(s - es) <= slack + 1e-30), float(np.max(np.abs(s - es) - slack))
    assert np.all(np.abs(allm - vals) <= slack[None, :] + 1e-30)
    assert np.array_equal(allm.max(0), s)
    top2 = np.sort(vals, axis=0)[-2:]
    clear = (top2[1] - top2[0]) > 4 * slack
    assert np.array_equal(a[clear], ea[clear])
    so, ao = oracle.score_corpus(feats, po, bank_cs)
    assert rel_err(s, so) <= TOL and rel_err(sf, so) <= TOL
    pc.close()
    corpus.close()
    # an utterance of odd length: the kernel reads its own phase-major copy of the bit form, every utterance on an even row
    Ts2 = [Ts[0] + 1] + list(Ts[1:3]) + [Ts[0] - 1]
    offs2 = np.concatenate(([0], np.cumsum(Ts2))).astype(np.int64)
    e2 = [synth.features(rng, T, rng.uniform(0.05, 0.45, size=(F, 8))) for T in Ts2]
    corpus2 = tsr_hip.Corpus(np.concatenate(e2).reshape(int(offs2[-1]), -1), offs2)
    pc2 = corpus2.part_features(parts, case["thr"], spread=case["spread"])
    assert any(int(t) & 1 for t in np.diff(pc2.offsets))
    pc2.score(bank, tsr_hip.TSR_FAST | tsr_hip.TSR_MX)
    assert ctx.get_option(tsr_hip.TSR_INFO_LAST_BANK_KERNEL) == tsr_hip.TSR_KERNEL_MX
    s2, a2 = pc2.download()
    feats2 = pc2.download_feats()
    es2, ea2, vals2, slack2 = _emulate(feats2, pc2.offsets, bank_cs)
    assert np.all(np.abs(s2 - es2) <= slack2 + 1e-30)
    so2, ao2 = oracle.score_corpus(feats2, pc2.offsets, bank_cs)
    assert rel_err(s2, so2) <= TOL
    # the same frames handed over as byte rows (a host corpus): packed into the phase-major form from the bytes
    dc = tsr_hip.Corpus(feats2, pc2.offsets)
    dc.score(bank, tsr_hip.TSR_FAST | tsr_hip.TSR_MX)
    assert ctx.get_option(tsr_hip.TSR_INFO_LAST_BANK_KERNEL) == tsr_hip.TSR_KERNEL_MX
    s3, a3 = dc.download()
    assert np.array_equal(s3, s2) and np.array_equal(a3, a2)
    dc.close()
    pc2.close()
    corpus2.close()
    bank.close()
