"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every
symbol include/tsr.h declares, refuses to compute without a GPU (no CPU fallback), and
the Python host layer validates arguments like the reference's compiled modules do."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "tsr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tsr_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built_lib():
    from tsr_hip import build, _lib
    build.build()
    return _lib.lib()


def test_header_symbols_are_exported(built_lib):
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(built_lib, n), "libtsr_hip.so does not export %s" % n


def test_python_signatures_cover_the_header(built_lib):
    from tsr_hip import _lib
    bound = set(_lib.SIGNATURES) | set(_lib._STR_FUNCS)
    assert set(_declared()) == bound


def test_version_and_strerror(built_lib):
    assert built_lib.tsr_version() == 100
    assert built_lib.tsr_strerror(0) == b"ok"
    assert b"invalid" in built_lib.tsr_strerror(1)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present")
def test_no_cpu_fallback_without_gpu(built_lib):
    """The product path must fail loudly when there is no device."""
    import template_speech_rec.compute_likelihood_linear_filter as cllf
    from tsr_hip import TsrError
    with pytest.raises(TsrError):
        cllf.detect(np.zeros((4, 2, 2), np.uint8), np.zeros((2, 2, 2), np.float32))


def test_product_path_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "template-speech-recognition_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower(), "%s mentions the oracle" % f


def test_buffer_checks_mirror_cython():
    import template_speech_rec.compute_likelihood_linear_filter as cllf
    import template_speech_rec.code_parts as cp
    import template_speech_rec.spread_waliji_patches as swp
    f32 = np.zeros((2, 2, 2), np.float32)
    with pytest.raises(ValueError, match="Buffer dtype mismatch, expected 'UINT_t' but got 'double'"):
        cllf.detect(np.zeros((4, 2, 2), np.float64), f32)
    with pytest.raises(ValueError, match="Buffer dtype mismatch, expected 'DTYPE_t' but got 'double'"):
        cllf.detect(np.zeros((4, 2, 2), np.uint8), f32.astype(np.float64))
    with pytest.raises(ValueError, match="wrong number of dimensions"):
        cllf.detect(np.zeros((4, 4), np.uint8), f32)
    with pytest.raises(TypeError):
        cllf.detect([[0]], f32)
    with pytest.raises(ValueError, match="longer than the utterance"):
        cllf.detect(np.zeros((1, 2, 2), np.uint8), f32)
    with pytest.raises(ValueError, match="expected 'int64_t' but got 'int'"):
        swp.spread_waliji_patches(np.zeros((3, 3), np.int32), 1, 1, 2)
    with pytest.raises(ValueError, match="expected 'DTYPE_t' but got 'float'"):
        cp.code_parts(np.zeros((6, 6, 8), np.uint8), np.zeros((2, 5, 5, 8), np.float32),
                      np.zeros((2, 5, 5, 8), np.float32), 1)
    with pytest.raises(ValueError, match="expected 'float64_t' but got 'float'"):  # dead in the reference
        cllf.detect_examples(np.zeros((1, 4, 2, 2), np.uint8), np.zeros((2, 2, 2)), np.zeros(1, np.int64), 4)


def test_construct_linear_filter_matches_oracle(oracle):
    from template_speech_rec import estimate_template as et
    rng = np.random.default_rng(5)
    T = rng.uniform(0, 1, size=(7, 4, 3))
    bgd = rng.uniform(0.05, 0.4, size=(4, 3))
    W, c = et.construct_linear_filter(T, bgd)
    Wo, co = oracle.construct_linear_filter(T, bgd)
    assert W.dtype == np.float32 and np.array_equal(W, Wo) and c == co
    cs = et.construct_linear_filter(T, bgd, all_cs=True)[1]
    assert np.allclose(cs, oracle.construct_linear_filter(T, bgd, all_cs=True)[1], rtol=0, atol=0)
    pairs = et.construct_linear_filters([T, T[:3]], bgd)
    po = oracle.construct_linear_filters([T, T[:3]], bgd)
    for (a, b), (ao, bo) in zip(pairs, po):
        assert np.array_equal(a, ao) and b == bo


def test_glue_layout_helpers(oracle):
    from template_speech_rec import get_train_data as gtrd
    rng = np.random.default_rng(6)
    part = rng.integers(0, 2, size=(8 * 5, 11))
    assert np.array_equal(gtrd.reorg_part_for_fast_filtering(part), oracle.reorg_part_for_fast_filtering(part))
    A = rng.integers(0, 9, size=(13, 7))
    assert np.array_equal(gtrd.collapse_to_grid(A, 5, 3), oracle.collapse_to_grid(A, 5, 3))


def test_synth_corpus_shapes():
    from tsr_hip import synth
    c = synth.corpus(2, N=3, T=[10, 12, 9], F=4, E=2, M=3, L=[3, 5, 4])
    assert c["feats"].shape == (31, 4, 2) and c["offsets"].tolist() == [0, 10, 22, 31]
    assert [w.shape[0] for w, _ in c["bank"]] == [3, 5, 4]
    assert set(np.unique(c["feats"])) <= {0, 1}


def test_isolated_virtual_windows_describe_the_reference_windows(oracle):
    """Host logic of the isolated classification: the (first frame, real range, row count, searched
    positions) description handed to tsr_corpus_classify_isolated must reproduce, row for row, the
    background-padded windows get_isolated_classify_windows builds (get_train_data.py:781-835)."""
    import template_speech_rec.get_train_data as gtrd
    from tsr_hip import synth
    rng = np.random.default_rng(8)
    for trial in range(30):
        F, E = int(rng.integers(1, 6)), int(rng.choice([1, 8]))
        q = synth.background(rng, (F, E)).astype(np.float32)
        Ls = [int(x) for x in rng.integers(1, 40, size=int(rng.integers(1, 5)))]
        bank = synth.bank(rng, q, Ls)
        T = int(rng.integers(8, 140))
        Em = synth.features(rng, T, q)
        cuts = np.sort(rng.choice(np.arange(2, T - 1), size=min(int(rng.integers(1, 8)), T - 3), replace=False))
        flts = np.concatenate(([0], cuts, [T])).astype(np.int64)
        wins, ws, we, pad0 = oracle.get_isolated_classify_windows(Em, None, flts.copy(), q, bank)
        frame0 = int(rng.integers(0, 1000))
        first, lo, hi, rows, n_valid, ws2, pad2, _ = gtrd._isolated_virtual_windows(T, flts, max(Ls), frame0=frame0)
        assert np.array_equal(ws2, ws.astype(np.int64)) and pad2 == int(pad0)
        assert np.array_equal(n_valid, we.astype(np.int64) - ws.astype(np.int64))
        for w, win in enumerate(wins):
            assert rows[w] == len(win)
            for r in range(len(win)):
                f = first[w] + r
                want = Em[f - frame0].astype(np.float32) if lo[w] <= f < hi[w] else q
                assert np.array_equal(win[r], want), (trial, w, r)
