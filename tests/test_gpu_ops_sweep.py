"""Seeded random sweep of the per-operator C-ABI entry points against the oracle: shapes the model path never uses
(channel counts that are not multiples of the 32-wide K step or the 16-wide N tile, sequences shorter than the dilated
receptive field, single rows, every window position near the end of the text).  Bit-exact for the integer outputs
(embedding gather, attention argmax), 2e-5 max-abs for LayerNorm-normalised outputs."""
import numpy as np
import pytest

from oracle import ophelia_oracle as O

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _ln(rng, prefix, C):
    return {prefix + "/gamma": (1 + 0.1 * rng.standard_normal(C)).astype(np.float32),
            prefix + "/beta": (0.1 * rng.standard_normal(C)).astype(np.float32)}


def test_conv1d_random_shapes():
    from ophelia_amd import modules as M
    rng = np.random.default_rng(2024)
    for case in range(40):
        B, T = int(rng.integers(1, 5)), int(rng.integers(1, 60))
        Cin, Cout = int(rng.integers(1, 400)), int(rng.integers(1, 400))
        size = int(rng.choice([1, 3]))
        rate = int(rng.choice([1, 2, 3, 9, 27]))
        padding = str(rng.choice(["SAME", "CAUSAL"]))
        act = rng.choice(["relu", "sigmoid", None])
        x = rng.standard_normal((B, T, Cin)).astype(np.float32)
        W = {"c/conv1d/kernel": (rng.standard_normal((size, Cin, Cout)) * (2.6 / (size * Cin)) ** 0.5).astype(np.float32),
             "c/conv1d/bias": (0.02 * rng.standard_normal(Cout)).astype(np.float32)}
        W.update(_ln(rng, "c/normalize", Cout))
        ref = O.conv1d(x, W, "c", rate=rate, padding=padding, activation_fn={"relu": O.relu, "sigmoid": O.sigmoid, None: None}[act])
        got = M.conv1d(x, W, "c", size=size, rate=rate, padding=padding, activation_fn=act)
        tol = TOL if Cout > 1 else 1e-3          # a 1-channel LayerNorm divides by sqrt(0 + 1e-12)
        assert got.shape == ref.shape and np.abs(got - ref).max() < tol, (case, B, T, Cin, Cout, size, rate, padding, act)


def test_hc_random_shapes():
    from ophelia_amd import modules as M
    rng = np.random.default_rng(2025)
    for case in range(30):
        B, T = int(rng.integers(1, 4)), int(rng.integers(1, 70))
        C = int(rng.integers(1, 130)) * 4                      # hc needs Cout == Cin; kernels want C % 4 == 0
        size = int(rng.choice([1, 3]))
        rate = int(rng.choice([1, 3, 9, 27]))
        padding = str(rng.choice(["SAME", "CAUSAL"]))
        x = rng.standard_normal((B, T, C)).astype(np.float32)
        W = {"h/conv1d/kernel": (rng.standard_normal((size, C, 2 * C)) * (2.6 / (size * C)) ** 0.5).astype(np.float32),
             "h/conv1d/bias": (0.02 * rng.standard_normal(2 * C)).astype(np.float32)}
        W.update(_ln(rng, "h/H1", C)); W.update(_ln(rng, "h/H2", C))
        ref = O.hc(x, W, "h", rate=rate, padding=padding)
        got = M.hc(x, W, "h", size=size, rate=rate, padding=padding)
        assert np.abs(got - ref).max() < TOL, (case, B, T, C, size, rate, padding)


def test_conv1d_transpose_random_shapes():
    from ophelia_amd import modules as M
    rng = np.random.default_rng(2026)
    for case in range(20):
        B, T = int(rng.integers(1, 4)), int(rng.integers(1, 40))
        Cin, Cout = int(rng.integers(1, 300)), int(rng.integers(2, 300))
        x = rng.standard_normal((B, T, Cin)).astype(np.float32)
        W = {"d/conv2d_transpose/kernel": (rng.standard_normal((1, 3, Cout, Cin)) * (2.6 / (3 * Cin)) ** 0.5).astype(np.float32),
             "d/conv2d_transpose/bias": (0.02 * rng.standard_normal(Cout)).astype(np.float32)}
        W.update(_ln(rng, "d/normalize", Cout))
        ref = O.conv1d_transpose(x, W, "d")
        got = M.conv1d_transpose(x, W, "d")
        assert got.shape == ref.shape == (B, 2 * T, Cout) and np.abs(got - ref).max() < TOL, (case, B, T, Cin, Cout)


def test_layernorm_and_embed_random_shapes():
    from ophelia_amd import modules as M
    rng = np.random.default_rng(2027)
    for case in range(25):
        C = int(rng.integers(2, 1281))
        x = (rng.standard_normal((int(rng.integers(1, 4)), int(rng.integers(1, 30)), C)) * rng.uniform(0.01, 30)).astype(np.float32)
        W = _ln(rng, "n", C)
        assert np.abs(M.normalize(x, W, "n") - O.normalize(x, W["n/gamma"], W["n/beta"])).max() < TOL, (case, C)
    for case in range(10):
        V, U = int(rng.integers(2, 300)), int(rng.integers(1, 200))
        tab = rng.standard_normal((V, U)).astype(np.float32)
        ids = rng.integers(0, V, size=(int(rng.integers(1, 5)), int(rng.integers(1, 50)))).astype(np.int32)
        assert np.array_equal(M.embed(ids, tab), O.embed(ids, tab)), (case, V, U)


def test_attention_every_window_position():
    from ophelia_amd import modules as M
    rng = np.random.default_rng(2028)
    for N, d, win in ((20, 64, 3), (37, 128, 1), (150, 256, 3), (64, 256, 8), (9, 32, 5)):
        class hp: pass
        hp.d, hp.max_N, hp.attention_win_size, hp.concatenate_query = d, N, win, True
        B, T = N, 3                                           # one utterance per possible prev_max, 0 .. N-1
        Q = rng.standard_normal((B, T, d)).astype(np.float32)
        K = rng.standard_normal((B, N, d)).astype(np.float32)
        V = rng.standard_normal((B, N, d)).astype(np.float32)
        p = np.arange(N, dtype=np.int32)
        R, al, mx = M.attention(hp, Q, K, V, p)
        R0, al0, mx0 = O.attention(hp, Q, K, V, p)
        assert np.array_equal(mx, mx0), (N, d, win)
        assert np.abs(al - al0).max() < 1e-6 and np.abs(R - R0).max() < 2e-5, (N, d, win)
