"""GPU parity: each HIP operator (through the per-op C-ABI entry points) against the
oracle on the same seeded inputs.  Tolerances: fp32 throughout; 2e-5 max-abs on
LayerNorm-normalised outputs (fp reassociation between MFMA k-order and BLAS)."""
import numpy as np
import pytest

from oracle import ophelia_oracle as O

pytestmark = pytest.mark.gpu
TOL = 2e-5
rng = np.random.default_rng(123)


def _r(*s, sc=1.0):
    return (rng.standard_normal(s) * sc).astype(np.float32)


def _ln(prefix, C):
    return {prefix + "/gamma": (1 + _r(C, sc=0.1)), prefix + "/beta": _r(C, sc=0.1)}


def test_embed_rows():
    from ophelia_amd import modules as M
    tab = _r(57, 128, sc=0.1)
    ids = rng.integers(0, 57, size=(4, 33)).astype(np.int32)
    ids[:, -5:] = 0
    assert np.array_equal(M.embed(ids, tab), O.embed(ids, tab))


@pytest.mark.parametrize("C", [80, 256, 512, 1024, 1025])
def test_layernorm(C):
    from ophelia_amd import modules as M
    x = _r(3, 50, C, sc=3.0)
    W = _ln("n", C)
    assert np.abs(M.normalize(x, W, "n") - O.normalize(x, W["n/gamma"], W["n/beta"])).max() < TOL


@pytest.mark.parametrize("Cin,Cout,act", [(80, 256, "relu"), (128, 512, "relu"), (512, 256, None), (256, 80, None),
                                          (1024, 1025, None), (1025, 1025, "relu"), (384, 256, None)])
def test_conv1d_k1(Cin, Cout, act):
    from ophelia_amd import modules as M
    B, T = 3, 45           # ragged vs the 64/128-row tiles
    x = _r(B, T, Cin)
    W = {"c/conv1d/kernel": _r(1, Cin, Cout, sc=(2.6 / Cin) ** 0.5), "c/conv1d/bias": _r(Cout, sc=0.02)}
    W.update(_ln("c/normalize", Cout))
    ref = O.conv1d(x, W, "c", activation_fn={"relu": O.relu, None: None}[act])
    got = M.conv1d(x, W, "c", activation_fn=act)
    assert np.abs(got - ref).max() < TOL


@pytest.mark.parametrize("C,size,rate,padding", [(256, 3, 1, "CAUSAL"), (256, 3, 27, "CAUSAL"), (512, 3, 9, "SAME"),
                                                 (512, 1, 1, "SAME"), (1024, 3, 1, "SAME"), (512, 3, 3, "SAME")])
def test_hc(C, size, rate, padding):
    from ophelia_amd import modules as M
    B, T = 2, 70
    x = _r(B, T, C)
    W = {"h/conv1d/kernel": _r(size, C, 2 * C, sc=(2.6 / (size * C)) ** 0.5), "h/conv1d/bias": _r(2 * C, sc=0.02)}
    W.update(_ln("h/H1", C)); W.update(_ln("h/H2", C))
    ref = O.hc(x, W, "h", rate=rate, padding=padding)
    got = M.hc(x, W, "h", size=size, rate=rate, padding=padding)
    assert np.abs(got - ref).max() < TOL


def test_hc_large_m_uses_128_tiles():
    from ophelia_amd import modules as M
    B, T, C = 16, 800, 512      # 12800 rows x 1024 cols -> 128x128 tile path
    x = _r(B, T, C)
    W = {"h/conv1d/kernel": _r(3, C, 2 * C, sc=(2.6 / (3 * C)) ** 0.5), "h/conv1d/bias": _r(2 * C, sc=0.02)}
    W.update(_ln("h/H1", C)); W.update(_ln("h/H2", C))
    assert np.abs(M.hc(x, W, "h", size=3, rate=3) - O.hc(x, W, "h", rate=3)).max() < TOL


@pytest.mark.parametrize("B,T,C", [(2, 25, 512), (1, 1, 64), (16, 200, 512)])
def test_conv1d_transpose(B, T, C):
    from ophelia_amd import modules as M
    x = _r(B, T, C)
    W = {"d/conv2d_transpose/kernel": _r(1, 3, C, C, sc=(2.6 / (3 * C)) ** 0.5), "d/conv2d_transpose/bias": _r(C, sc=0.02)}
    W.update(_ln("d/normalize", C))
    ref = O.conv1d_transpose(x, W, "d")
    got = M.conv1d_transpose(x, W, "d")
    assert got.shape == (B, 2 * T, C)
    assert np.abs(got - ref).max() < TOL


@pytest.mark.parametrize("B,T,C,prec,tol", [(2, 37, 512, 2, 2e-5), (16, 200, 512, 2, 2e-5), (1, 1, 512, 2, 2e-5), (3, 50, 512, 1, 3e-4),
                                            (2, 33, 256, 2, 2e-5), (16, 400, 512, 2, 2e-5), (11, 397, 512, 2, 2e-5)])
def test_conv1d_transpose_split_precisions(B, T, C, prec, tol):
    """The launches the SSRN path makes for D_4 / D_7 at the split precisions (2: the input as fp16 planes, both phases as one
    plane_gemm problem -- its 4-wave form, and from 33 row tiles on, (16, 400) and the ragged (11, 397), the 8-wave form with 128
    channels per workgroup; 1: both phases in one launch of conv_gemm_bf16x3_pair; then ln_rows; ragged last tile, T = 1) against
    the oracle (modules.py:209-258)."""
    from ophelia_amd import modules as M
    x = _r(B, T, C)
    W = {"d/conv2d_transpose/kernel": _r(1, 3, C, C, sc=(2.6 / (3 * C)) ** 0.5), "d/conv2d_transpose/bias": _r(C, sc=0.02)}
    W.update(_ln("d/normalize", C))
    ref = O.conv1d_transpose(x, W, "d")
    got = M.conv1d_transpose(x, W, "d", precision=prec)
    assert got.shape == (B, 2 * T, C)
    assert np.abs(got - ref).max() < tol
    # size-independent: an impulse at frame t0 touches output rows 2 t0 .. 2 t0 + 2 only (the tile and phase bookkeeping)
    if T > 8:
        x1 = x.copy(); x1[0, 5] += _r(C)
        changed = np.where(np.abs(M.conv1d_transpose(x1, W, "d", precision=prec) - got).max(axis=(0, 2)) > 0)[0]
        assert changed.tolist() == [10, 11, 12]


def test_conv1d_transpose_linearity_and_shift():
    """size-independent properties at full SSRN size: out[2t+1] depends on x[t] only; a one-frame
    impulse touches exactly output rows 2t, 2t+1, 2t+2 (before LayerNorm: use gamma=1,beta=0 rows)."""
    from ophelia_amd import modules as M
    B, T, C = 1, 200, 512
    W = {"d/conv2d_transpose/kernel": _r(1, 3, C, C, sc=0.05), "d/conv2d_transpose/bias": np.zeros(C, np.float32),
         "d/normalize/gamma": np.ones(C, np.float32), "d/normalize/beta": np.zeros(C, np.float32)}
    x0 = _r(B, T, C)
    x1 = x0.copy(); x1[0, 77] += _r(C)
    y0, y1 = M.conv1d_transpose(x0, W, "d"), M.conv1d_transpose(x1, W, "d")
    changed = np.where(np.abs(y1 - y0).max(axis=(0, 2)) > 0)[0]
    assert changed.tolist() == [154, 155, 156]


def test_attention_rows():
    from ophelia_amd import modules as M
    class hp: d = 256; max_N = 150; attention_win_size = 3; concatenate_query = True
    B, T = 4, 9
    Q, K, V = _r(B, T, 256), _r(B, 150, 256), _r(B, 150, 256)
    p = np.array([0, 57, 147, 149], np.int32)       # incl. the max_N-win edge cases
    R, al, mx = M.attention(hp, Q, K, V, p)
    R0, al0, mx0 = O.attention(hp, Q, K, V, p)
    assert np.array_equal(mx, mx0)
    assert np.abs(al - al0).max() < 1e-6 and np.abs(R - R0).max() < 1e-5
    assert np.all(al[1, :57] == 0) and np.all(al[1, 60:] == 0)
