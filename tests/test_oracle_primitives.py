"""Cross-checks each oracle primitive against independent torch CPU ops. CPU only."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import ophelia_oracle as O

rng = np.random.default_rng(0)


def _w(*shape, s=0.1):
    return (rng.standard_normal(shape) * s).astype(np.float32)


def test_layernorm_eps_and_biased_variance():
    x = _w(3, 7, 80, s=2.0); g = 1 + _w(80); b = _w(80)
    ref = F.layer_norm(torch.from_numpy(x), (80,), torch.from_numpy(g), torch.from_numpy(b), eps=1e-12).numpy()
    assert np.abs(O.normalize(x, g, b) - ref).max() < 2e-6


def _conv_case(size, rate, padding):
    B, T, Cin, Cout = 2, 37, 24, 40
    x = _w(B, T, Cin, s=1.0); k = _w(size, Cin, Cout); bias = _w(Cout)
    mine = O._conv_taps(x, k, bias, rate, padding)
    xt = torch.from_numpy(x).transpose(1, 2)
    w = torch.from_numpy(np.ascontiguousarray(k.transpose(2, 1, 0)))
    if padding == "CAUSAL":
        xt = F.pad(xt, ((size - 1) * rate, 0))
        ref = F.conv1d(xt, w, torch.from_numpy(bias), dilation=rate)
    else:
        ref = F.conv1d(xt, w, torch.from_numpy(bias), dilation=rate, padding="same")
    assert np.abs(mine - ref.transpose(1, 2).numpy()).max() < 1e-5


def test_conv_same_and_causal_dilated():
    for size, rate in [(1, 1), (3, 1), (3, 3), (3, 9), (3, 27)]:
        for padding in ("SAME", "CAUSAL"):
            _conv_case(size, rate, padding)


def test_conv1d_transpose_same_stride2():
    B, T, C = 2, 11, 16
    x = _w(B, T, C, s=1.0)
    W = {"D/conv2d_transpose/kernel": _w(1, 3, C, C), "D/conv2d_transpose/bias": _w(C),
         "D/normalize/gamma": np.ones(C, np.float32), "D/normalize/beta": np.zeros(C, np.float32)}
    mine = O.conv1d_transpose(x, W, "D")
    w = torch.from_numpy(np.ascontiguousarray(W["D/conv2d_transpose/kernel"][0].transpose(2, 1, 0)))  # (Cin,Cout,k)
    y = F.conv_transpose1d(torch.from_numpy(x).transpose(1, 2), w, torch.from_numpy(W["D/conv2d_transpose/bias"]),
                           stride=2)[:, :, :2 * T].transpose(1, 2)
    ref = F.layer_norm(y, (C,), eps=1e-12).numpy()
    assert mine.shape == (B, 2 * T, C)
    assert np.abs(mine - ref).max() < 1e-5


def test_embed_zeroes_row0_at_lookup():
    tab = _w(9, 5)
    out = O.embed(np.array([[0, 3, 0, 8]]), tab)
    assert not out[0, 0].any() and not out[0, 2].any()
    assert np.array_equal(out[0, 1], tab[3]) and tab[0].any()


def test_attention_mask_window_and_tail():
    class hp: d = 16; max_N = 10; attention_win_size = 3; concatenate_query = True
    Q = _w(2, 4, 16, s=1.0); K = _w(2, 10, 16, s=1.0); V = _w(2, 10, 16, s=1.0)
    R, al, mx = O.attention(hp, Q, K, V, np.array([2, 8]))
    # utterance 0: only keys 2,3,4 alive; utterance 1 (p=8 > max_N-win): keys 8,9 alive
    assert np.all(al[0, :2] == 0) and np.all(al[0, 5:] == 0) and np.all(al[0, 2:5] > 0)
    assert np.all(al[1, :8] == 0) and np.all(al[1, 8:] > 0)
    assert np.allclose(al.sum(1), 1, atol=1e-6)
    assert R.shape == (2, 4, 32) and np.array_equal(R[..., 16:], Q)
    assert mx.min() >= 2
    # dense softmax over all keys agrees with torch on the unmasked logits
    A = torch.from_numpy(Q[0]) @ torch.from_numpy(K[0, 2:5]).T / 4.0
    assert np.abs(torch.softmax(A, -1).numpy() - al[0, 2:5].T).max() < 1e-6


def test_mel2mag_chunking_is_py2_integer_division():
    class hp: pass
    calls = []
    orig = O.ssrn
    try:
        O.ssrn = lambda hp_, Yb, W, speakers=None: (None, calls.append(len(Yb)) or np.zeros((len(Yb), 1, 1), np.float32))
        O.synth_mel2mag(hp, None, np.zeros((300, 1, 1), np.float32), batchsize=128)
    finally:
        O.ssrn = orig
    assert calls == [150, 150]          # max(1, 300//128) = 2 chunks via array_split


def test_conv1d_transpose_is_the_adjoint_of_tf_same_stride2_conv():
    """From the documented definition rather than from another framework's kernel: tf.layers.conv2d_transpose(kernel (1, 3),
    strides (1, 2), 'same') is the gradient (linear adjoint) of conv2d over a length-2T input with stride 2 and SAME padding, whose
    padding rule is out = ceil(in / stride), pad_total = max((out - 1) * stride + k - in, 0), pad_left = pad_total // 2 (= 0 here,
    1 on the right).  The forward operator is written out as an explicit matrix per (input channel, output channel) pair, transposed,
    and compared with the oracle's closed form o[2t] = x[t].K0 + x[t-1].K2, o[2t+1] = x[t].K1 (modules.py:209-258, SURVEY [TF-sem])."""
    rng = np.random.default_rng(7)
    T, Cin, Cout, k, stride = 9, 5, 4, 3, 2
    n_in = 2 * T                                        # the forward conv's input length = the transposed conv's output length
    n_out = -(-n_in // stride)
    assert n_out == T
    pad_total = max((n_out - 1) * stride + k - n_in, 0)
    pad_left = pad_total // 2
    assert (pad_total, pad_left) == (1, 0)
    # conv2d_transpose's kernel variable is (1, k, Cout, Cin): filters of the FORWARD conv from Cout-channel input to Cin-channel output
    Kt = rng.standard_normal((1, k, Cout, Cin)).astype(np.float32)
    x = rng.standard_normal((1, T, Cin)).astype(np.float32)
    # forward: y[t, ci] = sum_j sum_co in[2 t + j - pad_left, co] * Kt[0, j, co, ci]   ->   matrix F of shape (T * Cin, 2T * Cout)
    Fm = np.zeros((T * Cin, n_in * Cout), np.float64)
    for t in range(T):
        for j in range(k):
            p = stride * t + j - pad_left
            if 0 <= p < n_in:
                for co in range(Cout):
                    for ci in range(Cin):
                        Fm[t * Cin + ci, p * Cout + co] += Kt[0, j, co, ci]
    raw = (Fm.T @ x[0].reshape(-1).astype(np.float64)).reshape(n_in, Cout)      # the adjoint applied to x
    W = {"D/conv2d_transpose/kernel": Kt, "D/conv2d_transpose/bias": np.zeros(Cout, np.float32),
         "D/normalize/gamma": np.ones(Cout, np.float32), "D/normalize/beta": np.zeros(Cout, np.float32)}
    mine = O.conv1d_transpose(x, W, "D")[0]
    mu = raw.mean(-1, keepdims=True)
    ref = (raw - mu) / np.sqrt(((raw - mu) ** 2).mean(-1, keepdims=True) + 1e-12)
    assert np.abs(mine - ref).max() < 1e-5
