"""Operator-level mirror of the reference's modules.py (embed 15-44, normalize 47-75,
conv1d 91-146, hc 148-207, conv1d_transpose 209-258) plus networks.Attention
(networks.py:286-325, monotonic branch).  Same names, argument meaning and (B,T,C)
layout; variables come from a dict keyed by the TF variable names under `scope`.
Every function executes on the GPU through the per-operator C-ABI entry points."""
import ctypes as C

import numpy as np

from . import _lib

_ACT = {None: 0, "relu": 1, "sigmoid": 2}
_PAD = {"same": 0, "causal": 1}


def _chk(rc):
    if rc != 0:
        raise _lib.OpheliaHipError("libophelia_hip op error %d: %s" % (rc, _lib.load().oph_op_last_error().decode()))


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def embed(inputs, lookup_table, zero_pad=True, device=0):
    if not zero_pad:
        raise NotImplementedError("the reference always uses zero_pad=True on this path")
    ids = np.ascontiguousarray(inputs, dtype=np.int32)
    tab = _f(lookup_table)
    out = np.empty(ids.shape + (tab.shape[1],), np.float32)
    _chk(_lib.load().oph_op_embed(device, _lib.iptr(ids), ids.size, _lib.fptr(tab), tab.shape[0], tab.shape[1], _lib.fptr(out)))
    return out


def normalize(inputs, W, scope="normalize", normtype="layer", device=0):
    if normtype != "layer":
        raise NotImplementedError("only normtype='layer' is on the hot path (every shipped config)")
    x = _f(inputs)
    C_ = x.shape[-1]
    g, b = _f(W[scope + "/gamma"]), _f(W[scope + "/beta"])
    y = np.empty_like(x)
    _chk(_lib.load().oph_op_layernorm(device, _lib.fptr(x), x.size // C_, C_, _lib.fptr(g), _lib.fptr(b), _lib.fptr(y)))
    return y


def conv1d(inputs, W, scope="conv1d", filters=None, size=1, rate=1, padding="SAME", activation_fn=None, device=0):
    x = _f(inputs)
    B, T, Cin = x.shape
    k, bias = _f(W[scope + "/conv1d/kernel"]), _f(W[scope + "/conv1d/bias"])
    g, b = _f(W[scope + "/normalize/gamma"]), _f(W[scope + "/normalize/beta"])
    assert k.shape[:2] == (size, Cin)
    Cout = k.shape[2]
    if filters is not None:
        assert filters == Cout
    y = np.empty((B, T, Cout), np.float32)
    _chk(_lib.load().oph_op_conv1d(device, _lib.fptr(x), B, T, Cin, Cout, size, rate, _PAD[padding.lower()],
                                   _lib.fptr(k), _lib.fptr(bias), _lib.fptr(g), _lib.fptr(b), _ACT[activation_fn], _lib.fptr(y)))
    return y


def hc(inputs, W, scope="hc", size=1, rate=1, padding="SAME", device=0):
    x = _f(inputs)
    B, T, Cc = x.shape
    k, bias = _f(W[scope + "/conv1d/kernel"]), _f(W[scope + "/conv1d/bias"])
    assert k.shape == (size, Cc, 2 * Cc)
    y = np.empty_like(x)
    _chk(_lib.load().oph_op_hc(device, _lib.fptr(x), B, T, Cc, size, rate, _PAD[padding.lower()], _lib.fptr(k), _lib.fptr(bias),
                               _lib.fptr(_f(W[scope + "/H1/gamma"])), _lib.fptr(_f(W[scope + "/H1/beta"])),
                               _lib.fptr(_f(W[scope + "/H2/gamma"])), _lib.fptr(_f(W[scope + "/H2/beta"])), _lib.fptr(y)))
    return y


def conv1d_transpose(inputs, W, scope="conv1d_transpose", device=0, precision=0):
    """precision: the arithmetic of the contraction, as Engine.set_precision's codes -- 0 fp32-operand MFMA, 1 split-bf16 x3,
    2 split-fp16 x3 (what the SSRN path runs by default)."""
    x = _f(inputs)
    B, T, Cin = x.shape
    k = _f(W[scope + "/conv2d_transpose/kernel"])          # (1, 3, Cout, Cin)
    assert k.shape[0] == 1 and k.shape[1] == 3 and k.shape[3] == Cin
    Cout = k.shape[2]
    y = np.empty((B, 2 * T, Cout), np.float32)
    _chk(_lib.load().oph_op_conv1d_transpose_prec(device, _lib.fptr(x), B, T, Cin, Cout, _lib.fptr(k),
                                                  _lib.fptr(_f(W[scope + "/conv2d_transpose/bias"])),
                                                  _lib.fptr(_f(W[scope + "/normalize/gamma"])),
                                                  _lib.fptr(_f(W[scope + "/normalize/beta"])), int(precision), _lib.fptr(y)))
    return y


def attention(hp, Q, K, V, prev_max_attentions, device=0):
    """networks.Attention(monotonic_attention=True): returns R (B,T,2d), alignments (B,N,T), max_attentions (B,T)."""
    Q, K, V = _f(Q), _f(K), _f(V)
    B, T, d = Q.shape
    N = K.shape[1]
    assert N == hp.max_N, "the reference builds its masks with hp.max_N (networks.py:304-305)"
    p = np.ascontiguousarray(prev_max_attentions, dtype=np.int32)
    R = np.empty((B, T, 2 * d), np.float32)
    al = np.empty((B, N, T), np.float32)
    mx = np.empty((B, T), np.int64)
    _chk(_lib.load().oph_op_attention(device, _lib.fptr(Q), _lib.fptr(K), _lib.fptr(V), _lib.iptr(p), B, T, N, d,
                                      hp.attention_win_size, _lib.fptr(R), _lib.fptr(al), mx.ctypes.data_as(_lib.c_i64p)))
    return R, al, mx
