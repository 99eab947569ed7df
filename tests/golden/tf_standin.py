# -*- coding: utf-8 -*-
"""
Golden-vector GENERATION tool (runs only in the build container; never shipped,
never imported by the product, never runs on the GPU box).

A tiny *eager* stand-in for the ~35 TensorFlow-1.12 symbols that the reference's
synth-mode graph code (architectures.py / networks.py / modules.py) touches, so
that the reference's OWN wiring code can be executed under Python 3 without
TensorFlow (which is absent here: SURVEY.md section 8c).  It is registered as the
module name `tensorflow` by make_golden.py *before* importing the reference.

The primitives are implemented with torch CPU ops (an implementation independent
of oracle/ophelia_oracle.py), following TF-1.12 semantics as read by the builder:
  layers.conv1d            -> F.conv1d on NCW, 'valid' / 'same' (+dilation)
  layers.conv2d_transpose  -> F.conv_transpose2d(stride=(1,2)) cropped to 2T ('same')
  contrib.layers.layer_norm-> F.layer_norm(eps=1e-12) over the last axis
  nn.softmax               -> torch.softmax(-1)
Variables are NOT initialised here: get_variable()/layers look the value up, by full
scope path, in VARS (filled by the caller with seeded weights) and assert the shape
the reference asks for -- which pins the reference's variable names and shapes.
"""
import contextlib
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

VARS = {}                 # full variable name -> np.float32 array (set by caller)
REQUESTED = []            # names requested by the reference code, in creation order
PLACEHOLDER_QUEUE = []    # values handed out to tf.placeholder() in creation order
_scope_stack = []

float32 = np.float32
int32 = np.int32
int64 = np.int64


class Tensor(np.ndarray):
    class _Shape(object):
        """TensorShape stand-in: NOT a tuple, so that '%s' % (t.shape) formats like it does in TF (modules.py:80)"""
        def __init__(self, s):
            self._s = list(s)

        def as_list(self):
            return list(self._s)

        def __iter__(self):
            return iter(self._s)

        def __len__(self):
            return len(self._s)

        def __getitem__(self, i):
            return self._s[i]

        def __eq__(self, other):
            return list(self._s) == list(other)

        def __repr__(self):
            return "(%s)" % ", ".join(str(v) for v in self._s)

    @property
    def shape(self):
        return Tensor._Shape(np.ndarray.shape.__get__(self))

    def get_shape(self):
        return Tensor._Shape(np.ndarray.shape.__get__(self))


def _t(x):
    return np.asarray(x).view(Tensor)


def convert_to_tensor(x, *a, **k):
    return _t(x)


@contextlib.contextmanager
def variable_scope(name, reuse=None, **kw):
    _scope_stack.append(name)
    try:
        yield
    finally:
        _scope_stack.pop()


def _full(name):
    return "/".join(_scope_stack + [name])


def get_variable(name, dtype=None, shape=None, initializer=None, **kw):
    full = _full(name)
    if full not in VARS:
        raise KeyError("reference asked for variable %r not supplied" % full)
    v = VARS[full]
    if shape is not None:
        assert list(v.shape) == list(shape), (full, v.shape, shape)
    if full not in REQUESTED:
        REQUESTED.append(full)
    return _t(v)


def placeholder(dtype, shape=None, name=None):
    return _t(PLACEHOLDER_QUEUE.pop(0))


def truncated_normal_initializer(mean=0.0, stddev=1.0, **kw):
    return ("truncated_normal", mean, stddev)


def concat(values, axis):
    return _t(np.concatenate([np.asarray(v) for v in values], axis))


def zeros(shape, dtype=np.float32):
    return _t(np.zeros(shape, dtype))


def zeros_like(x):
    return _t(np.zeros_like(np.asarray(x)))


def ones_like(x):
    return _t(np.ones_like(np.asarray(x)))


def ones(shape, dtype=np.float32):
    return _t(np.ones(shape, dtype))


def pad(x, paddings):
    return _t(np.pad(np.asarray(x), paddings))


def split(x, n, axis=-1):
    return [_t(a) for a in np.split(np.asarray(x), n, axis)]


def expand_dims(x, axis):
    return _t(np.expand_dims(np.asarray(x), axis))


def squeeze(x, axis=None):
    return _t(np.squeeze(np.asarray(x), axis))


def tile(x, multiples):
    return _t(np.tile(np.asarray(x), [int(m) for m in multiples]))


def shape(x):
    return list(np.asarray(x).shape)


def matmul(a, b, transpose_b=False):
    a = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    b = torch.from_numpy(np.ascontiguousarray(b, dtype=np.float32))
    if transpose_b:
        b = b.transpose(-1, -2)
    return _t(torch.matmul(a, b).numpy())


def rsqrt(x):
    return np.float32(1.0) / np.sqrt(np.float32(x))


def to_float(x):
    return np.float32(x)


def sequence_mask(lengths, maxlen):
    lengths = np.asarray(lengths).reshape(-1, 1)
    return _t(np.arange(maxlen).reshape(1, -1) < lengths)


def logical_or(a, b):
    return _t(np.logical_or(a, b))


def equal(a, b):
    return _t(np.asarray(a) == b)


def where(cond, x, y):
    return _t(np.where(np.asarray(cond), np.asarray(x), np.asarray(y)).astype(np.float32))


def argmax(x, axis):
    return _t(np.asarray(x).argmax(axis))


def transpose(x, perm):
    return _t(np.transpose(np.asarray(x), perm))


# ---- tf.nn -----------------------------------------------------------------
nn = types.ModuleType("tensorflow.nn")


def _embedding_lookup(table, ids):
    return _t(np.asarray(table)[np.asarray(ids).astype(np.int64)])


def _sigmoid(x, name=None):
    return _t(torch.sigmoid(torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))).numpy())


def _relu(x, name=None):
    return _t(np.maximum(np.asarray(x), np.float32(0)))


def _softmax(x, axis=-1):
    return _t(torch.softmax(torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)), -1).numpy())


nn.embedding_lookup = _embedding_lookup
nn.sigmoid = _sigmoid
nn.relu = _relu
nn.softmax = _softmax

# ---- tf.layers -------------------------------------------------------------
layers = types.ModuleType("tensorflow.layers")


def _conv1d(inputs, filters, kernel_size, dilation_rate=1, padding="valid", use_bias=True,
            kernel_initializer=None, reuse=None, **kw):
    x = np.ascontiguousarray(inputs, dtype=np.float32)
    cin = x.shape[-1]
    with variable_scope("conv1d"):
        kernel = get_variable("kernel", shape=[kernel_size, cin, filters])
        bias = get_variable("bias", shape=[filters]) if use_bias else None
    w = torch.from_numpy(np.ascontiguousarray(np.transpose(np.asarray(kernel), (2, 1, 0))))  # (Cout,Cin,k)
    xt = torch.from_numpy(x).transpose(1, 2)                                                 # (B,C,T)
    pad_mode = padding.lower()
    assert pad_mode in ("valid", "same")
    y = F.conv1d(xt, w, None if bias is None else torch.from_numpy(np.asarray(bias).copy()),
                 stride=1, padding=pad_mode, dilation=dilation_rate)
    return _t(y.transpose(1, 2).contiguous().numpy())


def _conv2d_transpose(inputs, filters, kernel_size, strides, padding="same", activation=None,
                      kernel_initializer=None, use_bias=True, **kw):
    x = np.ascontiguousarray(inputs, dtype=np.float32)       # (B,1,T,Cin)  NHWC
    assert padding.lower() == "same" and tuple(kernel_size) == (1, 3) and tuple(strides) == (1, 2)
    cin = x.shape[-1]
    with variable_scope("conv2d_transpose"):
        kernel = get_variable("kernel", shape=[1, 3, filters, cin])     # (kh,kw,Cout,Cin)
        bias = get_variable("bias", shape=[filters]) if use_bias else None
    w = torch.from_numpy(np.ascontiguousarray(np.transpose(np.asarray(kernel), (3, 2, 0, 1))))  # (Cin,Cout,kh,kw)
    xt = torch.from_numpy(np.ascontiguousarray(np.transpose(x, (0, 3, 1, 2))))                  # NCHW
    y = F.conv_transpose2d(xt, w, None if bias is None else torch.from_numpy(np.asarray(bias).copy()),
                           stride=(1, 2), padding=0)
    T = x.shape[2]
    y = y[:, :, :, :2 * T]            # TF 'same': output width = 2*T (drops the last sample)
    return _t(np.ascontiguousarray(np.transpose(y.numpy(), (0, 2, 3, 1))))


def _dropout(x, rate=0.0, training=False):
    assert not training
    return x


layers.conv1d = _conv1d
layers.conv2d_transpose = _conv2d_transpose
layers.dropout = _dropout

# ---- tf.contrib.layers -----------------------------------------------------
contrib = types.ModuleType("tensorflow.contrib")
contrib.layers = types.ModuleType("tensorflow.contrib.layers")


def _layer_norm(inputs, begin_norm_axis=-1, scope=None, reuse=None, **kw):
    assert begin_norm_axis == -1
    x = np.ascontiguousarray(inputs, dtype=np.float32)
    c = x.shape[-1]
    with variable_scope(scope):
        beta = get_variable("beta", shape=[c])
        gamma = get_variable("gamma", shape=[c])
    y = F.layer_norm(torch.from_numpy(x), (c,), torch.from_numpy(np.asarray(gamma).copy()),
                     torch.from_numpy(np.asarray(beta).copy()), eps=1e-12)
    return _t(y.numpy())


contrib.layers.layer_norm = _layer_norm
contrib.layers.variance_scaling_initializer = lambda *a, **k: ("variance_scaling",)
contrib.layers.batch_norm = None


def install():
    """Register this module as `tensorflow` (and empty stubs for the reference's other
    absent third-party imports) in sys.modules."""
    me = sys.modules[__name__]
    sys.modules["tensorflow"] = me
    sys.modules["tensorflow.nn"] = nn
    sys.modules["tensorflow.layers"] = layers
    sys.modules["tensorflow.contrib"] = contrib
    sys.modules["tensorflow.contrib.layers"] = contrib.layers
    for name in ("librosa", "librosa.filters", "librosa.effects", "soundfile"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
