"""Weight container helpers.  Weights are a dict {TF variable name: float32 ndarray},
named exactly as the reference graph names them, e.g.
'Text2Mel/TextEnc/HC_4/conv1d/kernel' (3,512,1024).  The inventory (names, shapes)
comes from the library itself (oph_weight_info)."""
import zlib

import numpy as np


def _trunc_normal(rng, shape, std):
    x = rng.standard_normal(shape)
    bad = np.abs(x) > 2.0
    while bad.any():
        x[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(x) > 2.0
    return (x * std).astype(np.float32)


def random_weights(inventory, seed):
    """Seeded synthetic weights for benchmarking / smoke runs (there are no checkpoints
    offline): conv kernels truncated-normal std sqrt(1.3*2/fan_in) (the reference's
    variance_scaling_initializer defaults, modules.py:134,191,249), embeddings TN std
    0.1 (modules.py:37), bias ~N(0,.02), gamma ~1+N(0,.05), beta ~N(0,.05).
    `inventory`: iterable of (name, shape)."""
    W = {}
    for name, shape in inventory:
        shape = tuple(int(s) for s in shape)
        rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))
        if name.endswith("lookup_table"):
            W[name] = _trunc_normal(rng, shape, 0.1)
        elif name.endswith("conv1d/kernel"):
            W[name] = _trunc_normal(rng, shape, np.sqrt(1.3 * 2.0 / (shape[0] * shape[1])))
        elif name.endswith("conv2d_transpose/kernel"):
            W[name] = _trunc_normal(rng, shape, np.sqrt(1.3 * 2.0 / (shape[1] * shape[2])))
        elif name.endswith("bias"):
            W[name] = (rng.standard_normal(shape) * 0.02).astype(np.float32)
        elif name.endswith("gamma"):
            W[name] = (1.0 + rng.standard_normal(shape) * 0.05).astype(np.float32)
        elif name.endswith("beta"):
            W[name] = (rng.standard_normal(shape) * 0.05).astype(np.float32)
        else:
            raise KeyError(name)
    return W


def save_npz(path, W):
    np.savez(path, **{k.replace("/", "|"): v for k, v in W.items()})


def load_npz(path):
    with np.load(path) as z:
        return {k.replace("|", "/"): np.asarray(z[k], np.float32) for k in z.files}


def flatten(W, inventory):
    """One contiguous float32 vector in inventory order (for the RCCL broadcast)."""
    return np.concatenate([np.asarray(W[n], np.float32).ravel() for n, _ in inventory])


def unflatten(flat, inventory):
    W, o = {}, 0
    for n, shape in inventory:
        k = int(np.prod(shape))
        W[n] = np.asarray(flat[o:o + k], np.float32).reshape(shape)
        o += k
    assert o == flat.size
    return W
