"""ctypes wrapper of oracle/oph_cpu.c.  TEST / BASELINE INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np

from . import build_oracle
from . import ophelia_oracle as O


class CpuDims(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("vocab", "e", "d", "c", "n_mels", "full_dim", "r", "max_N", "max_T", "win",
                                       "nspeakers", "spk_emb", "multispeaker")]


_f = C.POINTER(C.c_float)
_i = C.POINTER(C.c_int32)
_lib = None


def lib():
    global _lib
    if _lib is None:
        l = C.CDLL(build_oracle.build())
        l.oph_cpu_threads.restype = C.c_int
        l.oph_cpu_set_threads.argtypes = [C.c_int]
        l.oph_cpu_text_enc.argtypes = [C.POINTER(CpuDims), _f, _i, C.c_int, _f, _f]
        l.oph_cpu_text2mel.argtypes = [C.POINTER(CpuDims), _f, _f, _f, _i, _i, C.c_int, C.c_int, C.c_int, _f, _i, _f, _i]
        l.oph_cpu_ssrn.argtypes = [C.POINTER(CpuDims), _f, _f, C.c_int, C.c_int, _f]
        _lib = l
    return _lib


def usable_cores():
    """Cores this process may actually use: min(affinity mask, cgroup cpu quota)."""
    import math
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, math.floor(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return n


def dims(hp):
    ms = "audio_decoder_input" in (hp.multispeaker or [])
    return CpuDims(len(hp.vocab), hp.e, hp.d, hp.c, hp.n_mels, hp.full_dim, hp.r, hp.max_N, hp.max_T,
                   hp.attention_win_size, getattr(hp, "nspeakers", 0) if ms else 0,
                   getattr(hp, "speaker_embedding_size", 0) if ms else 0, int(ms))


def flat(hp, W, scope):
    names = [n for n in O.variable_shapes(hp) if n.startswith(scope)]
    return np.ascontiguousarray(np.concatenate([np.asarray(W[n], np.float32).ravel() for n in names]))


def _p(a, t=_f):
    return a.ctypes.data_as(t)


class CpuModel(object):
    def __init__(self, hp, W, threads=None):
        self.hp, self.d = hp, dims(hp)
        self.w_t2m = flat(hp, W, "Text2Mel")
        self.w_ssrn = flat(hp, W, "SSRN")
        if threads:
            lib().oph_cpu_set_threads(int(threads))
        self.threads = lib().oph_cpu_threads()

    def encode_text(self, L):
        L = np.ascontiguousarray(L, np.int32)
        B = len(L)
        K = np.empty((B, self.hp.max_N, self.hp.d), np.float32); V = np.empty_like(K)
        lib().oph_cpu_text_enc(C.byref(self.d), _p(self.w_t2m), _p(L, _i), B, _p(K), _p(V))
        return K, V

    def text2mel(self, K, V, ends, speakers=None, stop=True, max_steps=None):
        hp = self.hp
        B = len(K)
        K = np.ascontiguousarray(K, np.float32); V = np.ascontiguousarray(V, np.float32)
        ends = np.ascontiguousarray(ends, np.int32)
        spk = np.ascontiguousarray(np.zeros(B) if speakers is None else np.asarray(speakers).reshape(B), np.int32)
        Y = np.empty((B, hp.max_T, hp.n_mels), np.float32)
        al = np.empty((B, hp.max_N, hp.max_T), np.float32)
        t_ends = np.empty(B, np.int32); steps = C.c_int32()
        lib().oph_cpu_text2mel(C.byref(self.d), _p(self.w_t2m), _p(K), _p(V), _p(ends, _i), _p(spk, _i), B,
                               0 if stop else 1, hp.max_T if max_steps is None else int(max_steps),
                               _p(Y), _p(t_ends, _i), _p(al), C.byref(steps))
        return Y, t_ends.tolist(), al, steps.value

    def ssrn(self, Y):
        Y = np.ascontiguousarray(Y, np.float32)
        B, T, _ = Y.shape
        Z = np.empty((B, T * self.hp.r, self.hp.full_dim), np.float32)
        lib().oph_cpu_ssrn(C.byref(self.d), _p(self.w_ssrn), _p(Y), B, T, _p(Z))
        return Z
