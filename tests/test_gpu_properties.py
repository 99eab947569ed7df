"""Size-independent properties of the path, checked on the GPU at BASELINE size (B=16, max_N=150,
max_T=200) without needing the oracle:
  * determinism (bitwise) and batch independence: an utterance decoded alone equals the same utterance
    inside a batch of 16 (no op on the path mixes utterances; fixed-length mode removes the stop coupling);
  * attention: columns are probability vectors supported on the monotonic window, the argmax trace is
    non-decreasing and advances by at most win-1 per step (networks.py:303-315);
  * SSRN locality: perturbing mel frame t changes only the magnitude rows inside its receptive field.
    k=3 highway convs with rates 1,3 reach +-4 samples at each of the three rates (= 16 + 8 + 4 rows), the
    two 1024-wide highway convs +-2 rows; a stride-2 transposed conv maps sample t to outputs 2t..2t+2
    (o[2t] = x[t]K0 + x[t-1]K2, o[2t+1] = x[t]K1), i.e. one extra sample to the right at each of the two
    upsamplings (2 + 1 rows).  Frame t itself covers rows 4t..4t+3  =>  rows [4t-30, 4t+36]."""
import numpy as np
import pytest

from conftest import hp_from_snapshot

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full():
    from ophelia_amd.engine import Engine
    from ophelia_amd import weights as WT
    hp = hp_from_snapshot("lj_tutorial.cfg")
    eng = Engine(hp, device=0)
    eng.load_weights(WT.random_weights(eng.inventory(), 2))
    rng = np.random.Generator(np.random.PCG64(3))
    L = np.zeros((16, hp.max_N), np.int32)
    for b in range(16):
        n = int(rng.integers(75, 150))
        L[b, :n] = rng.integers(1, len(hp.vocab), size=n)
    ends = np.array([np.where(L[i] == 0)[0][0] for i in range(16)], np.int32)
    K, V = eng.encode_text(L)
    Y, t_ends, al, steps = eng.text2mel(K, V, ends, stop_mode=1)
    return hp, eng, L, ends, K, V, Y, al


def test_deterministic_and_batch_independent(full):
    hp, eng, L, ends, K, V, Y, al = full
    K2, V2 = eng.encode_text(L)
    Y2, _, al2, _ = eng.text2mel(K2, V2, ends, stop_mode=1)
    assert np.array_equal(K, K2) and np.array_equal(Y, Y2) and np.array_equal(al, al2)      # bitwise
    for i in (0, 7, 15):                                                                      # alone == in the batch
        Ki, Vi = eng.encode_text(L[i:i + 1])
        assert np.array_equal(Ki[0], K[i])
        Yi, _, ali, _ = eng.text2mel(Ki, Vi, ends[i:i + 1], stop_mode=1)
        assert np.array_equal(Yi[0], Y[i]) and np.array_equal(ali[0], al[i])


def test_attention_window_and_monotonic_trace(full):
    hp, eng, L, ends, K, V, Y, al = full
    assert np.allclose(al.sum(axis=1), 1.0, atol=1e-5)                 # every column is a distribution over keys
    assert (al >= 0).all() and ((al > 0).sum(axis=1) <= hp.attention_win_size).all()
    trace = al.argmax(axis=1)                                           # (B, T): max_attentions
    prev = np.concatenate((np.zeros((16, 1), np.int64), trace[:, :-1]), 1)
    assert (trace >= prev).all() and (trace - prev <= hp.attention_win_size - 1).all()
    # support of column t is inside [prev_max, prev_max + win)
    n = np.arange(hp.max_N)[None, :, None]
    inside = (n >= prev[:, None, :]) & (n < prev[:, None, :] + hp.attention_win_size)
    assert not (al[~inside] != 0).any()
    assert (Y > 0).all() and (Y < 1).all()                              # sigmoid outputs


def test_ssrn_receptive_field_locality(full):
    hp, eng, L, ends, K, V, Y, al = full
    eng.set_ssrn_precision(0)
    Z0 = eng.ssrn(Y[:2])
    Yp = Y[:2].copy()
    t = 100
    Yp[1, t] = 1.0 - Yp[1, t]
    Z1 = eng.ssrn(Yp)
    assert np.array_equal(Z0[0], Z1[0])                                 # the other utterance is untouched
    changed = np.where(np.abs(Z1[1] - Z0[1]).max(axis=1) > 0)[0]
    assert changed.min() == 4 * t - 30 and changed.max() == 4 * t + 36
    assert Z0.shape == (2, hp.max_T * hp.r, hp.full_dim) and (Z0 > 0).all() and (Z0 < 1).all()
    eng.set_ssrn_precision(2)
