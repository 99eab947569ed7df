"""BASELINE configs C4 and C5 at their TOTAL sizes on one GPU (the driver has no multi-GPU box at test time):
  C4  lj_tutorial.cfg, 128 synthetic utterances (8 row tiles of 16), 200 fixed-length steps
  C5  vctk_01.cfg multispeaker (speaker-embedding path), 32 utterances with speaker ids
The whole batch runs through the C ABI in ONE call; every utterance is compared with the oracle's exact incremental
algorithm (C5: all 32; C4: a stride-4 sample that touches all 8 row tiles) and -- a size-independent property --
with the same utterances run as the contiguous 16-utterance shards the multi-GPU layout gives each rank."""
import numpy as np
import pytest

from conftest import hp_from_snapshot
from oracle import ophelia_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _engine(hp, W):
    from ophelia_amd.engine import Engine
    eng = Engine(hp, device=0)
    eng.load_weights(W)
    return eng


def _oracle_check(hp, W, K, V, ends, idx, Y, al, Z, speakers=None):
    spk = None if speakers is None else speakers[idx]
    trace = []
    Y0, t0, al0 = O.synth_codedtext2mel_incremental(hp, W, K[idx], V[idx], ends[idx], speakers=spk, stop=False, trace=trace)
    assert np.array_equal(al[idx].argmax(1).T, np.array(trace)), "attention argmax trace diverged"
    ey, ea = np.abs(Y[idx] - Y0).max(), np.abs(al[idx] - al0).max()
    Z0 = O.synth_mel2mag(hp, W, Y0[:4])                      # SSRN is per-utterance too: four of them pin it
    ez = np.abs(Z[idx[:4]] - Z0).max()
    print("max-abs vs oracle: Y %.3e align %.3e Z %.3e (%d utterances)" % (ey, ea, ez, len(idx)))
    assert ey < TOL and ea < TOL and ez < TOL


def _ssrn_fp32_leg(eng, hp, W, Y, idx):
    """The fp32-operand MFMA flavour of SSRN (oph_set_ssrn_precision 0) on the same mels, against the oracle as well."""
    eng.set_ssrn_precision(0)
    Zp = eng.ssrn(np.ascontiguousarray(Y[idx[:4]]))
    ez = np.abs(Zp - O.synth_mel2mag(hp, W, Y[idx[:4]])).max()
    print("SSRN fp32-operand flavour vs oracle: %.3e" % ez)
    assert ez < TOL
    eng.set_ssrn_precision(2)


def test_c4_lj_tutorial_batch_128():
    hp = hp_from_snapshot("lj_tutorial.cfg")
    W = O.random_weights(hp, 2)
    B = 128
    L = O.random_text(hp, B, 3, min_len=75, max_len=149)
    ends = O.get_text_lengths(L)
    eng = _engine(hp, W)                                     # the product's default arithmetic everywhere (split-fp16 x3 SSRN: what bench.py times)
    K, V = eng.encode_text(L)
    Y, t_ends, al, steps = eng.text2mel(K, V, ends, stop_mode=1)
    assert steps == hp.max_T and Y.shape == (B, hp.max_T, hp.n_mels)
    Z = eng.ssrn(Y)
    assert Z.shape == (B, hp.max_T * hp.r, hp.full_dim)
    # every rank's shard (16 utterances each) reproduces its slice of the 128-batch
    for r in (0, 3, 7):
        sl = slice(16 * r, 16 * r + 16)
        Ks, Vs = eng.encode_text(L[sl])
        assert np.abs(Ks - K[sl]).max() < 1e-6 and np.abs(Vs - V[sl]).max() < 1e-6
        Ys, ts, als, _ = eng.text2mel(K[sl], V[sl], ends[sl], stop_mode=1)
        assert np.array_equal(als.argmax(1), al[sl].argmax(1))
        assert np.abs(Ys - Y[sl]).max() < 1e-5 and np.abs(als - al[sl]).max() < 1e-5
        assert ts.tolist() == t_ends[sl].tolist()
    Zs = eng.ssrn(Y[:16])
    assert np.abs(Zs - Z[:16]).max() < 1e-6
    idx = np.arange(1, B, 4)                                 # 32 utterances, 4 from each of the 8 row tiles
    _oracle_check(hp, W, K, V, ends, idx, Y, al, Z)
    _ssrn_fp32_leg(eng, hp, W, Y, idx)
    eng.close()


def test_c5_vctk_multispeaker_batch_32():
    hp = hp_from_snapshot("vctk_01.cfg")
    assert "audio_decoder_input" in hp.multispeaker
    W = O.random_weights(hp, 5)
    B = 32
    L = O.random_text(hp, B, 6, min_len=60, max_len=hp.max_N - 1)
    ends = O.get_text_lengths(L)
    rng = np.random.Generator(np.random.PCG64(5))
    speakers = rng.integers(1, hp.nspeakers, size=(B, 1)).astype(np.int32)       # speaker ids ~U{1..}
    eng = _engine(hp, W)                                     # default arithmetic (split-fp16 x3 SSRN)
    K, V = eng.encode_text(L, speakers)
    K0, V0 = O.encode_text(hp, W, L[:8], speakers=speakers[:8])
    assert np.abs(K[:8] - K0).max() < TOL and np.abs(V[:8] - V0).max() < TOL
    Y, t_ends, al, steps = eng.text2mel(K, V, ends, speakers, stop_mode=1)
    assert steps == hp.max_T
    Z = eng.ssrn(Y)
    # the 4-GPU layout: 8 utterances per rank
    for r in (0, 2):
        sl = slice(8 * r, 8 * r + 8)
        Ys, ts, als, _ = eng.text2mel(K[sl], V[sl], ends[sl], speakers[sl], stop_mode=1)
        assert np.array_equal(als.argmax(1), al[sl].argmax(1))
        assert np.abs(Ys - Y[sl]).max() < 1e-5 and np.abs(als - al[sl]).max() < 1e-5
    _oracle_check(hp, W, K, V, ends, np.arange(B), Y, al, Z, speakers=speakers)
    _ssrn_fp32_leg(eng, hp, W, Y, np.arange(B))
    eng.close()
