"""GPU parity at the edges: ragged batch sizes (1, 10, 20 = two 16-row tiles), the other BASELINE
dimension sets (C1 lj_test: V=65, max_N=180, max_T=210, B=10; C5 vctk_01 multispeaker B=8), the
host-polled early stop (stop after >8 steps), argument validation and handle re-use across batches."""
import numpy as np
import pytest

from conftest import hp_from_snapshot
from oracle import ophelia_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _run(hp, W, L, spk=None, stop=True):
    from ophelia_amd.engine import Engine
    eng = Engine(hp, device=0)
    eng.load_weights(W)
    ends = O.get_text_lengths(L)
    K, V = eng.encode_text(L, spk)
    Y, t_ends, al, steps = eng.text2mel(K, V, ends, spk, stop_mode=0 if stop else 1)
    Z = eng.ssrn(Y)
    eng.close()
    return K, V, Y, t_ends, al, steps, Z


def _check(hp, W, L, spk=None, stop=True):
    K, V, Y, t_ends, al, steps, Z = _run(hp, W, L, spk, stop)
    ends = O.get_text_lengths(L)
    if getattr(hp, "turn_off_monotonic_for_synthesis", False):
        hp.text_lengths = ends + 1                         # synthesize.py:505-507 (read by the oracle's attention)
    K0, V0 = O.encode_text(hp, W, L, speakers=spk)
    trace = []
    Y0, t0, al0 = O.synth_codedtext2mel_incremental(hp, W, K0, V0, ends, speakers=spk, stop=stop, trace=trace)
    Z0 = O.synth_mel2mag(hp, W, Y0)
    assert steps == len(trace) and t_ends.tolist() == t0
    assert np.abs(K - K0).max() < TOL and np.abs(V - V0).max() < TOL
    assert np.abs(Y - Y0).max() < TOL and np.abs(al - al0).max() < TOL
    # SSRN runs its default split-bf16 contractions here: 3e-5 with LayerNorm, ~1e-4 without (nothing re-normalises the
    # activations); the bar is 1e-3, asserted at a quarter of it like in test_gpu_model
    assert np.abs(Z - Z0).max() < 1e-3 / 4
    assert not Y[:, steps:].any() and not al[:, :, steps:].any()
    return steps, t0


@pytest.mark.parametrize("B", [1, 10, 20])
def test_ragged_batch_sizes(B):
    hp = hp_from_snapshot("lj_tutorial.cfg", max_N=40, max_T=30)
    W = O.random_weights(hp, 51)
    L = O.random_text(hp, B, 52, min_len=5, max_len=35)
    _check(hp, W, L, stop=False)


def test_c1_lj_test_dims_full_size():
    hp = hp_from_snapshot("lj_test.cfg")                     # V=65, max_N=180, max_T=210
    assert (len(hp.vocab), hp.max_N, hp.max_T) == (65, 180, 210)
    W = O.random_weights(hp, 1)
    L = O.random_text(hp, 10, 53, min_len=30, max_len=170)   # the 10-line test transcript case
    _check(hp, W, L, stop=False)


def test_c5_vctk_multispeaker_dims():
    hp = hp_from_snapshot("vctk_01.cfg")                     # max_N=80, max_T=100, 209 speakers, emb 128
    assert hp.multispeaker == ["audio_decoder_input"] and hp.nspeakers == 209
    W = O.random_weights(hp, 5)
    L = O.random_text(hp, 8, 54, min_len=20, max_len=79)
    spk = np.random.default_rng(5).integers(1, 109, size=(8, 1)).astype(np.int32)
    _check(hp, W, L, spk=spk, stop=False)
    # speaker id 0 is the padding speaker: its embedding row is zeroed at lookup (modules.py:38-40)
    spk[0, 0] = 0
    _check(hp, W, L, spk=spk, stop=False)


def test_variant_project_baseline_full_width_attention():
    """config/project/baseline.cfg: no LayerNorm, no monotonic window -- attention over every key of the text (+1);
    max_N = 150 spreads the keys over three lane slots of the full-attention path"""
    hp = hp_from_snapshot("project/baseline.cfg", max_T=24)
    assert hp.norm is None and hp.turn_off_monotonic_for_synthesis and hp.max_N == 150
    W = O.random_weights(hp, 61)
    L = O.random_text(hp, 5, 62, min_len=3, max_len=149)
    _check(hp, W, L, stop=False)


def test_variant_text_encoder_speaker_embeddings_full_width():
    """nancyplusnick_01.cfg ('text_encoder_input' + 'audio_decoder_input') and vctk_02.cfg ('text_encoder_towards_end'
    + 'audio_decoder_input') at their own max_N; speaker 0 is the zeroed padding row"""
    for cfg in ("nancyplusnick_01.cfg", "vctk_02.cfg", "vctk_03_lcc.cfg"):     # the last: learned channel contributions
        hp = hp_from_snapshot(cfg, max_T=20)
        W = O.random_weights(hp, 63)
        L = O.random_text(hp, 4, 64, min_len=10, max_len=hp.max_N - 1)
        spk = np.array([[1], [0], [hp.nspeakers - 1], [2]], np.int32)
        _check(hp, W, L, spk=spk, stop=False)


def test_variant_external_durations_full_width():
    """ssw10/G1AB_03.cfg at its own max_N = 173: ragged durations, zero-duration symbols, utterances of different
    length (the loop runs until the longest one is through), against the incremental oracle"""
    from ophelia_amd.data_load import durations_to_hard_attention_matrix, end_pad_for_reduction_shape_sync
    from ophelia_amd.engine import Engine
    hp = hp_from_snapshot("ssw10/G1AB_03.cfg", max_T=40)
    W = O.random_weights(hp, 65)
    B = 5
    L = O.random_text(hp, B, 66, min_len=3, max_len=hp.max_N - 1)
    ends = O.get_text_lengths(L)
    rng = np.random.default_rng(67)
    D = np.zeros((B, hp.max_T, hp.max_N), np.int32)
    for b in range(B):
        budget = int(rng.integers(8, hp.max_T + 1)) * hp.r - int(rng.integers(0, hp.r))
        dur = np.zeros(ends[b], np.int64)
        for _ in range(budget):
            dur[rng.integers(0, ends[b])] += 1
        A = end_pad_for_reduction_shape_sync(durations_to_hard_attention_matrix(dur), hp)[0::hp.r]
        D[b, :len(A), :A.shape[1]] = A
    eng = Engine(hp, device=0)
    eng.load_weights(W)
    K, V = eng.encode_text(L)
    Y, t_ends, al, steps = eng.text2mel_durations(K, V, D)
    eng.close()
    K0, V0 = O.encode_text(hp, W, L)
    Y0, t0, al0 = O.synth_codedtext2mel_incremental(hp, W, K0, V0, ends, durations=D)
    assert t_ends.tolist() == t0 == D.sum(axis=(1, 2)).tolist()
    assert steps == min(hp.max_T, max(t0) + 1)
    assert np.abs(Y - Y0).max() < TOL and np.array_equal(al, al0)


def test_reduction_factor_8_ssrn_has_three_upsampling_stages():
    """hp.r = 8 (networks.py:474-479): SSRN takes three stride-2 transposed convs, Z has 8 frames per mel frame"""
    hp = hp_from_snapshot("lj_tutorial.cfg", max_N=30, max_T=12)
    hp.r = 8
    W = O.random_weights(hp, 71)
    assert "SSRN/D_10/conv2d_transpose/kernel" in W          # C_1, HC_2-3, D_4, HC_5-6, D_7, HC_8-9, D_10
    L = O.random_text(hp, 3, 72, min_len=5, max_len=25)
    K, V, Y, t_ends, al, steps, Z = _run(hp, W, L, stop=False)
    assert Z.shape == (3, hp.max_T * 8, hp.full_dim)
    _check(hp, W, L, stop=False)


@pytest.mark.parametrize("e,d,c", [(64, 128, 256), (128, 192, 320), (32, 64, 128)])
def test_other_model_widths(e, d, c):
    """narrower models than the shipped e=128 / d=256 / c=512 (the kernels pad K to 32 and N to 16)"""
    hp = hp_from_snapshot("lj_tutorial.cfg", max_N=24, max_T=14)
    hp.e, hp.d, hp.c = e, d, c
    W = O.random_weights(hp, 73)
    L = O.random_text(hp, 4, 74, min_len=4, max_len=20)
    _check(hp, W, L, stop=False)


def test_host_polled_early_stop_after_many_steps():
    """short texts -> every utterance ends; the break step is past the first 8-step poll boundary"""
    hp = hp_from_snapshot("lj_tutorial.cfg", max_N=64, max_T=60)
    W = O.random_weights(hp, 43)
    rng = np.random.default_rng(1)
    L = np.zeros((6, hp.max_N), np.int32)
    for i, n in enumerate((2, 3, 4, 6, 8, 10)):
        L[i, :n] = rng.integers(1, len(hp.vocab), n)
    steps, t_ends = _check(hp, W, L, stop=True)
    assert 8 < steps < hp.max_T and max(t_ends) == steps - 1


def test_engine_reuse_across_batches_and_validation():
    from ophelia_amd.engine import Engine
    from ophelia_amd import _lib
    hp = hp_from_snapshot("lj_tutorial.cfg", max_N=32, max_T=20)
    W = O.random_weights(hp, 61)
    eng = Engine(hp, device=0)
    with pytest.raises(_lib.OpheliaHipError, match="not finalized"):
        eng.encode_text(np.zeros((2, 32), np.int32))
    bad = dict(W); bad.pop("SSRN/C_1/conv1d/bias")
    with pytest.raises(KeyError):
        eng.load_weights(bad)
    eng.load_weights(W)
    for seed in (1, 2):                  # same handle, two different batches: state is fully reset between them
        L = O.random_text(hp, 5, seed, min_len=4, max_len=30)
        ends = O.get_text_lengths(L)
        K, V = eng.encode_text(L)
        Y, t_ends, al, steps = eng.text2mel(K, V, ends, stop_mode=1)
        K0, V0 = O.encode_text(hp, W, L)
        Y0, t0, al0 = O.synth_codedtext2mel_incremental(hp, W, K0, V0, ends, stop=False)
        assert np.abs(Y - Y0).max() < TOL and t_ends.tolist() == t0
    # a batch with a different number of 16-row tiles on the same handle (state is rebuilt), then back again
    for B, seed in ((20, 3), (4, 4)):
        L = O.random_text(hp, B, seed, min_len=4, max_len=30)
        ends = O.get_text_lengths(L)
        K, V = eng.encode_text(L)
        Y, t_ends, al, steps = eng.text2mel(K, V, ends, stop_mode=1)
        K0, V0 = O.encode_text(hp, W, L)
        Y0, t0, al0 = O.synth_codedtext2mel_incremental(hp, W, K0, V0, ends, stop=False)
        assert np.abs(K - K0).max() < TOL and np.abs(Y - Y0).max() < TOL and t_ends.tolist() == t0
    with pytest.raises(_lib.OpheliaHipError, match="out of range"):
        eng.encode_text(np.full((2, 32), 999, np.int32))
    with pytest.raises(_lib.OpheliaHipError):
        eng.ssrn(np.zeros((2, hp.max_T + 5, hp.n_mels), np.float32))
    eng.close()


def test_random_flag_combinations():
    """option variants are independent switches in the reference; combinations no shipped config uses (e.g. LCC without
    LayerNorm and without the monotonic window, all three speaker positions at once, r = 8 with a narrow model) must
    behave like the oracle too"""
    rng = np.random.default_rng(4242)
    positions = ["text_encoder_input", "text_encoder_towards_end", "audio_encoder_input", "audio_decoder_input",
                 "learn_channel_contributions"]
    for case in range(14):
        hp = hp_from_snapshot("lj_tutorial.cfg", max_N=int(rng.integers(8, 40)), max_T=int(rng.integers(6, 20)))
        hp.e, hp.d, hp.c = int(rng.choice([32, 64, 128])), int(rng.choice([64, 128, 256])), int(rng.choice([128, 256]))
        hp.r = int(rng.choice([4, 8]))
        hp.norm = [None, "layer"][int(rng.integers(0, 2))]
        hp.turn_off_monotonic_for_synthesis = bool(rng.integers(0, 2))
        hp.attention_win_size = int(rng.integers(1, 6))
        hp.multispeaker = [p for p in positions if rng.random() < 0.4]
        hp.nspeakers, hp.speaker_embedding_size = int(rng.integers(2, 9)), int(rng.choice([16, 64, 128]))
        B = int(rng.integers(1, 6))
        W = O.random_weights(hp, 100 + case)
        L = O.random_text(hp, B, 200 + case, min_len=2, max_len=hp.max_N - 1)
        spk = rng.integers(0, hp.nspeakers, size=(B, 1)).astype(np.int32) if hp.multispeaker else None
        try:
            _check(hp, W, L, spk=spk, stop=bool(rng.integers(0, 2)))
        except AssertionError as e:
            raise AssertionError("case %d: %r" % (case, {k: getattr(hp, k) for k in (
                "max_N", "max_T", "e", "d", "c", "r", "norm", "turn_off_monotonic_for_synthesis", "attention_win_size",
                "multispeaker", "nspeakers", "speaker_embedding_size")})) from e
