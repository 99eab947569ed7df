"""GPU parity of libophelia_vocoder.so (through its C ABI) against oracle/griffin_lim_oracle.py.

Tolerances (float32 arithmetic on both sides, different FFT factorisations):
  stft      3e-6 of the largest bin          istft     3e-6 of the largest sample
  Griffin-Lim waveform after k iterations: the iteration is a fixed-point map with a discontinuity-free phase
  projection except at |est| ~ 0, so round-off grows slowly; bar = 2e-3 of the peak after 50 iterations, 1e-4 after 3.
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
pytestmark = pytest.mark.gpu

HP = SimpleNamespace(n_fft=2048, hop_length=275, win_length=1102, power=1.5, n_iter=50, preemphasis=0.97, max_db=100,
                     ref_db=20, sr=22050)


@pytest.fixture(scope="module")
def voc():
    from ophelia_amd.vocoder import Vocoder
    v = Vocoder(HP, 0)
    yield v
    v.close()


@pytest.fixture(scope="module")
def gl():
    from oracle import griffin_lim_oracle
    return griffin_lim_oracle


def _signal(n, seed=0):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 22050.0
    y = 0.4 * np.sin(2 * np.pi * 220 * t) + 0.2 * np.sin(2 * np.pi * 1730 * t + 1.0) + 0.05 * rng.standard_normal(n)
    return y.astype(np.float32)


def _speechlike_mag(T, seed):
    """smooth random (T, 1025) magnitudes in [0,1] with silence at both ends, like SSRN output"""
    rng = np.random.default_rng(seed)
    F = 1025
    env = np.clip(np.sin(np.linspace(0, np.pi, T)) * 1.5, 0, 1)[:, None]
    ridge = np.exp(-0.5 * ((np.arange(F)[None, :] - (120 + 80 * np.sin(np.arange(T)[:, None] / 9.0))) / 60.0) ** 2)
    harm = 0.5 + 0.5 * np.cos(np.arange(F)[None, :] / (6.0 + 2 * np.sin(np.arange(T)[:, None] / 15.0)))
    m = env * (0.25 + 0.55 * ridge * harm) + 0.03 * rng.standard_normal((T, F))
    return m.astype(np.float32)


@pytest.mark.parametrize("frames", [2, 5, 37, 300])
def test_stft(voc, gl, frames):
    y = _signal(HP.hop_length * (frames - 1), frames)
    D = voc.stft(y)
    ref = gl.stft(y, HP.n_fft, HP.hop_length, HP.win_length).T
    assert D.shape == ref.shape
    assert np.abs(D - ref).max() <= 3e-6 * np.abs(ref).max()


@pytest.mark.parametrize("frames", [2, 3, 23, 400])
def test_istft(voc, gl, frames):
    rng = np.random.default_rng(frames)
    D = (rng.standard_normal((frames, 1025)) + 1j * rng.standard_normal((frames, 1025))).astype(np.complex64)
    y = voc.istft(D)
    ref = gl.istft(D.T, HP.hop_length, HP.win_length)
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() <= 3e-6 * np.abs(ref).max()


def test_stft_istft_round_trip(voc):
    y = _signal(HP.hop_length * 99, 4)
    x = voc.istft(voc.stft(y))
    assert np.abs(x - y).max() < 2e-5


def test_deemphasis(voc):
    from scipy import signal
    for n in (1, 15, 1024, 1025, 50000):
        x = _signal(n, n)
        ref = signal.lfilter([1], [1, -HP.preemphasis], x).astype(np.float32)
        out = voc.deemphasis(x)
        assert np.abs(out - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()), n


def test_deemphasis_is_bit_exact_across_spans_and_slow_filters():
    """spans restart from a warm-up instead of the true state: still the float32 values of the sequential float64
    filter; a slowly decaying filter (0.999) runs sequentially"""
    from scipy import signal
    from ophelia_amd.vocoder import Vocoder
    x = _signal(70001, 3)
    for a in (0.97, 0.5, 0.999, 0.0):
        with Vocoder(SimpleNamespace(**dict(vars(HP), preemphasis=a)), 0) as v:
            ref = signal.lfilter([1], [1, -a], x).astype(np.float32)
            assert np.array_equal(v.deemphasis(x), ref), a


@pytest.fixture(params=[pytest.param(0, id="fused"), pytest.param(1, id="hipfft", marks=pytest.mark.hipfft_backend)])
def backend(request, voc):
    voc.set_backend(request.param)
    yield request.param
    voc.set_backend(0)


@pytest.mark.parametrize("n_iter,tol", [(0, 3e-6), (1, 2e-5), (3, 1e-4), (50, 2e-3)])
def test_griffin_lim_ragged_batch(voc, gl, backend, n_iter, tol):
    specs = [gl.amplitude_from_mag(HP, _speechlike_mag(T, T)).T.copy() for T in (40, 7, 64, 2, 3)]
    out = voc.griffin_lim_batch(specs, n_iter=n_iter)
    for S, y in zip(specs, out):
        ref = gl.griffin_lim(S.T, HP.n_fft, HP.hop_length, HP.win_length, n_iter)
        assert y.shape == ref.shape
        assert np.abs(y - ref).max() <= tol * np.abs(ref).max(), (S.shape, np.abs(y - ref).max(), np.abs(ref).max())


@pytest.mark.hipfft_backend
def test_backends_agree(voc):
    """fused in-LDS kernel vs hipFFT path: same arithmetic, different FFT factorisation"""
    mags = [_speechlike_mag(T, 3 * T) for T in (90, 2, 31)]
    voc.set_backend(1)
    a = voc.spectrogram2wav_batch(mags)
    voc.set_backend(0)
    b = voc.spectrogram2wav_batch(mags)
    for x, y in zip(a, b):
        assert np.abs(x - y).max() <= 2e-3 * np.abs(x).max()


BACKENDS = [pytest.param(0, id="fused"), pytest.param(1, id="hipfft", marks=pytest.mark.hipfft_backend)]


@pytest.mark.parametrize("be", BACKENDS)
@pytest.mark.parametrize("hop,win", [(200, 800), (256, 1024), (275, 1102), (512, 2048)])
def test_other_stft_geometries(gl, hop, win, be):
    """the 16 kHz configs (hop 200 / win 800) and nancy-style (256 / 1024) share n_fft = 2048"""
    from ophelia_amd.vocoder import Vocoder
    hp = SimpleNamespace(**dict(vars(HP), hop_length=hop, win_length=win, n_iter=4))
    with Vocoder(hp, 0) as v:
        mags = [_speechlike_mag(T, T + hop) for T in (25, 6)]
        for backend in (be,):
            v.set_backend(backend)
            for m, w in zip(mags, v.spectrogram2wav_batch(mags)):
                ref = gl.spectrogram2wav(hp, m)
                assert w.shape == ref.shape and np.abs(w - ref).max() <= 2e-4 * np.abs(ref).max(), (backend, hop)


@pytest.mark.hipfft_backend
def test_non_2048_fft_uses_generic_path(gl):
    from ophelia_amd.vocoder import Vocoder
    hp = SimpleNamespace(**dict(vars(HP), n_fft=1024, hop_length=256, win_length=1024, n_iter=3))
    with Vocoder(hp, 0) as v:
        m = _speechlike_mag(20, 9)[:, :513].copy()
        w = v.spectrogram2wav(m)
        ref = gl.spectrogram2wav(hp, m)
        assert np.abs(w - ref).max() <= 2e-4 * np.abs(ref).max()


def test_spectrogram2wav_matches_oracle(voc, gl):
    mags = [_speechlike_mag(T, 100 + T) for T in (48, 21)]
    mags[0][3, :5] = [-0.5, 1.5, 0.0, 1.0, 0.5]                 # exercises the clip
    out = voc.spectrogram2wav_batch(mags)
    for m, w in zip(mags, out):
        ref = gl.spectrogram2wav(HP, m)
        assert w.dtype == np.float32 and w.shape == ref.shape == (HP.hop_length * (len(m) - 1),)
        assert np.abs(w - ref).max() <= 2e-3 * np.abs(ref).max()


@pytest.mark.hipfft_backend
def test_full_size_batch_backends_agree(voc):
    """the bench workload (16 utterances x 800 frames, 50 iterations): fused kernel vs hipFFT path, every utterance"""
    mags = [_speechlike_mag(800, 1000 + b) for b in range(16)]
    voc.set_backend(1)
    a = voc.spectrogram2wav_batch(mags)
    voc.set_backend(0)
    b = voc.spectrogram2wav_batch(mags)
    for x, y in zip(a, b):
        assert x.shape == y.shape == (HP.hop_length * 799,)
        assert np.isfinite(y).all() and np.abs(x - y).max() <= 2e-3 * np.abs(x).max()


def test_batch_equals_single(voc):
    mags = [_speechlike_mag(T, 7 * T) for T in (33, 12, 50)]
    together = voc.spectrogram2wav_batch(mags)
    for m, w in zip(mags, together):
        alone = voc.spectrogram2wav(m)
        assert np.array_equal(alone, w)                        # utterances never interact, bitwise


def test_module_level_api_and_wav_file(voc, tmp_path):
    import wave
    from ophelia_amd import vocoder
    m = _speechlike_mag(30, 5)
    w = vocoder.spectrogram2wav(HP, m)
    assert np.array_equal(w, voc.spectrogram2wav(m))
    with pytest.raises(NotImplementedError):
        vocoder.spectrogram2wav(HP, m, trim_output=True)
    p = str(tmp_path / "a.wav")
    vocoder.write_wav(p, w, HP.sr)
    with wave.open(p, "rb") as f:
        assert (f.getnchannels(), f.getsampwidth(), f.getframerate(), f.getnframes()) == (1, 2, HP.sr, len(w))
        pcm = np.frombuffer(f.readframes(len(w)), "<i2")
    assert np.abs(pcm - np.clip(np.rint(w * 32767.0), -32768, 32767)).max() == 0


def test_rejects_bad_input(voc):
    from ophelia_amd._lib import OpheliaHipError
    with pytest.raises(OpheliaHipError):
        voc.spectrogram2wav_batch([np.zeros((1, 1025), np.float32)])
    with pytest.raises(ValueError):
        voc.spectrogram2wav_batch([np.zeros((5, 1024), np.float32)])


def test_from_engine_resident_mag(voc):
    """Z stays in HBM between SSRN and the vocoder: same samples as the host-buffer route."""
    from conftest import hp_from_snapshot
    from ophelia_amd.engine import Engine
    from oracle import ophelia_oracle as orc
    hp = hp_from_snapshot("lj_tutorial.cfg", max_N=40, max_T=24)
    eng = Engine(hp, 0)
    eng.load_weights(orc.random_weights(hp, 3))
    L = orc.random_text(hp, 3, 5)
    ends = np.array([int(np.count_nonzero(row)) for row in L], np.int32)
    eng.stage_text(L, ends)
    eng.run_resident()
    Z = eng.fetch_mag()
    n_frames = np.array([hp.max_T * hp.r, 17, 40], np.int32)
    a = voc.spectrogram2wav_from_engine(eng, n_frames)
    b = voc.spectrogram2wav_batch([Z[i, :n] for i, n in enumerate(n_frames)])
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    eng.close()


@pytest.mark.parametrize("be", BACKENDS)
def test_random_stft_geometries(gl, be):
    """hop / window lengths no shipped config uses (odd values, window = n_fft, hop = window, tiny hops are refused
    only when more than the supported overlap would be needed by the generic gather -- none here), ragged batches"""
    from ophelia_amd.vocoder import Vocoder
    rng = np.random.default_rng(77)
    for case in range(8):
        win = int(rng.integers(300, 2049))
        hop = int(rng.integers(max(64, win // 5 + 1), win + 1))
        hp = SimpleNamespace(**dict(vars(HP), hop_length=hop, win_length=win, n_iter=int(rng.integers(0, 4)),
                                    power=float(rng.choice([1.0, 1.2, 1.5])), preemphasis=float(rng.choice([0.0, 0.9, 0.97]))))
        mags = [_speechlike_mag(int(T), case * 10 + int(T)) for T in rng.integers(2, 40, size=3)]
        with Vocoder(hp, 0) as v:
            for backend in (be,):
                v.set_backend(backend)
                for m, w in zip(mags, v.spectrogram2wav_batch(mags)):
                    ref = gl.spectrogram2wav(hp, m)
                    assert w.shape == ref.shape, (case, hop, win)
                    assert np.abs(w - ref).max() <= 3e-4 * max(np.abs(ref).max(), 1e-6), (case, backend, hop, win, hp.n_iter)
