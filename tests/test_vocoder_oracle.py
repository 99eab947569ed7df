"""Pins oracle/griffin_lim_oracle.py (restated librosa 0.6.2 stft/istft + reference utils.py:69-116) as far as is possible
offline: against scipy.signal's independent STFT/ISTFT, by reconstruction, and by Griffin-Lim's convergence property."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
from scipy import signal

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import griffin_lim_oracle as gl  # noqa: E402

HP = SimpleNamespace(n_fft=2048, hop_length=275, win_length=1102, power=1.5, n_iter=50, preemphasis=0.97, max_db=100,
                     ref_db=20, sr=22050)


def _signal(n, seed=0):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 22050.0
    y = 0.4 * np.sin(2 * np.pi * 220 * t) + 0.2 * np.sin(2 * np.pi * 1730 * t + 1.0) + 0.05 * rng.standard_normal(n)
    return y.astype(np.float32)


def test_window_matches_scipy():
    w = signal.get_window("hann", HP.win_length, fftbins=True)
    assert np.abs(gl.hann_periodic(HP.win_length) - w).max() < 1e-15
    p = gl.padded_window(HP.n_fft, HP.win_length)
    assert p[:473].sum() == 0 and p[1575:].sum() == 0 and np.abs(p[473:1575] - w).max() < 1e-15


def test_reflect_index_matches_numpy_pad():
    for n, pad in ((5, 3), (7, 20), (2, 9), (1300, 1024)):
        y = np.arange(n, dtype=np.float64)
        ref = np.pad(y, pad, mode="reflect")
        assert np.array_equal(y[gl.reflect_index(np.arange(-pad, n + pad), n)], ref)


@pytest.mark.parametrize("n_fft,hop,win", [(2048, 275, 1102), (512, 128, 512), (256, 50, 200)])
def test_stft_matches_scipy(n_fft, hop, win):
    y = _signal(hop * 37)
    D = gl.stft(y, n_fft, hop, win)
    assert D.dtype == np.complex64 and D.shape == (1 + n_fft // 2, 1 + len(y) // hop)
    w = gl.padded_window(n_fft, win)
    ypad = np.pad(y.astype(np.float64), n_fft // 2, mode="reflect")
    _, _, Z = signal.stft(ypad, window=w, nperseg=n_fft, noverlap=n_fft - hop, boundary=None, padded=False,
                          return_onesided=True)
    Z = Z * w.sum()                                       # scipy scales by 1/sum(w)
    assert Z.shape == D.shape
    assert np.abs(D - Z).max() <= 2e-6 * np.abs(Z).max()


@pytest.mark.parametrize("n_fft,hop,win", [(2048, 275, 1102), (512, 128, 512)])
def test_istft_matches_scipy(n_fft, hop, win):
    rng = np.random.default_rng(3)
    T = 23
    D = (rng.standard_normal((1 + n_fft // 2, T)) + 1j * rng.standard_normal((1 + n_fft // 2, T))).astype(np.complex64)
    y = gl.istft(D, hop, win)
    assert y.dtype == np.float32 and len(y) == hop * (T - 1)
    w = gl.padded_window(n_fft, win)
    _, x = signal.istft(D / w.sum(), window=w, nperseg=n_fft, noverlap=n_fft - hop, input_onesided=True, boundary=False)
    x = x[n_fft // 2: n_fft // 2 + len(y)]
    assert np.abs(y - x).max() <= 1e-5 * np.abs(x).max()


def test_window_sumsquare_direct():
    wss = gl.window_sumsquare(9, HP.hop_length, HP.win_length, HP.n_fft)
    w2 = gl.padded_window(HP.n_fft, HP.win_length) ** 2
    ref = np.zeros(HP.n_fft + HP.hop_length * 8)
    for t in range(9):
        ref[t * HP.hop_length:t * HP.hop_length + HP.n_fft] += w2
    assert np.allclose(wss, ref, rtol=1e-6, atol=1e-7)
    assert (wss[HP.n_fft // 2:-HP.n_fft // 2] > 0.3).all()          # the kept region is always covered


def test_stft_istft_reconstructs():
    y = _signal(HP.hop_length * 40)
    D = gl.stft(y, HP.n_fft, HP.hop_length, HP.win_length)
    x = gl.istft(D, HP.hop_length, HP.win_length)
    assert len(x) == len(y)
    assert np.abs(x - y).max() < 2e-5


def test_deemphasis_matches_lfilter():
    x = _signal(5000, 5)
    ref = signal.lfilter([1], [1, -0.97], x)
    assert np.abs(gl.deemphasis(x, 0.97) - ref).max() < 1e-12


def test_griffin_lim_converges_and_matches_magnitude():
    y = _signal(HP.hop_length * 30, 7)
    S = np.abs(gl.stft(y, HP.n_fft, HP.hop_length, HP.win_length)).astype(np.float32)
    trace = []
    x = gl.griffin_lim(S, HP.n_fft, HP.hop_length, HP.win_length, 30, trace=trace)
    assert len(x) == len(y) and x.dtype == np.float32
    assert trace[0] > 2 * trace[-1]
    assert all(b <= a * 1.0005 for a, b in zip(trace[1:], trace[2:]))      # inconsistency is non-increasing
    S2 = np.abs(gl.stft(x, HP.n_fft, HP.hop_length, HP.win_length))
    assert np.linalg.norm(S2 - S) / np.linalg.norm(S) < 0.2


def test_spectrogram2wav_shapes_and_zero_iterations():
    rng = np.random.default_rng(11)
    mag = rng.uniform(-0.1, 1.1, (12, 1 + HP.n_fft // 2)).astype(np.float32)
    wav = gl.spectrogram2wav(HP, mag, n_iter=0)
    assert wav.dtype == np.float32 and wav.shape == (HP.hop_length * 11,)
    S = gl.amplitude_from_mag(HP, mag)
    assert S.shape == (1025, 12)
    lo = (10.0 ** ((0 * 100 - 100 + 20) * 0.05)) ** 1.5
    hi = (10.0 ** ((1 * 100 - 100 + 20) * 0.05)) ** 1.5
    assert np.isclose(S.min(), lo, rtol=1e-5) and np.isclose(S.max(), hi, rtol=1e-5)
    ref = signal.lfilter([1], [1, -0.97], gl.istft(S, HP.hop_length, HP.win_length)).astype(np.float32)
    assert np.array_equal(wav, ref)


@pytest.mark.parametrize("n_fft,hop,win", [(2048, 275, 1102), (1024, 256, 1024), (512, 128, 400)])
def test_stft_istft_match_torch(n_fft, hop, win):
    """A third independent implementation: torch.stft / torch.istft (CPU) have librosa's conventions built in -- the
    window of win_length centred in n_fft, centre = reflect padding by n_fft/2, inverse normalised by the window
    envelope (sum of squared shifted windows) and trimmed to hop*(frames-1).  librosa 0.6.2 itself is absent: the
    restatement stays 'parity unpinned' against it, this narrows what could be wrong."""
    torch = pytest.importorskip("torch")
    y = _signal(hop * 41, seed=5)
    wt = torch.from_numpy(gl.hann_periodic(win).astype(np.float64))
    D = gl.stft(y, n_fft, hop, win)
    Dt = torch.stft(torch.from_numpy(y.astype(np.float64)), n_fft=n_fft, hop_length=hop, win_length=win, window=wt,
                    center=True, pad_mode="reflect", normalized=False, onesided=True, return_complex=True).numpy()
    assert D.shape == Dt.shape
    assert np.abs(D - Dt).max() <= 2e-6 * np.abs(Dt).max()
    rng = np.random.default_rng(8)
    T = 29
    S = (rng.standard_normal((1 + n_fft // 2, T)) + 1j * rng.standard_normal((1 + n_fft // 2, T)))
    S[0].imag = 0; S[-1].imag = 0                                   # DC and Nyquist bins of a real signal's spectrum
    x = gl.istft(S.astype(np.complex64), hop, win)
    xt = torch.istft(torch.from_numpy(S), n_fft=n_fft, hop_length=hop, win_length=win, window=wt, center=True,
                     normalized=False, onesided=True, length=hop * (T - 1)).numpy()
    assert x.shape == xt.shape
    inner = slice(n_fft, len(x) - n_fft)                             # away from the ends, where the envelope rules can differ
    assert np.abs(x - xt)[inner].max() <= 2e-5 * np.abs(xt).max()
    assert np.abs(x - xt).max() <= 2e-5 * np.abs(xt).max(), "edges differ: window-envelope / tiny rules"
