"""
CPU oracle for the Griffin-Lim vocoder step.  TEST INFRASTRUCTURE ONLY: imported by tests/ and
the benchmark's CPU leg, never by the product path (ophelia_amd/vocoder.py calls libophelia_vocoder.so and fails loudly
without it).

What it restates
  reference utils.py:69-97   spectrogram2wav   (de-normalise, 10**(x/20), **power, griffin_lim, de-pre-emphasis)
  reference utils.py:99-109  griffin_lim       (n_iter x { istft, stft, phase = est/max(1e-8,|est|) }, final istft)
  reference utils.py:111-116 invert_spectrogram = librosa.istft(S, hop_length, win_length=win_length, window="hann")

The STFT pair itself lives in a third-party dependency that is not vendored in the reference tree:
librosa==0.6.2 (reference requirements.txt) with scipy==1.1.0 for get_window / lfilter.  Its published algorithm
(librosa/core/spectrum.py, v0.6.2) is restated here:

  stft(y, n_fft, hop, win_length, window='hann', center=True, pad_mode='reflect')
      w = pad_center(get_window('hann', win_length, fftbins=True), n_fft)      # periodic Hann, zero-padded both sides
      y = np.pad(y, n_fft//2, mode='reflect');  n_frames = 1 + (len(y) - n_fft)//hop
      D[:, t] = fft(w * y[t*hop : t*hop+n_fft])[:1+n_fft//2]                   -> complex64
  istft(D, hop, win_length, window='hann', center=True)
      y = zeros(n_fft + hop*(n_frames-1), float32)
      for t: y[t*hop : t*hop+n_fft] += w * ifft(hermitian_extend(D[:, t])).real   # in frame order, float32 accumulate
      wss = window_sumsquare(...)  = sum_t pad_center(get_window(..)**2, n_fft) shifted by t*hop   (float32)
      y[wss > tiny(float32)] /= wss[...]
      return y[n_fft//2 : -n_fft//2]

PARITY UNPINNED: neither librosa nor the reference's Python-2 environment is available offline, so this restatement
cannot be checked against outputs of the real thing.  What pins it instead (tests/test_vocoder_oracle.py): the framing
and transforms agree with scipy.signal.stft / istft (an independent implementation of the same definitions) to float32
round-off, stft->istft reconstructs band-limited signals, and the Griffin-Lim loop reduces the spectral
inconsistency monotonically.
"""
from __future__ import annotations

import numpy as np

TINY_F32 = np.finfo(np.float32).tiny


def hann_periodic(win_length: int) -> np.ndarray:
    """scipy.signal.get_window('hann', win_length, fftbins=True) (float64)."""
    n = np.arange(win_length, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)


def pad_center(w: np.ndarray, size: int) -> np.ndarray:
    """librosa.util.pad_center: lpad = (size - n)//2, zeros either side."""
    n = len(w)
    lpad = (size - n) // 2
    out = np.zeros(size, w.dtype)
    out[lpad:lpad + n] = w
    return out


def padded_window(n_fft: int, win_length: int) -> np.ndarray:
    return pad_center(hann_periodic(win_length), n_fft)


def reflect_index(j: np.ndarray, n: int) -> np.ndarray:
    """index into a length-n signal of np.pad(..., mode='reflect') position j (may be <0 or >=n; repeated reflection)."""
    if n == 1:
        return np.zeros_like(j)
    period = 2 * (n - 1)
    j = np.mod(j, period)
    return np.where(j >= n, period - j, j)


def stft(y: np.ndarray, n_fft: int, hop_length: int, win_length: int) -> np.ndarray:
    """librosa.stft(y, n_fft, hop_length, win_length=win_length)  ->  (1+n_fft//2, n_frames) complex64."""
    y = np.asarray(y)
    w = padded_window(n_fft, win_length)
    half = n_fft // 2
    idx = reflect_index(np.arange(-half, len(y) + half), len(y))
    ypad = y[idx]
    n_frames = 1 + (len(ypad) - n_fft) // hop_length
    frames = np.stack([ypad[t * hop_length:t * hop_length + n_fft] for t in range(n_frames)], axis=1)  # (n_fft, n_frames)
    D = np.fft.rfft(w[:, None] * frames, axis=0)
    return D.astype(np.complex64)


def window_sumsquare(n_frames: int, hop_length: int, win_length: int, n_fft: int) -> np.ndarray:
    """librosa.filters.window_sumsquare('hann', n_frames, hop, win_length, n_fft, dtype=float32, norm=None)."""
    n = n_fft + hop_length * (n_frames - 1)
    x = np.zeros(n, np.float32)
    win_sq = pad_center(hann_periodic(win_length) ** 2, n_fft)
    for t in range(n_frames):
        s = t * hop_length
        x[s:min(n, s + n_fft)] += win_sq[:max(0, min(n_fft, n - s))].astype(np.float32)
    return x


def istft(D: np.ndarray, hop_length: int, win_length: int) -> np.ndarray:
    """librosa.istft(D, hop_length, win_length=win_length, window='hann')  ->  float32 of length hop*(n_frames-1)."""
    n_fft = 2 * (D.shape[0] - 1)
    n_frames = D.shape[1]
    w = padded_window(n_fft, win_length)
    y = np.zeros(n_fft + hop_length * (n_frames - 1), np.float32)
    # ifft(hermitian extension).real == irfft (the imaginary parts of the DC and Nyquist bins drop out in both)
    t_frames = np.fft.irfft(D.astype(np.complex64), n=n_fft, axis=0).astype(np.float32)
    for t in range(n_frames):
        s = t * hop_length
        y[s:s + n_fft] = y[s:s + n_fft] + w * t_frames[:, t]
    wss = window_sumsquare(n_frames, hop_length, win_length, n_fft)
    nz = wss > TINY_F32
    y[nz] /= wss[nz]
    half = n_fft // 2
    return y[half:-half]


def griffin_lim(spectrogram: np.ndarray, n_fft: int, hop_length: int, win_length: int, n_iter: int,
                trace: list | None = None) -> np.ndarray:
    """utils.py:99-109.  spectrogram: (1+n_fft//2, T) magnitudes.  `trace` collects the spectral inconsistency
    || |stft(istft(X))| - S ||_F / ||S||_F per iteration (diagnostic, not part of the reference)."""
    X_best = spectrogram.copy()
    for _ in range(n_iter):
        X_t = istft(X_best, hop_length, win_length)
        est = stft(X_t, n_fft, hop_length, win_length)
        if trace is not None:
            trace.append(float(np.linalg.norm(np.abs(est) - spectrogram) / max(np.linalg.norm(spectrogram), 1e-30)))
        phase = est / np.maximum(1e-8, np.abs(est))
        X_best = spectrogram * phase
    X_t = istft(X_best, hop_length, win_length)
    return np.real(X_t)


def deemphasis(x: np.ndarray, preemphasis: float) -> np.ndarray:
    """scipy.signal.lfilter([1], [1, -preemphasis], x): y[n] = x[n] + preemphasis*y[n-1] in float64."""
    y = np.empty(len(x), np.float64)
    acc = 0.0
    xs = np.asarray(x, np.float64)
    for i in range(len(xs)):
        acc = xs[i] + preemphasis * acc
        y[i] = acc
    return y


def amplitude_from_mag(hp, mag: np.ndarray) -> np.ndarray:
    """utils.py:78-88: (T, F) normalised dB magnitudes -> (F, T) linear amplitudes ** power."""
    m = np.asarray(mag, np.float32).T
    m = (np.clip(m, 0, 1) * np.float32(hp.max_db)) - np.float32(hp.max_db) + np.float32(hp.ref_db)
    m = np.power(np.float32(10.0), m * np.float32(0.05))
    return (m ** np.float32(hp.power)).astype(np.float32)


def spectrogram2wav(hp, mag: np.ndarray, n_iter: int | None = None) -> np.ndarray:
    """utils.py:69-97 with trim_output=False.  mag: (T, 1+n_fft//2) in [0,1]  ->  float32 wav of hop*(T-1) samples."""
    from scipy import signal
    S = amplitude_from_mag(hp, mag)
    wav = griffin_lim(S, hp.n_fft, hp.hop_length, hp.win_length, hp.n_iter if n_iter is None else n_iter)
    wav = signal.lfilter([1], [1, -hp.preemphasis], wav)
    return wav.astype(np.float32)
