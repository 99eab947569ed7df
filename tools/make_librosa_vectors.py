#!/usr/bin/env python
"""A Griffin-Lim pin produced by librosa 0.6.2 (the reference's requirements.txt) for a maintainer who has it:

    python tools/make_librosa_vectors.py        # writes tests/golden/librosa_vectors.npz

The vocoder oracle (oracle/griffin_lim_oracle.py) restates librosa 0.6.2's stft / istft from their documentation and is checked
against scipy / torch only ("parity unpinned"); tests/test_librosa_vectors.py compares it -- and with -m gpu
the HIP vocoder -- with what librosa computed, and skips until the file exists.  The driver below is written from the call
signatures of utils.spectrogram2wav / griffin_lim / invert_spectrogram (utils.py:69-116); nothing of the reference is imported."""
import copy
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    try:
        import librosa
        from scipy import signal
    except ImportError:
        sys.exit("this script needs librosa (the reference pins librosa==0.6.2) and scipy")
    p = dict(n_fft=2048, hop_length=275, win_length=1102, n_iter=50, power=1.5, preemphasis=0.97, max_db=100.0, ref_db=20.0)   # lj_tutorial.cfg:53-66
    rng = np.random.RandomState(5)
    out = {"librosa_version": np.array(librosa.__version__), "params": np.array([p[k] for k in sorted(p)], np.float64), "param_names": np.array(sorted(p))}
    # stft / istft of a seeded signal
    y = (0.1 * rng.randn(275 * 40)).astype(np.float32)
    D = librosa.stft(y, p["n_fft"], p["hop_length"], win_length=p["win_length"])
    out.update(stft_y=y, stft_D=D.astype(np.complex64), istft_y=librosa.istft(D, p["hop_length"], win_length=p["win_length"], window="hann").astype(np.float32))
    # spectrogram2wav on a smooth random magnitude in [0, 1] (frames, 1 + n_fft / 2)
    T = 60
    mag = np.clip(0.35 + 0.25 * np.cumsum(rng.randn(T, 1025), 0) / np.sqrt(np.arange(1, T + 1))[:, None], 0, 1).astype(np.float32)
    for n_iter in (1, 3, 50):
        m = mag.T
        m = (np.clip(m, 0, 1) * p["max_db"]) - p["max_db"] + p["ref_db"]
        m = np.power(10.0, m * 0.05)
        spec = m ** p["power"]
        X_best = copy.deepcopy(spec)
        for _ in range(n_iter):
            X_t = librosa.istft(X_best, p["hop_length"], win_length=p["win_length"], window="hann")
            est = librosa.stft(X_t, p["n_fft"], p["hop_length"], win_length=p["win_length"])
            phase = est / np.maximum(1e-8, np.abs(est))
            X_best = spec * phase
        X_t = librosa.istft(X_best, p["hop_length"], win_length=p["win_length"], window="hann")
        wav = signal.lfilter([1], [1, -p["preemphasis"]], np.real(X_t))
        out["gl_wav_%d" % n_iter] = wav.astype(np.float32)
    out["gl_mag"] = mag
    np.savez_compressed(os.path.join(GOLD, "librosa_vectors.npz"), **out)
    print("wrote", os.path.join(GOLD, "librosa_vectors.npz"))


if __name__ == "__main__":
    main()
