import sys, time, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from types import SimpleNamespace
from ophelia_amd.vocoder import Vocoder
HP = SimpleNamespace(n_fft=2048, hop_length=275, win_length=1102, power=1.5, n_iter=50, preemphasis=0.97, max_db=100, ref_db=20, sr=22050)
v = Vocoder(HP, 0)
rng = np.random.default_rng(0)
B, T = 16, 800
mags = [rng.uniform(0, 1, (T, 1025)).astype(np.float32) for _ in range(B)]
for i in range(6):
    v.set_backend(int(os.environ.get("VB", "0")) if i >= 3 else 1)
    t = time.perf_counter(); out = v.spectrogram2wav_batch(mags); dt = time.perf_counter() - t
    print("wall %.1f ms  device %.2f ms" % (dt * 1e3, v.last_device_ms()))
