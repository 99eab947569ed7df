"""Griffin-Lim vocoder on the GPU: host-side mirror of the reference's utils.py:69-116 over libophelia_vocoder.so
(C ABI: include/ophelia_vocoder.h).  No CPU fallback: everything here calls the HIP library and raises when it is
missing.

  Vocoder(hp)                      one handle per GPU / thread; batches ragged utterances through one FFT plan
  spectrogram2wav(hp, mag)         same signature and result as utils.spectrogram2wav (trim_output=False)
  write_wav(path, wav, sr)         what soundfile.write(outfile, wav, hp.sr) produces: 16-bit PCM WAV
"""
import ctypes as C
import wave

import numpy as np

from . import _lib


class Vocoder(object):
    def __init__(self, hp, device=0):
        self.lib = _lib.load_vocoder()
        self.hp = hp
        self.params = _lib.OphGLParams(int(hp.n_fft), int(hp.hop_length), int(hp.win_length), int(hp.n_iter),
                                       float(hp.power), float(hp.preemphasis), float(hp.max_db), float(hp.ref_db))
        self.nbin = int(hp.n_fft) // 2 + 1
        self.hop = int(hp.hop_length)
        self._h = C.c_void_p()
        rc = self.lib.oph_vocoder_create(C.byref(self.params), int(device), C.byref(self._h))
        if rc != 0:
            raise _lib.OpheliaHipError("oph_vocoder_create failed (%d): %s"
                                       % (rc, self.lib.oph_vocoder_last_error(None).decode()))

    def _chk(self, rc):
        if rc != 0:
            raise _lib.OpheliaHipError("libophelia_vocoder error %d: %s"
                                       % (rc, self.lib.oph_vocoder_last_error(self._h).decode()))

    def close(self):
        if self._h:
            self.lib.oph_vocoder_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- batched entry points -------------------------------------------------------------------------------
    def _pack(self, rows_list):
        rows_list = [np.ascontiguousarray(m, np.float32) for m in rows_list]
        for m in rows_list:
            if m.ndim != 2 or m.shape[1] != self.nbin:
                raise ValueError("expected (T, %d) spectrograms, got %s" % (self.nbin, (m.shape,)))
        n_frames = np.array([m.shape[0] for m in rows_list], np.int32)
        packed = np.ascontiguousarray(np.concatenate(rows_list, axis=0))
        return packed, n_frames

    def _split(self, flat, n_frames):
        cuts = np.cumsum([self.hop * (int(f) - 1) for f in n_frames])[:-1]
        return [np.array(a) for a in np.split(flat, cuts)]

    def spectrogram2wav_batch(self, mags):
        """[ (T_b, 1+n_fft//2) normalised magnitudes ] -> [ float32 wav of hop*(T_b-1) samples ]  (utils.py:69-97)."""
        packed, n_frames = self._pack(mags)
        wav = np.empty(int(sum(self.hop * (int(f) - 1) for f in n_frames)), np.float32)
        self._chk(self.lib.oph_spectrogram2wav(self._h, _lib.fptr(packed), _lib.iptr(n_frames), len(n_frames),
                                               _lib.fptr(wav)))
        return self._split(wav, n_frames)

    def spectrogram2wav(self, mag):
        return self.spectrogram2wav_batch([mag])[0]

    def spectrogram2wav_from_engine(self, engine, n_frames):
        """Vocode the SSRN output still resident in `engine`'s HBM (no host copy of Z); n_frames[b] = t_ends[b]*hp.r
        is the trim of synthesize.py:607."""
        d_mag, stride, B = C.c_void_p(), C.c_int64(), C.c_int32()
        engine._chk(engine.lib.oph_device_mag(engine._h, C.byref(d_mag), C.byref(stride), C.byref(B)))
        n_frames = np.ascontiguousarray(n_frames, np.int32)
        if len(n_frames) != B.value:
            raise ValueError("n_frames has %d entries, the staged batch %d" % (len(n_frames), B.value))
        wav = np.empty(int(sum(self.hop * (int(f) - 1) for f in n_frames)), np.float32)
        self._chk(self.lib.oph_spectrogram2wav_device(self._h, d_mag, stride.value, _lib.iptr(n_frames), len(n_frames),
                                                      _lib.fptr(wav)))
        return self._split(wav, n_frames)

    def griffin_lim_batch(self, specs, n_iter=None):
        """utils.py:99-109 on linear amplitudes, frame-major [(T_b, 1+n_fft//2)] (the transpose of the reference's
        (F, T) argument)."""
        packed, n_frames = self._pack(specs)
        y = np.empty(int(sum(self.hop * (int(f) - 1) for f in n_frames)), np.float32)
        self._chk(self.lib.oph_vocoder_griffin_lim(self._h, _lib.fptr(packed), _lib.iptr(n_frames), len(n_frames),
                                                   -1 if n_iter is None else int(n_iter), _lib.fptr(y)))
        return self._split(y, n_frames)

    # ---- stages (unit parity) -------------------------------------------------------------------------------
    def stft(self, y):
        """librosa.stft(y, n_fft, hop, win_length) transposed: (1 + len//hop, 1+n_fft//2) complex64."""
        y = np.ascontiguousarray(y, np.float32)
        D = np.empty((len(y) // self.hop + 1, self.nbin), np.complex64)
        self._chk(self.lib.oph_vocoder_stft(self._h, _lib.fptr(y), len(y), D.ctypes.data_as(_lib.c_f32p)))
        return D

    def istft(self, D):
        """librosa.istft(D.T, hop, win_length) for frame-major D (T, 1+n_fft//2)."""
        D = np.ascontiguousarray(D, np.complex64)
        y = np.empty(self.hop * (D.shape[0] - 1), np.float32)
        self._chk(self.lib.oph_vocoder_istft(self._h, D.ctypes.data_as(_lib.c_f32p), D.shape[0], _lib.fptr(y)))
        return y

    def deemphasis(self, x):
        x = np.ascontiguousarray(x, np.float32)
        y = np.empty_like(x)
        self._chk(self.lib.oph_vocoder_deemphasis(self._h, _lib.fptr(x), len(x), _lib.fptr(y)))
        return y

    def set_backend(self, backend):
        """0 = fused in-LDS Griffin-Lim kernel when n_fft == 2048 (default), 1 = generic hipFFT path"""
        self._chk(self.lib.oph_vocoder_set_backend(self._h, int(backend)))

    def last_device_ms(self):
        ms = C.c_float()
        self._chk(self.lib.oph_vocoder_last_device_ms(self._h, C.byref(ms)))
        return ms.value


_cached = {}


def _vocoder_for(hp, device=0):
    key = (int(hp.n_fft), int(hp.hop_length), int(hp.win_length), int(hp.n_iter), float(hp.power),
           float(hp.preemphasis), float(hp.max_db), float(hp.ref_db), int(device))
    if key not in _cached:
        _cached[key] = Vocoder(hp, device)
    return _cached[key]


def spectrogram2wav(hp, mag, trim_output=False, device=0):
    """utils.py:69-97.  mag: (T, 1+n_fft//2) -> 1-D float32 wav."""
    if trim_output:
        raise NotImplementedError("trim_output=True (librosa.effects.trim) is outside the supported path; "
                                  "the reference leaves it off since generation stops early (utils.py:93)")
    return _vocoder_for(hp, device).spectrogram2wav(mag)


def write_wav(path, wav, sr):
    """soundfile.write(path, float32 wav, sr) -> RIFF/WAVE 16-bit PCM (libsndfile's default subtype for .wav);
    samples are scaled by 0x7FFF and rounded like libsndfile does, and clipped instead of wrapping."""
    pcm = np.clip(np.rint(np.asarray(wav, np.float64) * 32767.0), -32768, 32767).astype("<i2")
    with wave.open(path, "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(int(sr))
        f.writeframes(pcm.tobytes())
