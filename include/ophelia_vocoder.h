/*
 * ophelia_vocoder.h -- C ABI of libophelia_vocoder.so: the Griffin-Lim vocoder step that follows the Text2Mel+SSRN
 * hot path, on MI355X (gfx950).  A separate library so that the hot-path library
 * (ophelia_hip.h) carries no FFT dependency; both share the HIP runtime, so device pointers obtained from
 * oph_device_mag() can be handed to oph_spectrogram2wav_device() without a round trip through host memory.
 *
 * Replaces, in the reference tree:
 *     utils.py:69-97    spectrogram2wav(hp, mag)
 *     utils.py:99-109   griffin_lim(hp, spectrogram)
 *     utils.py:111-116  invert_spectrogram  -> librosa.istft   (librosa==0.6.2, not vendored by the reference)
 *     utils.py:103      librosa.stft(X_t, n_fft, hop_length, win_length=win_length)
 *     synthesize.py:604-617  the per-utterance loop / ProcessPoolExecutor over synth_wave (batched here instead)
 *
 * Layout: spectrograms are frame-major exactly as SSRN emits them and as the reference's `mag` argument is:
 * (T, 1+n_fft/2) rows, C-contiguous (the reference transposes to (F,T) only because librosa wants that).
 * Complex data is interleaved (re, im) float32.  Batches are ragged: utterance b has n_frames[b] rows and
 * produces hop_length*(n_frames[b]-1) samples; rows and samples of consecutive utterances are concatenated.
 * No CPU fallback: oph_vocoder_create fails if no gfx950 device / hipFFT is usable.
 */
#ifndef OPHELIA_VOCODER_H
#define OPHELIA_VOCODER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OPH_VOCODER_ABI_VERSION 1

/* hp.n_fft, hp.hop_length, hp.win_length, hp.n_iter, hp.power, hp.preemphasis, hp.max_db, hp.ref_db
 * (config/lj_tutorial.cfg: 2048, 275, 1102, 50, 1.5, 0.97, 100, 20) */
typedef struct oph_gl_params {
    int32_t n_fft;
    int32_t hop_length;
    int32_t win_length;
    int32_t n_iter;
    double power;
    double preemphasis;       /* double: the reference filters with the Python float 0.97, not its float32 rounding */
    double max_db;
    double ref_db;
} oph_gl_params;

typedef struct oph_vocoder oph_vocoder;

int oph_vocoder_abi_version(void);
int oph_vocoder_create(const oph_gl_params* p, int device, oph_vocoder** out);
int oph_vocoder_destroy(oph_vocoder* v);
const char* oph_vocoder_last_error(const oph_vocoder* v);   /* v may be NULL: last create error */

/* spectrogram2wav (utils.py:69-97, trim_output=False) for a ragged batch.
 *   mag      host, concatenated (sum_b n_frames[b], 1+n_fft/2) normalised dB magnitudes in [0,1] (clipped like the
 *            reference does)
 *   wav      host, concatenated float32 samples, hop_length*(n_frames[b]-1) per utterance
 * n_frames[b] >= 2 required (librosa fails on shorter input as well).                                             */
int oph_spectrogram2wav(oph_vocoder* v, const float* mag, const int32_t* n_frames, int B, float* wav);

/* Same, reading the magnitudes from device memory: utterance b's rows start at d_mag + b*utt_stride floats
 * (e.g. the SSRN output of libophelia_hip via oph_device_mag, trimmed by n_frames[b] = t_ends[b]*r as in
 * synthesize.py:607).  The caller guarantees d_mag is complete (oph_synchronize) before the call.             */
int oph_spectrogram2wav_device(oph_vocoder* v, const float* d_mag, int64_t utt_stride, const int32_t* n_frames,
                               int B, float* wav);

/* Stages of the above, for unit parity and measurement (host buffers):
 * oph_vocoder_griffin_lim  utils.py:99-109 on linear amplitudes S (rows, 1+n_fft/2); n_iter < 0 = the configured one
 * oph_vocoder_stft         librosa.stft(y, n_fft, hop, win_length) -> D (1+len/hop, 1+n_fft/2) complex interleaved
 * oph_vocoder_istft        librosa.istft(D, hop, win_length)       -> y of hop*(n_frames-1) samples
 * oph_vocoder_deemphasis   scipy.signal.lfilter([1],[1,-preemphasis], x)                                        */
int oph_vocoder_griffin_lim(oph_vocoder* v, const float* S, const int32_t* n_frames, int B, int n_iter, float* y);
int oph_vocoder_stft(oph_vocoder* v, const float* y, int64_t len, float* D);
int oph_vocoder_istft(oph_vocoder* v, const float* D, int n_frames, float* y);
int oph_vocoder_deemphasis(oph_vocoder* v, const float* x, int64_t len, float* y);

/* Implementation choice for griffin_lim / spectrogram2wav: 0 (default) = the fused in-LDS iteration kernel when
 * n_fft == 2048 (every shipped config but one), else the generic path; 1 = always the generic path (hipFFT batched
 * transforms + streaming kernels).  Both implement the same arithmetic; env OPH_VOCODER_BACKEND sets the initial value. */
int oph_vocoder_set_backend(oph_vocoder* v, int backend);

/* Device time (ms, HIP events on the vocoder's stream) of the last spectrogram2wav / griffin_lim call, excluding the
 * host<->device copies. */
int oph_vocoder_last_device_ms(const oph_vocoder* v, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* OPHELIA_VOCODER_H */
