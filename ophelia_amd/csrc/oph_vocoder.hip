// libophelia_vocoder.so -- batched Griffin-Lim on gfx950 (C ABI: include/ophelia_vocoder.h).
//
// Data layout in HBM (frame-major, all utterances of a batch concatenated; G = total frames):
//   S      [G][nbin]        f32   target amplitudes (mag de-normalised, 10^(x/20), ^power)
//   X      [G][nbin]        f32x2 current spectrum estimate (R2C output, phase-projected in place, C2R input)
//   tfr    [G][n_fft]       f32   inverse transforms of the frames (unscaled)
//   wfr    [G][n_fft]       f32   windowed analysis frames (zero outside the window support, written once)
//   y      [sum_b hop*(F_b-1)]    f32   time signal per utterance
// One Griffin-Lim iteration = C2R (hipFFT) -> gl_overlap_add -> gl_frame -> R2C (hipFFT) -> gl_phase.
// Every kernel is a streaming pass (HBM-bound); accumulation orders follow librosa 0.6.2 (frames in ascending order,
// float32), see oracle/griffin_lim_oracle.py for the restated algorithm and the reference lines.
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ophelia_vocoder.h"

namespace {

thread_local std::string g_create_error;

#define VCHECK(expr)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            v->err = std::string(#expr) + ": " + hipGetErrorString(e_);                          \
            return OPHV_ERR_DEVICE;                                                              \
        }                                                                                        \
    } while (0)
#define FCHECK(expr)                                                                             \
    do {                                                                                         \
        hipfftResult r_ = (expr);                                                                \
        if (r_ != HIPFFT_SUCCESS) {                                                              \
            v->err = std::string(#expr) + ": hipfft error " + std::to_string((int)r_);           \
            return OPHV_ERR_DEVICE;                                                              \
        }                                                                                        \
    } while (0)

enum { OPHV_OK = 0, OPHV_ERR_INVALID = -1, OPHV_ERR_STATE = -2, OPHV_ERR_DEVICE = -3, OPHV_ERR_UNSUPPORTED = -4 };

struct Tables {            // per-batch tables in device memory
    int* frame_utt = nullptr;     // [G]   utterance of global frame g
    int* foff = nullptr;          // [B+1] first global frame of utterance b
    long long* yoff = nullptr;    // [B+1] first sample of utterance b in y / wav
    long long* src_off = nullptr; // [B]   float offset of utterance b's first row in the source spectrogram
};

}  // namespace

struct oph_vocoder {
    oph_gl_params p{};
    int device = 0;
    int nbin = 0, lpad = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_ms = 0.f;
    std::string err;
    float *d_w = nullptr, *d_w2 = nullptr;          // padded window and its square, [n_fft]
    // capacity-managed buffers
    long long capG = 0, capY = 0, capStage = 0;
    int capB = 0;
    float *S = nullptr, *tfr = nullptr, *wfr = nullptr, *y = nullptr, *wav = nullptr, *stage = nullptr;
    float2* X = nullptr;
    Tables t;
    hipfftHandle plan_c2r = 0, plan_r2c = 0;
    long long planG = 0;
};

namespace {

// ---------------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------------

// utils.py:78-88: S = (10^((clip(mag,0,1)*max_db - max_db + ref_db)/20))^power ; X = S + 0i (copy.deepcopy start)
__global__ __launch_bounds__(256) void gl_prepare(const float* __restrict__ src, const long long* __restrict__ src_off,
                                                 const int* __restrict__ frame_utt, const int* __restrict__ foff,
                                                 float* __restrict__ S, float2* __restrict__ X, int nbin, float max_db,
                                                 float ref_db, float power, int denorm) {
    const int g = blockIdx.x;
    const int b = frame_utt[g];
    const float* row = src + src_off[b] + (long long)(g - foff[b]) * nbin;
    for (int k = threadIdx.x; k < nbin; k += 256) {
        float m = row[k];
        float s;
        if (denorm) {
            m = fminf(fmaxf(m, 0.f), 1.f) * max_db - max_db + ref_db;
            s = powf(powf(10.0f, m * 0.05f), power);
        } else {
            s = m;
        }
        S[(long long)g * nbin + k] = s;
        X[(long long)g * nbin + k] = make_float2(s, 0.f);
    }
}

// librosa.istft's overlap-add + window-sum-square normalisation + centre trim, as a gather: output sample n of
// utterance b (untrimmed position m = n + n_fft/2) sums w[i]*ifft_frame_f[i] over the frames whose window support
// covers m, in ascending frame order like the reference's loop.
__global__ __launch_bounds__(256) void gl_overlap_add(const float* __restrict__ tfr, const float* __restrict__ w,
                                                     const float* __restrict__ w2, const int* __restrict__ foff,
                                                     const long long* __restrict__ yoff, float* __restrict__ y,
                                                     int n_fft, int hop, int lpad, int win, float scale) {
    const int b = blockIdx.y;
    const int F = foff[b + 1] - foff[b];
    const long long len = (long long)hop * (F - 1);
    const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
    if (n >= len) return;
    const long long m = n + n_fft / 2;
    // support: lpad <= m - f*hop < lpad+win
    long long flo = (m - (lpad + win - 1) + hop - 1) / hop;
    if (m - (lpad + win - 1) < 0) flo = 0;
    long long fhi = (m - lpad) / hop;
    if (fhi > F - 1) fhi = F - 1;
    float acc = 0.f, ws = 0.f;
    const float* base = tfr + (long long)foff[b] * n_fft;
    for (long long f = flo; f <= fhi; ++f) {
        const int i = (int)(m - f * hop);
        acc += w[i] * (base[f * n_fft + i] * scale);
        ws += w2[i];
    }
    if (ws > 1.17549435e-38f) acc /= ws;
    y[yoff[b] + n] = acc;
}

// librosa.stft's reflect padding + framing + analysis window: wfr[g][i] = w[i] * ypad[f*hop + i] on the window support
// (the rest of each row was zeroed when the buffer was allocated and is never written).
__global__ __launch_bounds__(256) void gl_frame(const float* __restrict__ y, const float* __restrict__ w,
                                               const int* __restrict__ frame_utt, const int* __restrict__ foff,
                                               const long long* __restrict__ yoff, float* __restrict__ wfr, int n_fft,
                                               int hop, int lpad, int win) {
    const int g = blockIdx.x;
    const int b = frame_utt[g];
    const int f = g - foff[b];
    const int F = foff[b + 1] - foff[b];
    const long long len = (long long)hop * (F - 1);
    const long long period = 2 * (len - 1);
    const float* yb = y + yoff[b];
    float* row = wfr + (long long)g * n_fft;
    for (int i = lpad + threadIdx.x; i < lpad + win; i += 256) {
        long long j = (long long)f * hop + i - n_fft / 2;       // position in the unpadded signal
        if (j < 0 || j >= len) {                                // np.pad(mode='reflect'), repeated if needed
            if (len == 1) j = 0;
            else {
                j %= period;
                if (j < 0) j += period;
                if (j >= len) j = period - j;
            }
        }
        row[i] = w[i] * yb[j];
    }
}

// utils.py:104-105: phase = est / max(1e-8, |est|); X_best = spectrogram * phase
__global__ __launch_bounds__(256) void gl_phase(const float* __restrict__ S, float2* __restrict__ X, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float2 e = X[i];
    const float a = fmaxf(1e-8f, sqrtf(e.x * e.x + e.y * e.y));
    const float s = S[i];
    X[i] = make_float2(s * (e.x / a), s * (e.y / a));
}

// scipy.signal.lfilter([1],[1,-a]): y[n] = x[n] + a*y[n-1], float64 state.  One wave per utterance; each lane owns 16
// consecutive samples of a 1024-sample chunk, the linear recurrence is combined across lanes with a log-step scan.
__global__ __launch_bounds__(64) void gl_deemphasis(const float* __restrict__ x, const long long* __restrict__ yoff,
                                                   float* __restrict__ out, double a) {
    const int b = blockIdx.x;
    const long long off = yoff[b], len = yoff[b + 1] - yoff[b];
    const int lane = threadIdx.x;
    double apow[17];
    apow[0] = 1.0;
#pragma unroll
    for (int j = 1; j <= 16; ++j) apow[j] = apow[j - 1] * a;
    const double A16 = apow[16];
    const double Alane = pow(A16, (double)lane);
    double carry = 0.0;
    for (long long base = 0; base < len; base += 1024) {
        const long long s = base + lane * 16;
        double v[16];
        double prev = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const double xv = (s + j < len) ? (double)x[off + s + j] : 0.0;
            prev = xv + a * prev;
            v[j] = prev;
        }
        double sc = prev, fac = A16;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double tprev = __shfl_up(sc, o);
            if (lane >= o) sc += tprev * fac;
            fac *= fac;
        }
        double excl = __shfl_up(sc, 1);
        if (lane == 0) excl = 0.0;
        const double cin = excl + carry * Alane;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const double yv = v[j] + apow[j + 1] * cin;
            if (s + j < len) out[off + s + j] = (float)yv;
            if (j == 15) prev = yv;
        }
        carry = __shfl(prev, 63);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------

template <typename T>
int dev_alloc(oph_vocoder* v, T** p, long long n, bool zero) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    VCHECK(hipMalloc((void**)p, (size_t)n * sizeof(T)));
    if (zero) VCHECK(hipMemsetAsync(*p, 0, (size_t)n * sizeof(T), v->stream));
    return OPHV_OK;
}

struct Batch {
    int B = 0;
    long long G = 0, Y = 0;
    int maxF = 0;
    std::vector<int> foff, frame_utt;
    std::vector<long long> yoff;
};

int plan_batch(oph_vocoder* v, const int32_t* n_frames, int B, Batch& bt) {
    if (B <= 0 || !n_frames) { v->err = "empty batch"; return OPHV_ERR_INVALID; }
    bt.B = B;
    bt.foff.assign(B + 1, 0);
    bt.yoff.assign(B + 1, 0);
    for (int b = 0; b < B; ++b) {
        if (n_frames[b] < 2) {
            v->err = "n_frames[" + std::to_string(b) + "] = " + std::to_string(n_frames[b]) + ": at least 2 frames needed";
            return OPHV_ERR_INVALID;
        }
        bt.foff[b + 1] = bt.foff[b] + n_frames[b];
        bt.yoff[b + 1] = bt.yoff[b] + (long long)v->p.hop_length * (n_frames[b] - 1);
        if (n_frames[b] > bt.maxF) bt.maxF = n_frames[b];
    }
    bt.G = bt.foff[B];
    bt.Y = bt.yoff[B];
    bt.frame_utt.resize(bt.G);
    for (int b = 0; b < B; ++b)
        for (int g = bt.foff[b]; g < bt.foff[b + 1]; ++g) bt.frame_utt[g] = b;
    return OPHV_OK;
}

// grow-only buffers + FFT plans for G frames (rounded up so that nearby batch sizes share a plan)
int ensure_capacity(oph_vocoder* v, const Batch& bt) {
    const int n_fft = v->p.n_fft;
    const long long G = (bt.G + 255) / 256 * 256;
    if (G > v->capG) {
        int rc;
        if ((rc = dev_alloc(v, &v->S, G * v->nbin, false))) return rc;
        if ((rc = dev_alloc(v, &v->X, G * v->nbin, true))) return rc;
        if ((rc = dev_alloc(v, &v->tfr, G * n_fft, false))) return rc;
        if ((rc = dev_alloc(v, &v->wfr, G * n_fft, true))) return rc;
        if ((rc = dev_alloc(v, &v->t.frame_utt, G, true))) return rc;
        v->capG = G;
    }
    if (bt.Y > v->capY) {
        int rc;
        if ((rc = dev_alloc(v, &v->y, bt.Y, false))) return rc;
        if ((rc = dev_alloc(v, &v->wav, bt.Y, false))) return rc;
        v->capY = bt.Y;
    }
    if (bt.B > v->capB) {
        int rc;
        if ((rc = dev_alloc(v, &v->t.foff, bt.B + 1, false))) return rc;
        if ((rc = dev_alloc(v, &v->t.yoff, bt.B + 1, false))) return rc;
        if ((rc = dev_alloc(v, &v->t.src_off, bt.B, false))) return rc;
        v->capB = bt.B;
    }
    if (v->planG != v->capG) {
        if (v->plan_c2r) hipfftDestroy(v->plan_c2r);
        if (v->plan_r2c) hipfftDestroy(v->plan_r2c);
        v->plan_c2r = v->plan_r2c = 0;
        FCHECK(hipfftPlan1d(&v->plan_c2r, n_fft, HIPFFT_C2R, (int)v->capG));
        FCHECK(hipfftPlan1d(&v->plan_r2c, n_fft, HIPFFT_R2C, (int)v->capG));
        FCHECK(hipfftSetStream(v->plan_c2r, v->stream));
        FCHECK(hipfftSetStream(v->plan_r2c, v->stream));
        v->planG = v->capG;
    }
    return OPHV_OK;
}

int upload_tables(oph_vocoder* v, const Batch& bt, const std::vector<long long>& src_off) {
    VCHECK(hipMemcpyAsync(v->t.frame_utt, bt.frame_utt.data(), bt.G * sizeof(int), hipMemcpyHostToDevice, v->stream));
    VCHECK(hipMemcpyAsync(v->t.foff, bt.foff.data(), (bt.B + 1) * sizeof(int), hipMemcpyHostToDevice, v->stream));
    VCHECK(hipMemcpyAsync(v->t.yoff, bt.yoff.data(), (bt.B + 1) * sizeof(long long), hipMemcpyHostToDevice, v->stream));
    VCHECK(hipMemcpyAsync(v->t.src_off, src_off.data(), bt.B * sizeof(long long), hipMemcpyHostToDevice, v->stream));
    VCHECK(hipStreamSynchronize(v->stream));     // the host vectors go out of scope after the call
    return OPHV_OK;
}

int launch_istft(oph_vocoder* v, const Batch& bt) {
    const auto& p = v->p;
    FCHECK(hipfftExecC2R(v->plan_c2r, (hipfftComplex*)v->X, v->tfr));
    dim3 grid((unsigned)(((long long)p.hop_length * (bt.maxF - 1) + 255) / 256), bt.B);
    hipLaunchKernelGGL(gl_overlap_add, grid, dim3(256), 0, v->stream, v->tfr, v->d_w, v->d_w2, v->t.foff, v->t.yoff,
                       v->y, p.n_fft, p.hop_length, v->lpad, p.win_length, 1.0f / p.n_fft);
    return OPHV_OK;
}

int launch_stft(oph_vocoder* v, const Batch& bt) {
    const auto& p = v->p;
    hipLaunchKernelGGL(gl_frame, dim3((unsigned)bt.G), dim3(256), 0, v->stream, v->y, v->d_w, v->t.frame_utt, v->t.foff,
                       v->t.yoff, v->wfr, p.n_fft, p.hop_length, v->lpad, p.win_length);
    FCHECK(hipfftExecR2C(v->plan_r2c, v->wfr, (hipfftComplex*)v->X));
    return OPHV_OK;
}

// S/X prepared -> y (device).  utils.py:99-109
int run_griffin_lim(oph_vocoder* v, const Batch& bt, int n_iter) {
    int rc;
    const long long n = bt.G * v->nbin;
    for (int it = 0; it < n_iter; ++it) {
        if ((rc = launch_istft(v, bt))) return rc;
        if ((rc = launch_stft(v, bt))) return rc;
        hipLaunchKernelGGL(gl_phase, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, v->stream, v->S, v->X, n);
    }
    if ((rc = launch_istft(v, bt))) return rc;
    VCHECK(hipGetLastError());
    return OPHV_OK;
}

int run_pipeline(oph_vocoder* v, const float* d_src, const std::vector<long long>& src_off, const Batch& bt,
                 int denorm, int n_iter, bool deemph, float* wav_host) {
    int rc;
    if ((rc = upload_tables(v, bt, src_off))) return rc;
    VCHECK(hipEventRecord(v->ev0, v->stream));
    hipLaunchKernelGGL(gl_prepare, dim3((unsigned)bt.G), dim3(256), 0, v->stream, d_src, v->t.src_off, v->t.frame_utt,
                       v->t.foff, v->S, v->X, v->nbin, v->p.max_db, v->p.ref_db, v->p.power, denorm);
    if ((rc = run_griffin_lim(v, bt, n_iter))) return rc;
    const float* result = v->y;
    if (deemph) {
        hipLaunchKernelGGL(gl_deemphasis, dim3(bt.B), dim3(64), 0, v->stream, v->y, v->t.yoff, v->wav,
                           (double)v->p.preemphasis);
        result = v->wav;
    }
    VCHECK(hipEventRecord(v->ev1, v->stream));
    VCHECK(hipMemcpyAsync(wav_host, result, bt.Y * sizeof(float), hipMemcpyDeviceToHost, v->stream));
    VCHECK(hipStreamSynchronize(v->stream));
    VCHECK(hipEventElapsedTime(&v->last_ms, v->ev0, v->ev1));
    return OPHV_OK;
}

int ensure_stage(oph_vocoder* v, long long n) {
    if (n > v->capStage) {
        int rc;
        if ((rc = dev_alloc(v, &v->stage, n, false))) return rc;
        v->capStage = n;
    }
    return OPHV_OK;
}

int from_host(oph_vocoder* v, const float* host_rows, const int32_t* n_frames, int B, int denorm, int n_iter,
              bool deemph, float* wav) {
    if (!v) return OPHV_ERR_INVALID;
    if (!host_rows || !wav) { v->err = "null buffer"; return OPHV_ERR_INVALID; }
    VCHECK(hipSetDevice(v->device));
    Batch bt;
    int rc;
    if ((rc = plan_batch(v, n_frames, B, bt))) return rc;
    if ((rc = ensure_capacity(v, bt))) return rc;
    if ((rc = ensure_stage(v, bt.G * v->nbin))) return rc;
    VCHECK(hipMemcpyAsync(v->stage, host_rows, bt.G * v->nbin * sizeof(float), hipMemcpyHostToDevice, v->stream));
    std::vector<long long> src_off(B);
    for (int b = 0; b < B; ++b) src_off[b] = (long long)bt.foff[b] * v->nbin;
    return run_pipeline(v, v->stage, src_off, bt, denorm, n_iter, deemph, wav);
}

}  // namespace

extern "C" {

int oph_vocoder_abi_version(void) { return OPH_VOCODER_ABI_VERSION; }

const char* oph_vocoder_last_error(const oph_vocoder* v) { return v ? v->err.c_str() : g_create_error.c_str(); }

int oph_vocoder_create(const oph_gl_params* p, int device, oph_vocoder** out) {
    if (!p || !out) { g_create_error = "null argument"; return OPHV_ERR_INVALID; }
    *out = nullptr;
    if (p->n_fft < 2 || (p->n_fft & 1) || p->hop_length < 1 || p->win_length < 1 || p->win_length > p->n_fft ||
        p->hop_length > p->win_length || p->n_iter < 0) {
        g_create_error = "unsupported STFT geometry (need even n_fft >= win_length >= hop_length >= 1, n_iter >= 0)";
        return OPHV_ERR_UNSUPPORTED;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        g_create_error = "no usable HIP device " + std::to_string(device) + " (there is no CPU fallback)";
        return OPHV_ERR_DEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess || std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("device is not gfx950: ") + prop.gcnArchName;
        return OPHV_ERR_UNSUPPORTED;
    }
    oph_vocoder* v = new oph_vocoder();
    v->p = *p;
    v->device = device;
    v->nbin = p->n_fft / 2 + 1;
    v->lpad = (p->n_fft - p->win_length) / 2;          // librosa.util.pad_center
    auto fail = [&](const std::string& m) {
        g_create_error = m;
        oph_vocoder_destroy(v);
        return OPHV_ERR_DEVICE;
    };
    if (hipSetDevice(device) != hipSuccess) return fail("hipSetDevice failed");
    if (hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking) != hipSuccess) return fail("stream creation failed");
    if (hipEventCreate(&v->ev0) != hipSuccess || hipEventCreate(&v->ev1) != hipSuccess) return fail("event creation failed");
    // periodic Hann (scipy get_window('hann', win_length, fftbins=True)), centred in n_fft, evaluated in double
    std::vector<float> w(p->n_fft, 0.f), w2(p->n_fft, 0.f);
    for (int i = 0; i < p->win_length; ++i) {
        const double x = 0.5 - 0.5 * std::cos(2.0 * M_PI * i / p->win_length);
        w[v->lpad + i] = (float)x;
        w2[v->lpad + i] = (float)(x * x);
    }
    if (hipMalloc((void**)&v->d_w, p->n_fft * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&v->d_w2, p->n_fft * sizeof(float)) != hipSuccess ||
        hipMemcpy(v->d_w, w.data(), p->n_fft * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(v->d_w2, w2.data(), p->n_fft * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        return fail("window upload failed");
    *out = v;
    return OPHV_OK;
}

int oph_vocoder_destroy(oph_vocoder* v) {
    if (!v) return OPHV_OK;
    (void)hipSetDevice(v->device);
    if (v->stream) (void)hipStreamSynchronize(v->stream);
    if (v->plan_c2r) hipfftDestroy(v->plan_c2r);
    if (v->plan_r2c) hipfftDestroy(v->plan_r2c);
    void* bufs[] = {v->d_w, v->d_w2, v->S, v->X, v->tfr, v->wfr, v->y, v->wav, v->stage, v->t.frame_utt, v->t.foff,
                    v->t.yoff, v->t.src_off};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    if (v->ev0) (void)hipEventDestroy(v->ev0);
    if (v->ev1) (void)hipEventDestroy(v->ev1);
    if (v->stream) (void)hipStreamDestroy(v->stream);
    delete v;
    return OPHV_OK;
}

int oph_spectrogram2wav(oph_vocoder* v, const float* mag, const int32_t* n_frames, int B, float* wav) {
    return from_host(v, mag, n_frames, B, 1, v ? v->p.n_iter : 0, true, wav);
}

int oph_spectrogram2wav_device(oph_vocoder* v, const float* d_mag, int64_t utt_stride, const int32_t* n_frames, int B,
                               float* wav) {
    if (!v) return OPHV_ERR_INVALID;
    if (!d_mag || !wav) { v->err = "null buffer"; return OPHV_ERR_INVALID; }
    VCHECK(hipSetDevice(v->device));
    Batch bt;
    int rc;
    if ((rc = plan_batch(v, n_frames, B, bt))) return rc;
    for (int b = 0; b < B; ++b)
        if ((int64_t)n_frames[b] * v->nbin > utt_stride) { v->err = "n_frames exceeds utt_stride"; return OPHV_ERR_INVALID; }
    if ((rc = ensure_capacity(v, bt))) return rc;
    std::vector<long long> src_off(B);
    for (int b = 0; b < B; ++b) src_off[b] = (long long)b * utt_stride;
    return run_pipeline(v, d_mag, src_off, bt, 1, v->p.n_iter, true, wav);
}

int oph_vocoder_griffin_lim(oph_vocoder* v, const float* S, const int32_t* n_frames, int B, int n_iter, float* y) {
    return from_host(v, S, n_frames, B, 0, (v && n_iter < 0) ? v->p.n_iter : n_iter, false, y);
}

int oph_vocoder_stft(oph_vocoder* v, const float* y, int64_t len, float* D) {
    if (!v) return OPHV_ERR_INVALID;
    if (!y || !D || len < 1 || len % v->p.hop_length) {
        v->err = "stft: length must be a positive multiple of hop_length (as every signal istft returns is)";
        return OPHV_ERR_INVALID;
    }
    VCHECK(hipSetDevice(v->device));
    const int32_t F = (int32_t)(len / v->p.hop_length) + 1;
    Batch bt;
    int rc;
    if ((rc = plan_batch(v, &F, 1, bt))) return rc;
    if ((rc = ensure_capacity(v, bt))) return rc;
    if ((rc = upload_tables(v, bt, std::vector<long long>(1, 0)))) return rc;
    VCHECK(hipMemcpyAsync(v->y, y, len * sizeof(float), hipMemcpyHostToDevice, v->stream));
    if ((rc = launch_stft(v, bt))) return rc;
    VCHECK(hipMemcpyAsync(D, v->X, (size_t)F * v->nbin * sizeof(float2), hipMemcpyDeviceToHost, v->stream));
    VCHECK(hipStreamSynchronize(v->stream));
    return OPHV_OK;
}

int oph_vocoder_istft(oph_vocoder* v, const float* D, int n_frames, float* y) {
    if (!v) return OPHV_ERR_INVALID;
    if (!y || !D) { v->err = "null buffer"; return OPHV_ERR_INVALID; }
    VCHECK(hipSetDevice(v->device));
    const int32_t F = n_frames;
    Batch bt;
    int rc;
    if ((rc = plan_batch(v, &F, 1, bt))) return rc;
    if ((rc = ensure_capacity(v, bt))) return rc;
    if ((rc = upload_tables(v, bt, std::vector<long long>(1, 0)))) return rc;
    VCHECK(hipMemcpyAsync(v->X, D, (size_t)F * v->nbin * sizeof(float2), hipMemcpyHostToDevice, v->stream));
    if ((rc = launch_istft(v, bt))) return rc;
    VCHECK(hipMemcpyAsync(y, v->y, bt.Y * sizeof(float), hipMemcpyDeviceToHost, v->stream));
    VCHECK(hipStreamSynchronize(v->stream));
    return OPHV_OK;
}

int oph_vocoder_deemphasis(oph_vocoder* v, const float* x, int64_t len, float* y) {
    if (!v) return OPHV_ERR_INVALID;
    if (!x || !y || len < 1) { v->err = "deemphasis: bad buffer"; return OPHV_ERR_INVALID; }
    VCHECK(hipSetDevice(v->device));
    float *dx = nullptr, *dy = nullptr;
    long long* doff = nullptr;
    const long long off[2] = {0, (long long)len};
    VCHECK(hipMalloc((void**)&dx, len * sizeof(float)));
    VCHECK(hipMalloc((void**)&dy, len * sizeof(float)));
    VCHECK(hipMalloc((void**)&doff, sizeof(off)));
    VCHECK(hipMemcpyAsync(dx, x, len * sizeof(float), hipMemcpyHostToDevice, v->stream));
    VCHECK(hipMemcpyAsync(doff, off, sizeof(off), hipMemcpyHostToDevice, v->stream));
    hipLaunchKernelGGL(gl_deemphasis, dim3(1), dim3(64), 0, v->stream, dx, doff, dy, (double)v->p.preemphasis);
    VCHECK(hipMemcpyAsync(y, dy, len * sizeof(float), hipMemcpyDeviceToHost, v->stream));
    VCHECK(hipStreamSynchronize(v->stream));
    (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(doff);
    return OPHV_OK;
}

int oph_vocoder_last_device_ms(const oph_vocoder* v, float* ms) {
    if (!v || !ms) return OPHV_ERR_INVALID;
    *ms = v->last_ms;
    return OPHV_OK;
}

}  // extern "C"
