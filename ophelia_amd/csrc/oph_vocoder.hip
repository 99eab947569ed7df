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

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
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
    int4* meta = nullptr;         // [G]   {frame index in its utterance, frames of the utterance, first global frame,
                                  //        first sample of the utterance in y}: one 16-byte load per frame
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
    long long capG = 0, capGg = 0, capY = 0, capStage = 0;    // capGg: frames the generic backend's buffers hold
    int capB = 0;
    float *S = nullptr, *tfr = nullptr, *wfr = nullptr, *y = nullptr, *wav = nullptr, *stage = nullptr;
    float2* X = nullptr;
    Tables t;
    hipfftHandle plan_c2r = 0, plan_r2c = 0;
    long long planG = 0;
    // fused path (n_fft == 2048): windowed inverse-transform segments, ping-pong, [G][wstride]; window-sum-square [Y]
    int backend = 0;                 // 0 = auto (fused when n_fft == 2048), 1 = hipFFT path
    int wstride = 0;
    long long capGF = 0, capYF = 0;
    float *wseg[2] = {nullptr, nullptr}, *wss = nullptr;
    float2* d_tw = nullptr;          // W_2048^k, k < 1024
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
        if (X) X[(long long)g * nbin + k] = make_float2(s, 0.f);
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

// scipy.signal.lfilter([1],[1,-a]): y[n] = x[n] + a*y[n-1], float64 state.  One wave per (utterance, span of `span`
// samples); each lane owns 16 consecutive samples of a 1024-sample chunk and the linear recurrence is combined across
// lanes with a log-step scan.  A span does not wait for its predecessor: it re-runs the recurrence over the `warm`
// samples before it from a zero state, whose missing history has decayed below a^warm <= 1e-17 of the signal, i.e.
// under the float64 round-off of the sequential filter (warm >= len reproduces the sequential order exactly).
__global__ __launch_bounds__(64) void gl_deemphasis(const float* __restrict__ x, const long long* __restrict__ yoff,
                                                   float* __restrict__ out, double a, int span, int warm) {
    const int b = blockIdx.y;
    const long long off = yoff[b], len = yoff[b + 1] - yoff[b];
    const long long first = (long long)blockIdx.x * span;            // first sample this wave stores
    if (first >= len) return;
    const long long last = min(len, first + span);
    long long begin = first - warm;
    if (begin < 0) begin = 0;
    begin &= ~1023ll;                                                // chunks stay aligned for every span
    const int lane = threadIdx.x;
    double apow[17];
    apow[0] = 1.0;
#pragma unroll
    for (int j = 1; j <= 16; ++j) apow[j] = apow[j - 1] * a;
    const double A16 = apow[16];
    const double Alane = pow(A16, (double)lane);
    double carry = 0.0;
    for (long long base = begin; base < last; base += 1024) {
        const long long s = base + lane * 16;
        double v[16];
        double prev = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const double xv = (s + j < last) ? (double)x[off + s + j] : 0.0;
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
            if (s + j >= first && s + j < last) out[off + s + j] = (float)yv;
            if (j == 15) prev = yv;
        }
        carry = __shfl(prev, 63);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// fused Griffin-Lim iteration for n_fft = 2048: one workgroup owns one frame at a time and keeps it in LDS through
//   gather(overlap-add of the neighbours' windowed segments, /wss, reflect pad, window) -> real FFT -> phase
//   projection against S -> inverse real FFT -> window -> store this frame's windowed segment.
// The only HBM traffic per frame and iteration is S (4.1 KB), the signal under the window (4.4 KB) and the 4.4 KB
// segment written.  Each thread owns eight consecutive samples of the frame (window values in registers).  The real transforms run as a 1024-point complex radix-4 Stockham FFT (5 passes,
// one butterfly per thread, LDS ping-pong) with the even/odd split folded, together with the phase projection and the
// inverse transform's split, into one in-place pass over bin pairs (k, 1024-k).
// ---------------------------------------------------------------------------------------------------------------
constexpr int FN = 2048, FH = 1024;

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
// LDS index swizzle of the two transform buffers: with it every ds_read_b64 / ds_write_b64 of the five radix-4 passes
// is bank-conflict free (checked exhaustively against the CDNA4 lane-group / bank rules); a bijection inside each
// block of 16 complex values, so the float view of a buffer stays pairwise contiguous.
__device__ __forceinline__ int PX(int i) { return i ^ (((i >> 4) * 5) & 15); }

// Twiddle table in LDS (float2 entries), one compact run per pass so that lanes read consecutive entries:
//   pass Ns (4,16,64,256): entry (r-1)*Ns + k = W_2048^(k*r*512/Ns), r = 1..3          offsets 0, 12, 60, 252
//   pair pass:             W_2048^k, k = 0..256                                        offset 1020
constexpr int TW_PAIR = 1020, TW_TOTAL = 1280;
__host__ __device__ constexpr int tw_off(int Ns) { return Ns == 4 ? 0 : Ns == 16 ? 12 : Ns == 64 ? 60 : 252; }

// Per-thread constants of the transform passes, computed once per workgroup lifetime and kept PACKED (two 16-bit LDS
// slot numbers per register): the kernel is VALU-issue bound, and re-deriving the swizzled slots in every pass of every
// frame was 40 % of its instructions.
struct FusedConst {
    int rd;                  // PX(tid); reads are at rd + 256 r in every pass
    unsigned wr[5][2];       // pass p: slots PX(j0 + r Ns) for r = (0,1) and (2,3)
    unsigned pk[2];          // pair pass: PX(k) | PX(1024-k) << 16 for k = tid+1 and k = tid+257
    int own;                 // slot of logical element 4 tid; logical 4 tid + r lives at own ^ r
};

__device__ __forceinline__ void fused_const_init(FusedConst& fc, int tid) {
    fc.rd = PX(tid);
#pragma unroll
    for (int p = 0; p < 5; ++p) {
        const int Ns = 1 << (2 * p);
        const int k = tid & (Ns - 1);
        const int j0 = ((tid - k) << 2) + k;
        fc.wr[p][0] = (unsigned)PX(j0) | ((unsigned)PX(j0 + Ns) << 16);
        fc.wr[p][1] = (unsigned)PX(j0 + 2 * Ns) | ((unsigned)PX(j0 + 3 * Ns) << 16);
    }
    fc.pk[0] = (unsigned)PX(tid + 1) | ((unsigned)PX(FH - 1 - tid) << 16);
    fc.pk[1] = (unsigned)PX(tid + 257) | ((unsigned)PX(FH - 257 - tid) << 16);
    fc.own = PX(4 * tid);
}

// a * b, or a * conj(b)
template <bool CONJ>
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) {
    return CONJ ? make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y)
                : make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

template <bool INV, int P>      // pass P: Ns = 4^P
__device__ __forceinline__ void fft_pass(const float2* __restrict__ in, float2* __restrict__ out, const float2* sT,
                                         const FusedConst& fc, int tid) {
    constexpr int Ns = 1 << (2 * P);
    float2 v0 = in[fc.rd], v1 = in[fc.rd + 256], v2 = in[fc.rd + 512], v3 = in[fc.rd + 768];
    if (P > 0) {
        const float2* tp = sT + tw_off(Ns) + (tid & (Ns - 1));
        v1 = cmulc<INV>(v1, tp[0]); v2 = cmulc<INV>(v2, tp[Ns]); v3 = cmulc<INV>(v3, tp[2 * Ns]);
    }
    const float2 a0 = cadd(v0, v2), a1 = csub(v0, v2), a2 = cadd(v1, v3), d = csub(v1, v3);
    const float2 a3 = INV ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x);
    out[fc.wr[P][0] & 0xffffu] = cadd(a0, a2);
    out[fc.wr[P][0] >> 16] = cadd(a1, a3);
    out[fc.wr[P][1] & 0xffffu] = csub(a0, a2);
    out[fc.wr[P][1] >> 16] = csub(a1, a3);
}

__device__ __forceinline__ float2 project(float2 X, float s) {      // utils.py:104-105
    const float a = fmaxf(1e-8f, sqrtf(X.x * X.x + X.y * X.y));
    return make_float2(s * (X.x / a), s * (X.y / a));
}

// bins k and 1024-k (1 <= k <= 512): split the packed transform, project, re-pack for the inverse transform.
// wk = W_2048^k.
template <bool INIT>
__device__ __forceinline__ void pair_pass(float2* Z, float2 wk, unsigned slots, bool self_paired, float Sk, float Sp) {
    const int sk = (int)(slots & 0xffffu), sp = (int)(slots >> 16);     // slots of bin k and bin 1024-k
    float2 Pk, Pp;
    if (!INIT) {
        const float2 a = Z[sk], b = cconj(Z[sp]);
        const float2 e = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y + b.y));
        const float2 d = csub(a, b);
        const float2 o = make_float2(0.5f * d.y, -0.5f * d.x);
        const float2 wo = cmul(wk, o);
        Pk = project(cadd(e, wo), Sk);
        Pp = project(cconj(csub(e, wo)), Sp);
    } else {
        Pk = make_float2(Sk, 0.f);
        Pp = make_float2(Sp, 0.f);
    }
    const float2 b2 = cconj(Pp);
    const float2 e2 = cadd(Pk, b2);
    const float2 t = cmul(cconj(wk), csub(Pk, b2));
    const float2 o2 = make_float2(-t.y, t.x);
    Z[sk] = cadd(e2, o2);
    if (!self_paired) Z[sp] = cconj(csub(e2, o2));
}

// overlap-add of the stored windowed segments at trimmed position n of an utterance (frames in ascending order),
// normalised by the window-sum-square like librosa.istft.  32-bit index arithmetic: an utterance is < 2^31 samples.
// Up to five overlapping frames (every shipped geometry) are loaded together; absent ones contribute +0.
__device__ __forceinline__ float ola_at(const float* __restrict__ seg, const float* __restrict__ wss_b, int n, int F,
                                        int hop, int lpad, int win, int wstride) {
    const unsigned c = (unsigned)(n + FH - lpad);          // position relative to the start of frame 0's support
    int fhi = (int)(c / (unsigned)hop);
    int off = (int)(c - (unsigned)fhi * (unsigned)hop);    // offset inside frame fhi's segment (< hop <= win)
    int flo = fhi;
    while (flo > 0 && off + hop < win) { off += hop; --flo; }
    if (fhi > F - 1) fhi = F - 1;
    const int cnt = fhi - flo + 1;
    const float* p = seg + (long long)flo * wstride + off;
    const int step = wstride - hop;
    const float ws = wss_b[n];
    float acc = 0.f;
    if (cnt <= 5) {
        float v[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) v[i] = i < cnt ? p[(long long)i * step] : 0.f;
#pragma unroll
        for (int i = 0; i < 5; ++i) acc += v[i];
    } else {
        for (int i = 0; i < cnt; ++i) acc += p[(long long)i * step];
    }
    if (ws > 1.17549435e-38f) acc /= ws;
    return acc;
}

template <bool INIT>
__global__ __launch_bounds__(256) void gl_fused(const float* __restrict__ S, const float* __restrict__ y,
                                               float* __restrict__ seg_out, const int4* __restrict__ meta,
                                               const float2* __restrict__ tw, const float* __restrict__ wfull, int G,
                                               int hop, int lpad, int win, int wstride, int nbin) {
    __shared__ float2 bufA[FH], bufB[FH], sT[TW_TOTAL];
    const int tid = threadIdx.x;
    for (int i = tid; i < TW_TOTAL; i += 256) sT[i] = tw[i];
    // this thread always owns samples 8 tid .. 8 tid + 7 of a frame: its eight window values (zero outside the window
    // support) live in registers for both the analysis and the synthesis side
    const int i0 = 8 * tid;
    float wv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) wv[e] = wfull[i0 + e];
    const bool touch = i0 + 8 > lpad && i0 < lpad + win;
    FusedConst fc;
    fused_const_init(fc, tid);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int chunk = (G + 7) / 8;
    const int gend = min(G, (xcd + 1) * chunk);
    int g = xcd * chunk + slot;
    int4 md = g < gend ? meta[g] : make_int4(0, 2, 0, 0);
    __syncthreads();
    // W_2048^k for this thread's two bin pairs: k = tid+1 directly, k = tid+257 through W^k = -i conj(W^(512-k))
    const float2 wk1 = sT[TW_PAIR + tid + 1];
    const float2 wr = sT[TW_PAIR + 255 - tid];
    const float2 wk2 = make_float2(-wr.y, -wr.x);
    for (; g < gend; g += nslots) {
        const int f = md.x, F = md.y, y0 = md.w;
        if (g + nslots < gend) md = meta[g + nslots];    // next frame's descriptor travels during this frame
        const float* Srow = S + (long long)g * nbin;
        // requested now, consumed after the forward transform
        const float s0 = Srow[0], sN = Srow[FH];
        const float sk1 = Srow[tid + 1], sp1 = Srow[FH - 1 - tid];
        const float sk2 = Srow[tid + 257], sp2 = Srow[FH - 257 - tid];
        if (!INIT) {
            // librosa.stft framing of the current signal: reflect padding + analysis window, zero outside the support
            const int len = hop * (F - 1);
            const float* yb = y + y0;
            const int nb = f * hop + i0 - FH;            // trimmed position of this thread's first sample
            float yv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) yv[e] = 0.f;
            if (touch) {
                if (nb >= 0 && nb + 8 <= len) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) yv[e] = yb[nb + e];
                } else {
                    const int period = 2 * (len - 1);
#pragma unroll 1
                    for (int e = 0; e < 8; ++e) {
                        if (i0 + e < lpad || i0 + e >= lpad + win) continue;
                        int n = nb + e;
                        if (n < 0 || n >= len) {                 // np.pad(mode='reflect'), repeated if needed
                            if (len == 1) n = 0;
                            else {
                                n %= period;
                                if (n < 0) n += period;
                                if (n >= len) n = period - n;
                            }
                        }
                        const float val = yb[n];
#pragma unroll
                        for (int q = 0; q < 8; ++q) if (q == e) yv[q] = val;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) bufA[fc.own ^ r] = make_float2(wv[2 * r] * yv[2 * r], wv[2 * r + 1] * yv[2 * r + 1]);
            __syncthreads();
            fft_pass<false, 0>(bufA, bufB, sT, fc, tid); __syncthreads();
            fft_pass<false, 1>(bufB, bufA, sT, fc, tid); __syncthreads();
            fft_pass<false, 2>(bufA, bufB, sT, fc, tid); __syncthreads();
            fft_pass<false, 3>(bufB, bufA, sT, fc, tid); __syncthreads();
            fft_pass<false, 4>(bufA, bufB, sT, fc, tid); __syncthreads();
        }
        if (tid == 0) {
            float P0 = s0, PN = sN;
            if (!INIT) {
                const float2 z = bufB[0];
                const float x0 = z.x + z.y, xN = z.x - z.y;
                P0 = s0 * (x0 / fmaxf(1e-8f, fabsf(x0)));
                PN = sN * (xN / fmaxf(1e-8f, fabsf(xN)));
            }
            bufB[0] = make_float2(P0 + PN, P0 - PN);
        }
        pair_pass<INIT>(bufB, wk1, fc.pk[0], false, sk1, sp1);
        pair_pass<INIT>(bufB, wk2, fc.pk[1], tid == 255, sk2, sp2);      // tid 255: bin 512 pairs with itself
        __syncthreads();
        fft_pass<true, 0>(bufB, bufA, sT, fc, tid); __syncthreads();
        fft_pass<true, 1>(bufA, bufB, sT, fc, tid); __syncthreads();
        fft_pass<true, 2>(bufB, bufA, sT, fc, tid); __syncthreads();
        fft_pass<true, 3>(bufA, bufB, sT, fc, tid); __syncthreads();
        fft_pass<true, 4>(bufB, bufA, sT, fc, tid); __syncthreads();
        if (touch) {
            float* out = seg_out + (long long)g * wstride + (i0 - lpad);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float2 tv = bufA[fc.own ^ r];
                const int ia = i0 + 2 * r, ib = ia + 1;
                if (ia >= lpad && ia < lpad + win) out[2 * r] = wv[2 * r] * (tv.x * (1.0f / FN));
                if (ib >= lpad && ib < lpad + win) out[2 * r + 1] = wv[2 * r + 1] * (tv.y * (1.0f / FN));
            }
        }
        __syncthreads();
    }
}

// window-sum-square at every kept sample (librosa.filters.window_sumsquare, float32, ascending frames)
__global__ __launch_bounds__(256) void gl_wss(const float* __restrict__ w2, const int* __restrict__ foff,
                                             const long long* __restrict__ yoff, float* __restrict__ wss, int n_fft,
                                             int hop, int lpad, int win) {
    const int b = blockIdx.y;
    const int F = foff[b + 1] - foff[b];
    const long long len = (long long)hop * (F - 1);
    const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
    if (n >= len) return;
    const long long m = n + n_fft / 2;
    long long flo = m - (lpad + win - 1);
    flo = flo > 0 ? (flo + hop - 1) / hop : 0;
    long long fhi = (m - lpad) / hop;
    if (fhi > F - 1) fhi = F - 1;
    float ws = 0.f;
    for (long long f = flo; f <= fhi; ++f) ws += w2[m - f * hop];
    wss[yoff[b] + n] = ws;
}

// final istft of the fused path: y[n] from the stored segments
__global__ __launch_bounds__(256) void gl_ola_seg(const float* __restrict__ seg, const float* __restrict__ wss,
                                                 const int* __restrict__ foff, const long long* __restrict__ yoff,
                                                 float* __restrict__ y, int hop, int lpad, int win, int wstride) {
    const int b = blockIdx.y;
    const int F = foff[b + 1] - foff[b];
    const long long len = (long long)hop * (F - 1);
    const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
    if (n >= len) return;
    y[yoff[b] + n] = ola_at(seg + (long long)foff[b] * wstride, wss + yoff[b], (int)n, F, hop, lpad, win, wstride);
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
    std::vector<int4> meta;
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
    if (bt.Y >= (1ll << 31)) { v->err = "batch too large: more than 2^31 samples"; return OPHV_ERR_INVALID; }
    bt.frame_utt.resize(bt.G);
    bt.meta.resize(bt.G);
    for (int b = 0; b < B; ++b)
        for (int g = bt.foff[b]; g < bt.foff[b + 1]; ++g) {
            bt.frame_utt[g] = b;
            bt.meta[g] = make_int4(g - bt.foff[b], n_frames[b], bt.foff[b], (int)bt.yoff[b]);
        }
    return OPHV_OK;
}

// grow-only buffers + FFT plans for G frames (rounded up so that nearby batch sizes share a plan)
int ensure_capacity(oph_vocoder* v, const Batch& bt) {
    const int n_fft = v->p.n_fft;
    const long long G = (bt.G + 255) / 256 * 256;
    if (G > v->capG) {
        int rc;
        if ((rc = dev_alloc(v, &v->S, G * v->nbin, false))) return rc;
        if ((rc = dev_alloc(v, &v->t.frame_utt, G, true))) return rc;
        if ((rc = dev_alloc(v, &v->t.meta, G, true))) return rc;
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
    (void)n_fft;
    return OPHV_OK;
}

// Buffers and hipFFT plans of the generic backend, created only when that backend actually runs: the default fused
// path never touches hipFFT (rocFFT compiles its kernels at plan creation, which can fail for reasons outside this
// library -- seen once as HIPFFT_PARSE_ERROR late in a long-lived process -- so a failed creation is retried once
// after the device has drained).
int ensure_generic(oph_vocoder* v) {
    const int n_fft = v->p.n_fft;
    if (v->capGg < v->capG) {
        int rc;
        if ((rc = dev_alloc(v, &v->X, v->capG * v->nbin, true))) return rc;
        if ((rc = dev_alloc(v, &v->tfr, v->capG * n_fft, false))) return rc;
        if ((rc = dev_alloc(v, &v->wfr, v->capG * n_fft, true))) return rc;
        v->capGg = v->capG;
    }
    if (v->planG != v->capG) {
        if (v->plan_c2r) hipfftDestroy(v->plan_c2r);
        if (v->plan_r2c) hipfftDestroy(v->plan_r2c);
        v->plan_c2r = v->plan_r2c = 0;
        for (int attempt = 0; attempt < 2; ++attempt) {
            const hipfftResult a = hipfftPlan1d(&v->plan_c2r, n_fft, HIPFFT_C2R, (int)v->capG);
            const hipfftResult b = a == HIPFFT_SUCCESS ? hipfftPlan1d(&v->plan_r2c, n_fft, HIPFFT_R2C, (int)v->capG) : a;
            if (a == HIPFFT_SUCCESS && b == HIPFFT_SUCCESS) break;
            if (v->plan_c2r) hipfftDestroy(v->plan_c2r);
            if (v->plan_r2c) hipfftDestroy(v->plan_r2c);
            v->plan_c2r = v->plan_r2c = 0;
            if (attempt == 1) { v->err = "hipfftPlan1d failed: hipfft error " + std::to_string((int)(a != HIPFFT_SUCCESS ? a : b)); return OPHV_ERR_DEVICE; }
            (void)hipDeviceSynchronize();
            (void)hipGetLastError();
        }
        FCHECK(hipfftSetStream(v->plan_c2r, v->stream));
        FCHECK(hipfftSetStream(v->plan_r2c, v->stream));
        v->planG = v->capG;
    }
    return OPHV_OK;
}

// launch geometry of gl_deemphasis: spans of 8192 samples with a warm-up long enough for a^warm <= 1e-17; filters that
// decay too slowly for that (a > ~0.9976) run each utterance as one sequential span
void launch_deemphasis(oph_vocoder* v, const float* x, const long long* yoff, float* out, int B, long long max_len) {
    const double a = std::fabs((double)v->p.preemphasis);
    int span = 8192, warm = 0;
    if (a > 0.0 && a < 1.0) {
        const double need = std::ceil(-39.2 / std::log(a));
        warm = need > 16384.0 ? -1 : (int)need;
    } else if (a >= 1.0) {
        warm = -1;
    }
    if (warm < 0 || max_len <= span) { span = (int)std::min<long long>(max_len, 0x7fffffff); warm = 0; }
    if (span < 1) span = 1;
    dim3 grid((unsigned)((max_len + span - 1) / span), B);
    hipLaunchKernelGGL(gl_deemphasis, grid, dim3(64), 0, v->stream, x, yoff, out, (double)v->p.preemphasis, span, warm);
}

bool use_fused(const oph_vocoder* v) { return v->backend == 0 && v->p.n_fft == FN; }

// buffers of the fused path: the windowed segments + the window-sum-square of every kept sample
int ensure_capacity_fused(oph_vocoder* v, const Batch& bt) {
    const long long G = (bt.G + 255) / 256 * 256;
    int rc;
    if (G > v->capGF) {
        if ((rc = dev_alloc(v, &v->wseg[0], G * v->wstride, true))) return rc;
        v->capGF = G;
    }
    if (bt.Y > v->capYF) {
        if ((rc = dev_alloc(v, &v->wss, bt.Y, false))) return rc;
        v->capYF = bt.Y;
    }
    return OPHV_OK;
}

int upload_tables(oph_vocoder* v, const Batch& bt, const std::vector<long long>& src_off) {
    VCHECK(hipMemcpyAsync(v->t.frame_utt, bt.frame_utt.data(), bt.G * sizeof(int), hipMemcpyHostToDevice, v->stream));
    VCHECK(hipMemcpyAsync(v->t.meta, bt.meta.data(), bt.G * sizeof(int4), hipMemcpyHostToDevice, v->stream));
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

// S prepared -> y (device), fused path: utils.py:99-109 with
//   istft           = inverse half of gl_fused (windowed segment per frame) + gl_ola_seg (overlap-add, /wss, trim)
//   stft + phase    = forward half of the next gl_fused launch, which goes straight on to the next inverse transform
int run_griffin_lim_fused(oph_vocoder* v, const Batch& bt, int n_iter) {
    const auto& p = v->p;
    dim3 ygrid((unsigned)(((long long)p.hop_length * (bt.maxF - 1) + 255) / 256), bt.B);
    hipLaunchKernelGGL(gl_wss, ygrid, dim3(256), 0, v->stream, v->d_w2, v->t.foff, v->t.yoff, v->wss, p.n_fft,
                       p.hop_length, v->lpad, p.win_length);
    long long want = (bt.G + 7) / 8 * 8;
    // persistent workgroups: 4 per CU are resident (102 VGPRs); measured 16 x 800 frames: 2/CU 7.3 ms, 3: 6.3, 4: 6.0,
    // 5: 6.8 (the fifth waits for a slot), 8: 6.0
    static const int per_cu = getenv("OPH_VOC_WGS_PER_CU") ? std::max(1, atoi(getenv("OPH_VOC_WGS_PER_CU"))) : 4;
    const unsigned nblk = (unsigned)(want < 256 * per_cu ? want : 256 * per_cu);
    hipLaunchKernelGGL(gl_fused<true>, dim3(nblk), dim3(256), 0, v->stream, v->S, (const float*)nullptr, v->wseg[0],
                       v->t.meta, v->d_tw, v->d_w, (int)bt.G, p.hop_length, v->lpad, p.win_length, v->wstride,
                       v->nbin);
    for (int it = 0; it <= n_iter; ++it) {
        hipLaunchKernelGGL(gl_ola_seg, ygrid, dim3(256), 0, v->stream, v->wseg[0], v->wss, v->t.foff, v->t.yoff, v->y,
                           p.hop_length, v->lpad, p.win_length, v->wstride);
        if (it == n_iter) break;
        hipLaunchKernelGGL(gl_fused<false>, dim3(nblk), dim3(256), 0, v->stream, v->S, v->y, v->wseg[0], v->t.meta,
                           v->d_tw, v->d_w, (int)bt.G, p.hop_length, v->lpad, p.win_length, v->wstride, v->nbin);
    }
    VCHECK(hipGetLastError());
    return OPHV_OK;
}

// S/X prepared -> y (device), generic path over hipFFT.  utils.py:99-109
int run_griffin_lim(oph_vocoder* v, const Batch& bt, int n_iter) {
    int rc;
    if ((rc = ensure_generic(v))) return rc;
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
    const bool fused = use_fused(v);
    if (!fused && (rc = ensure_generic(v))) return rc;
    hipLaunchKernelGGL(gl_prepare, dim3((unsigned)bt.G), dim3(256), 0, v->stream, d_src, v->t.src_off, v->t.frame_utt,
                       v->t.foff, v->S, fused ? (float2*)nullptr : v->X, v->nbin, (float)v->p.max_db, (float)v->p.ref_db,
                       (float)v->p.power,
                       denorm);
    if ((rc = fused ? run_griffin_lim_fused(v, bt, n_iter) : run_griffin_lim(v, bt, n_iter))) return rc;
    const float* result = v->y;
    if (deemph) {
        launch_deemphasis(v, v->y, v->t.yoff, v->wav, bt.B, (long long)v->p.hop_length * (bt.maxF - 1));
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
    if (use_fused(v) && (rc = ensure_capacity_fused(v, bt))) return rc;
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
    v->wstride = (p->win_length + 3) / 4 * 4;
    if (p->n_fft == FN) {
        std::vector<float2> tw(TW_TOTAL, make_float2(0.f, 0.f));
        auto root = [](long long e) {                       // W_2048^e evaluated in double
            const double a = -2.0 * M_PI * (double)(e % FN) / FN;
            return make_float2((float)std::cos(a), (float)std::sin(a));
        };
        for (int Ns : {4, 16, 64, 256})
            for (int r = 1; r <= 3; ++r)
                for (int k = 0; k < Ns; ++k) tw[tw_off(Ns) + (r - 1) * Ns + k] = root((long long)k * r * (512 / Ns));
        for (int k = 0; k <= 256; ++k) tw[TW_PAIR + k] = root(k);
        if (hipMalloc((void**)&v->d_tw, TW_TOTAL * sizeof(float2)) != hipSuccess ||
            hipMemcpy(v->d_tw, tw.data(), TW_TOTAL * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess)
            return fail("twiddle upload failed");
    }
    if (const char* e = std::getenv("OPH_VOCODER_BACKEND")) v->backend = std::atoi(e);
    *out = v;
    return OPHV_OK;
}

int oph_vocoder_set_backend(oph_vocoder* v, int backend) {
    if (!v || backend < 0 || backend > 1) return OPHV_ERR_INVALID;
    v->backend = backend;
    return OPHV_OK;
}

int oph_vocoder_destroy(oph_vocoder* v) {
    if (!v) return OPHV_OK;
    (void)hipSetDevice(v->device);
    if (v->stream) (void)hipStreamSynchronize(v->stream);
    if (v->plan_c2r) hipfftDestroy(v->plan_c2r);
    if (v->plan_r2c) hipfftDestroy(v->plan_r2c);
    void* bufs[] = {v->d_w, v->d_w2, v->S, v->X, v->tfr, v->wfr, v->y, v->wav, v->stage, v->t.frame_utt, v->t.foff,
                    v->t.yoff, v->t.src_off, v->t.meta, v->wseg[0], v->wseg[1], v->wss, v->d_tw};
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
    if (use_fused(v) && (rc = ensure_capacity_fused(v, bt))) return rc;
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
    if ((rc = ensure_generic(v))) return rc;
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
    if ((rc = ensure_generic(v))) return rc;
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
    launch_deemphasis(v, dx, doff, dy, 1, (long long)len);
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
