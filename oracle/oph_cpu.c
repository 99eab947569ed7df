/*
 * ORACLE -- TEST / BASELINE INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Plain C (C99 + OpenMP) restatement of Ophelia's Text2Mel + SSRN synthesis path on
 * the CPU, algorithm-faithful to the reference: the decode loop recomputes the whole
 * AudioEnc -> Attention -> AudioDec graph over all max_T positions at EVERY step and
 * keeps one column (synthesize.py:181-209).  It is the "cpu_baseline" leg of bench.py
 * (kind "port") and a second checker next to oracle/ophelia_oracle.py.
 *
 * PARITY STATUS: "parity unpinned" at the TensorFlow-primitive level (see the header
 * of oracle/ophelia_oracle.py); pinned against the reference-wiring goldens through
 * tests/test_oracle_c.py.
 *
 * Reference lines followed: modules.py 15-44 (embed) 47-75 (normalize) 91-146 (conv1d)
 * 148-207 (hc) 209-258 (conv1d_transpose); networks.py 121-212, 214-284, 286-325,
 * 360-435, 437-537; architectures.py 188-239; synthesize.py 150-260.
 *
 * Weights arrive as one flat float array in the inventory order produced by
 * oracle.ophelia_oracle.variable_shapes(hp) (TF variable creation order).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int vocab, e, d, c, n_mels, full_dim, r, max_N, max_T, win, nspeakers, spk_emb, multispeaker;
} cpu_dims;

#define LN_EPS 1e-12f
#define MASK_VALUE (-4294967296.0f)

void oph_cpu_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int oph_cpu_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* C[M][N] += A[M][K] (row stride lda) * B[K][N]; rows of A may be "virtual zeros" when arow[i] < 0 */
static void gemm_acc(const float* A, int lda, const int* arow, int M, int K, const float* B, int N, float* Cm, int ldc) {
#pragma omp parallel for schedule(static)
    for (int i0 = 0; i0 < M; i0 += 4) {
        const int mr = M - i0 < 4 ? M - i0 : 4;
        const float* ap[4];
        for (int r = 0; r < 4; ++r) {
            const int src = r < mr ? (arow ? arow[i0 + r] : i0 + r) : -1;
            ap[r] = src >= 0 ? A + (size_t)src * lda : NULL;
        }
        for (int j0 = 0; j0 < N; j0 += 64) {
            const int nb = N - j0 < 64 ? N - j0 : 64;
            float acc[4][64];
            for (int r = 0; r < 4; ++r)
                for (int j = 0; j < 64; ++j) acc[r][j] = 0.f;
            for (int k = 0; k < K; ++k) {
                const float* b = B + (size_t)k * N + j0;
                const float a0 = ap[0] ? ap[0][k] : 0.f, a1 = ap[1] ? ap[1][k] : 0.f;
                const float a2 = ap[2] ? ap[2][k] : 0.f, a3 = ap[3] ? ap[3][k] : 0.f;
                if (nb == 64) {
                    for (int j = 0; j < 64; ++j) {
                        const float bv = b[j];
                        acc[0][j] += a0 * bv; acc[1][j] += a1 * bv; acc[2][j] += a2 * bv; acc[3][j] += a3 * bv;
                    }
                } else {
                    for (int j = 0; j < nb; ++j) {
                        const float bv = b[j];
                        acc[0][j] += a0 * bv; acc[1][j] += a1 * bv; acc[2][j] += a2 * bv; acc[3][j] += a3 * bv;
                    }
                }
            }
            for (int r = 0; r < mr; ++r) {
                float* c = Cm + (size_t)(i0 + r) * ldc + j0;
                for (int j = 0; j < nb; ++j) c[j] += acc[r][j];
            }
        }
    }
}

static void layer_norm_rows(float* x, int M, int ld, int C, const float* gamma, const float* beta) {
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        float* r = x + (size_t)m * ld;
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += r[c];
        const float mean = s / (float)C;
        float q = 0.f;
        for (int c = 0; c < C; ++c) { const float dl = r[c] - mean; q += dl * dl; }
        const float rstd = 1.0f / sqrtf(q / (float)C + LN_EPS);
        for (int c = 0; c < C; ++c) r[c] = (r[c] - mean) * rstd * gamma[c] + beta[c];
    }
}

static inline float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

/* conv over (B,T,Cin) rows with kernel (size,Cin,Cout); offsets off[k] in time; out = bias + sum */
static void conv_rows(const float* x, int B, int T, int Cin, const float* kernel, const float* bias, int size,
                      const int* off, int Cout, float* out) {
    const int M = B * T;
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < Cout; ++n) out[(size_t)m * Cout + n] = bias[n];
    int* arow = (int*)malloc(sizeof(int) * (size_t)M);
    for (int k = 0; k < size; ++k) {
        for (int m = 0; m < M; ++m) {
            const int t = m % T, tt = t + off[k];
            arow[m] = (tt >= 0 && tt < T) ? m + off[k] : -1;
        }
        gemm_acc(x, Cin, arow, M, Cin, kernel + (size_t)k * Cin * Cout, Cout, out, Cout);
    }
    free(arow);
}

typedef struct { const float* p; } wcur;   /* cursor over the flat weight array */
static const float* take(wcur* w, size_t n) { const float* r = w->p; w->p += n; return r; }

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2 };

/* conv1d (modules.py:91-146): inventory order kernel, bias, beta, gamma */
static float* conv1d_layer(wcur* w, float* x, int B, int T, int Cin, int Cout, int causal, int act) {
    (void)causal;
    const float* kernel = take(w, (size_t)Cin * Cout);
    const float* bias = take(w, Cout);
    const float* beta = take(w, Cout);
    const float* gamma = take(w, Cout);
    float* out = (float*)malloc(sizeof(float) * (size_t)B * T * Cout);
    const int off0 = 0;
    conv_rows(x, B, T, Cin, kernel, bias, 1, &off0, Cout, out);
    layer_norm_rows(out, B * T, Cout, Cout, gamma, beta);
    const size_t n = (size_t)B * T * Cout;
    if (act == ACT_RELU) {
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; ++i) out[i] = out[i] > 0.f ? out[i] : 0.f;
    } else if (act == ACT_SIGMOID) {
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; ++i) out[i] = sigm(out[i]);
    }
    free(x);
    return out;
}

/* hc (modules.py:148-207): inventory order kernel, bias, H1/beta, H1/gamma, H2/beta, H2/gamma */
static float* hc_layer(wcur* w, float* x, int B, int T, int C, int size, int rate, int causal) {
    const float* kernel = take(w, (size_t)size * C * 2 * C);
    const float* bias = take(w, 2 * C);
    const float* b1 = take(w, C); const float* g1 = take(w, C);
    const float* b2 = take(w, C); const float* g2 = take(w, C);
    int off[3];
    for (int k = 0; k < size; ++k) off[k] = causal ? -(size - 1 - k) * rate : (k - (size - 1) / 2) * rate;
    const int M = B * T;
    float* h = (float*)malloc(sizeof(float) * (size_t)M * 2 * C);
    conv_rows(x, B, T, C, kernel, bias, size, off, 2 * C, h);
    layer_norm_rows(h, M, 2 * C, C, g1, b1);
    layer_norm_rows(h + C, M, 2 * C, C, g2, b2);
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        const float* hr = h + (size_t)m * 2 * C;
        float* xr = x + (size_t)m * C;
        for (int c = 0; c < C; ++c) {
            const float g = sigm(hr[c]);
            xr[c] = g * hr[C + c] + (1.0f - g) * xr[c];
        }
    }
    free(h);
    return x;
}

/* conv1d_transpose (modules.py:209-258), kernel (1,3,Cout,Cin): inventory kernel, bias, beta, gamma */
static float* convT_layer(wcur* w, float* x, int B, int T, int C) {
    const float* kt = take(w, (size_t)3 * C * C);
    const float* bias = take(w, C);
    const float* beta = take(w, C);
    const float* gamma = take(w, C);
    /* transpose taps to (Cin,Cout) for the row-major GEMM */
    float* wt = (float*)malloc(sizeof(float) * (size_t)3 * C * C);
    for (int k = 0; k < 3; ++k)
        for (int n = 0; n < C; ++n)
            for (int c = 0; c < C; ++c) wt[((size_t)k * C + c) * C + n] = kt[((size_t)k * C + n) * C + c];
    const int M = B * T;
    float* even = (float*)malloc(sizeof(float) * (size_t)M * C);
    float* odd = (float*)malloc(sizeof(float) * (size_t)M * C);
    int offs[2] = {0, -1};
    /* even: x[t].K0^T + x[t-1].K2^T ; odd: x[t].K1^T */
    float* k02 = (float*)malloc(sizeof(float) * (size_t)2 * C * C);
    memcpy(k02, wt, sizeof(float) * (size_t)C * C);
    memcpy(k02 + (size_t)C * C, wt + (size_t)2 * C * C, sizeof(float) * (size_t)C * C);
    conv_rows(x, B, T, C, k02, bias, 2, offs, C, even);
    conv_rows(x, B, T, C, wt + (size_t)C * C, bias, 1, offs, C, odd);
    float* out = (float*)malloc(sizeof(float) * (size_t)2 * M * C);
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        memcpy(out + (size_t)(2 * m) * C, even + (size_t)m * C, sizeof(float) * C);
        memcpy(out + (size_t)(2 * m + 1) * C, odd + (size_t)m * C, sizeof(float) * C);
    }
    layer_norm_rows(out, 2 * M, C, C, gamma, beta);
    free(wt); free(even); free(odd); free(k02); free(x);
    return out;
}

/* ---- networks ------------------------------------------------------------------------------ */
static size_t conv_params(int cin, int cout) { return (size_t)cin * cout + 3 * (size_t)cout; }
static size_t hc_params(int c, int size) { return (size_t)size * c * 2 * c + 2 * (size_t)c + 4 * (size_t)c; }

static size_t textenc_params(const cpu_dims* m) {
    return (size_t)m->vocab * m->e + conv_params(m->e, 2 * m->d) + conv_params(2 * m->d, 2 * m->d) +
           10 * hc_params(2 * m->d, 3) + 2 * hc_params(2 * m->d, 1);
}
static size_t audioenc_params(const cpu_dims* m) {
    return conv_params(m->n_mels, m->d) + 2 * conv_params(m->d, m->d) + 10 * hc_params(m->d, 3);
}
static size_t audiodec_params(const cpu_dims* m) {
    size_t n = conv_params(2 * m->d, m->d) + 6 * hc_params(m->d, 3) + 3 * conv_params(m->d, m->d) + conv_params(m->d, m->n_mels);
    if (m->multispeaker) n += (size_t)m->nspeakers * m->spk_emb + conv_params(m->d + m->spk_emb, m->d);
    return n;
}

/* TextEnc networks.py:121-212 : L (B,N) -> KV (B,N,2d) */
int oph_cpu_text_enc(const cpu_dims* m, const float* weights, const int32_t* L, int B, float* K, float* V) {
    wcur w = {weights};
    const int N = m->max_N, d = m->d;
    const float* table = take(&w, (size_t)m->vocab * m->e);
    float* x = (float*)malloc(sizeof(float) * (size_t)B * N * m->e);
    for (int i = 0; i < B * N; ++i) {
        const int id = L[i];
        for (int c = 0; c < m->e; ++c) x[(size_t)i * m->e + c] = id == 0 ? 0.f : table[(size_t)id * m->e + c];
    }
    x = conv1d_layer(&w, x, B, N, m->e, 2 * d, 0, ACT_RELU);
    x = conv1d_layer(&w, x, B, N, 2 * d, 2 * d, 0, ACT_NONE);
    for (int o = 0; o < 2; ++o)
        for (int j = 0, r = 1; j < 4; ++j, r *= 3) x = hc_layer(&w, x, B, N, 2 * d, 3, r, 0);
    for (int o = 0; o < 2; ++o) x = hc_layer(&w, x, B, N, 2 * d, 3, 1, 0);
    for (int o = 0; o < 2; ++o) x = hc_layer(&w, x, B, N, 2 * d, 1, 1, 0);
    for (int i = 0; i < B * N; ++i) {
        memcpy(K + (size_t)i * d, x + (size_t)i * 2 * d, sizeof(float) * d);
        memcpy(V + (size_t)i * d, x + (size_t)i * 2 * d + d, sizeof(float) * d);
    }
    free(x);
    return 0;
}

/* AudioEnc networks.py:214-284 over T positions */
static float* audio_enc(const cpu_dims* m, const float* wts, const float* S, int B, int T) {
    wcur w = {wts};
    const int d = m->d;
    float* x = (float*)malloc(sizeof(float) * (size_t)B * T * m->n_mels);
    memcpy(x, S, sizeof(float) * (size_t)B * T * m->n_mels);
    x = conv1d_layer(&w, x, B, T, m->n_mels, d, 1, ACT_RELU);
    x = conv1d_layer(&w, x, B, T, d, d, 1, ACT_RELU);
    x = conv1d_layer(&w, x, B, T, d, d, 1, ACT_NONE);
    for (int o = 0; o < 2; ++o)
        for (int j = 0, r = 1; j < 4; ++j, r *= 3) x = hc_layer(&w, x, B, T, d, 3, r, 1);
    for (int o = 0; o < 2; ++o) x = hc_layer(&w, x, B, T, d, 3, 3, 1);
    return x;
}

/* Attention networks.py:286-325 (dense over all N keys, masked, as the reference computes it) */
static float* attention(const cpu_dims* m, const float* Q, const float* K, const float* V, const int32_t* prev_max,
                        int B, int T, float* align_col, int col, int64_t* amax_col) {
    const int N = m->max_N, d = m->d;
    float* R = (float*)malloc(sizeof(float) * (size_t)B * T * 2 * d);
    const float scale = 1.0f / sqrtf((float)d);
#pragma omp parallel for schedule(static)
    for (int bt = 0; bt < B * T; ++bt) {
        const int b = bt / T, t = bt % T, p = prev_max[b];
        const float* q = Q + (size_t)bt * d;
        float* A = (float*)malloc(sizeof(float) * N);
        float mx = -INFINITY;
        for (int n = 0; n < N; ++n) {
            float s = 0.f;
            const float* kr = K + ((size_t)b * N + n) * d;
            for (int c = 0; c < d; ++c) s += q[c] * kr[c];
            s *= scale;
            const int masked = (n < p) || ((N - 1 - n) < (N - m->win - p));
            A[n] = masked ? MASK_VALUE : s;
            if (A[n] > mx) mx = A[n];
        }
        float den = 0.f;
        for (int n = 0; n < N; ++n) { A[n] = expf(A[n] - mx); den += A[n]; }
        int arg = 0; float best = -1.f;
        for (int n = 0; n < N; ++n) { A[n] /= den; if (A[n] > best) { best = A[n]; arg = n; } }
        float* r = R + (size_t)bt * 2 * d;
        for (int c = 0; c < d; ++c) r[c] = 0.f;
        for (int n = 0; n < N; ++n) {
            if (A[n] == 0.f) continue;
            const float* vr = V + ((size_t)b * N + n) * d;
            for (int c = 0; c < d; ++c) r[c] += A[n] * vr[c];
        }
        memcpy(r + d, q, sizeof(float) * d);
        if (t == col) {
            for (int n = 0; n < N; ++n) align_col[(size_t)b * N + n] = A[n];
            amax_col[b] = arg;
        }
        free(A);
    }
    return R;
}

/* AudioDec networks.py:360-435 */
static float* audio_dec(const cpu_dims* m, const float* wts, float* R, const int32_t* spk, int B, int T) {
    wcur w = {wts};
    const int d = m->d;
    float* x = conv1d_layer(&w, R, B, T, 2 * d, d, 1, ACT_NONE);
    if (m->multispeaker) {
        const float* table = take(&w, (size_t)m->nspeakers * m->spk_emb);
        const int cc = d + m->spk_emb;
        float* y = (float*)malloc(sizeof(float) * (size_t)B * T * cc);
        for (int bt = 0; bt < B * T; ++bt) {
            const int id = spk[bt / T];
            memcpy(y + (size_t)bt * cc, x + (size_t)bt * d, sizeof(float) * d);
            for (int c = 0; c < m->spk_emb; ++c) y[(size_t)bt * cc + d + c] = id == 0 ? 0.f : table[(size_t)id * m->spk_emb + c];
        }
        free(x);
        x = conv1d_layer(&w, y, B, T, cc, d, 0, ACT_NONE);
    }
    for (int j = 0, r = 1; j < 4; ++j, r *= 3) x = hc_layer(&w, x, B, T, d, 3, r, 1);
    for (int o = 0; o < 2; ++o) x = hc_layer(&w, x, B, T, d, 3, 1, 1);
    for (int o = 0; o < 3; ++o) x = conv1d_layer(&w, x, B, T, d, d, 1, ACT_RELU);
    x = conv1d_layer(&w, x, B, T, d, m->n_mels, 1, ACT_SIGMOID);     /* squash_output_t2m */
    return x;
}

/* synth_codedtext2mel synthesize.py:150-230 -- faithful full recompute per step.
 * weights = flat Text2Mel weights (TextEnc | AudioEnc | AudioDec).  max_steps < max_T bounds the
 * number of loop iterations executed (for bounded baseline samples); stop: 0 reference, 1 never. */
int oph_cpu_text2mel(const cpu_dims* m, const float* weights, const float* K, const float* V, const int32_t* ends,
                     const int32_t* spk, int B, int stop_mode, int max_steps, float* Y, int32_t* t_ends,
                     float* alignments, int32_t* steps_run) {
    const int T = m->max_T, N = m->max_N, nm = m->n_mels;
    const float* w_ae = weights + textenc_params(m);
    const float* w_ad = w_ae + audioenc_params(m);
    memset(Y, 0, sizeof(float) * (size_t)B * T * nm);
    memset(alignments, 0, sizeof(float) * (size_t)B * N * T);
    int32_t* prev = (int32_t*)calloc(B, sizeof(int32_t));
    int64_t* amax = (int64_t*)calloc(B, sizeof(int64_t));
    float* acol = (float*)malloc(sizeof(float) * (size_t)B * N);
    float* S = (float*)malloc(sizeof(float) * (size_t)B * T * nm);
    for (int b = 0; b < B; ++b) t_ends[b] = T;
    int steps = 0;
    for (int j = 0; j < T && j < max_steps; ++j) {
        for (int b = 0; b < B; ++b) {                       /* S = shift right by one frame, architectures.py:191 */
            memset(S + (size_t)b * T * nm, 0, sizeof(float) * nm);
            memcpy(S + (size_t)b * T * nm + nm, Y + (size_t)b * T * nm, sizeof(float) * (size_t)(T - 1) * nm);
        }
        float* Q = audio_enc(m, w_ae, S, B, T);
        float* R = attention(m, Q, K, V, prev, B, T, acol, j, amax);
        free(Q);
        float* Yall = audio_dec(m, w_ad, R, spk, B, T);
        for (int b = 0; b < B; ++b) {
            memcpy(Y + ((size_t)b * T + j) * nm, Yall + ((size_t)b * T + j) * nm, sizeof(float) * nm);
            for (int n = 0; n < N; ++n) alignments[((size_t)b * N + n) * T + j] = acol[(size_t)b * N + n];
            prev[b] = (int32_t)amax[b];
        }
        free(Yall);
        steps = j + 1;
        int all = 1;
        for (int b = 0; b < B; ++b) {
            if (t_ends[b] == T && amax[b] >= ends[b]) t_ends[b] = j;
            if (t_ends[b] == T) all = 0;
        }
        if (stop_mode == 0 && all) break;
    }
    *steps_run = steps;
    free(prev); free(amax); free(acol); free(S);
    return 0;
}

/* SSRN networks.py:437-537 : Y (B,T,n_mels) -> Z (B,r*T,full_dim); weights = flat SSRN weights */
int oph_cpu_ssrn(const cpu_dims* m, const float* weights, const float* Y, int B, int T, float* Z) {
    wcur w = {weights};
    const int c = m->c, F = m->full_dim;
    float* x = (float*)malloc(sizeof(float) * (size_t)B * T * m->n_mels);
    memcpy(x, Y, sizeof(float) * (size_t)B * T * m->n_mels);
    x = conv1d_layer(&w, x, B, T, m->n_mels, c, 0, ACT_NONE);
    for (int j = 0, r = 1; j < 2; ++j, r *= 3) x = hc_layer(&w, x, B, T, c, 3, r, 0);
    const int ntr = m->r == 4 ? 2 : 3;
    for (int o = 0; o < ntr; ++o) {
        x = convT_layer(&w, x, B, T, c);
        T *= 2;
        for (int j = 0, r = 1; j < 2; ++j, r *= 3) x = hc_layer(&w, x, B, T, c, 3, r, 0);
    }
    x = conv1d_layer(&w, x, B, T, c, 2 * c, 0, ACT_NONE);
    for (int o = 0; o < 2; ++o) x = hc_layer(&w, x, B, T, 2 * c, 3, 1, 0);
    x = conv1d_layer(&w, x, B, T, 2 * c, F, 0, ACT_NONE);
    for (int o = 0; o < 2; ++o) x = conv1d_layer(&w, x, B, T, F, F, 0, ACT_RELU);
    x = conv1d_layer(&w, x, B, T, F, F, 0, ACT_SIGMOID);             /* squash_output_ssrn */
    memcpy(Z, x, sizeof(float) * (size_t)B * T * F);
    free(x);
    return 0;
}

size_t oph_cpu_text2mel_params(const cpu_dims* m) { return textenc_params(m) + audioenc_params(m) + audiodec_params(m); }
