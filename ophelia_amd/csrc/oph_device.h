// Device-side helpers shared by the kernel translation units of libophelia_hip.so (gfx950 only):
// DPP wave reductions, activations, and the attention row (networks.py:286-325).
#pragma once
#include "oph_internal.h"

namespace oph {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Wave-wide sum, result uniform in all 64 lanes.  A __shfl_xor butterfly is 6 dependent
// ds_bpermute round trips (~60 cycles each); profiles/r01 ablation: 1.4 us of a 8 us decoder layer
// was LayerNorm reductions.  DPP does the 16-lane row reduction in 4 VALU ops (xor 1, xor 2,
// half-mirror, mirror), and the 4 row totals are combined through v_readlane (scalar).
template <int CTRL>
static __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
static __device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);     // row_half_mirror
    v += dpp_mov<0x140>(v);     // row_mirror  -> every lane holds its 16-lane row total
    const int iv = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
}
static __device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
static __device__ __forceinline__ float apply_act(float x, int act) {
    return act == ACT_RELU ? fmaxf(x, 0.0f) : (act == ACT_SIGMOID ? sigmoidf_(x) : x);
}
// Hardware transcendental forms (v_exp_f32 / v_rcp_f32 / v_rsq_f32, ~1 ulp each).  The latency-critical decoder run uses
// them: the IEEE expf / division / sqrt expansions are ~50 dependent instructions per element, which at one wave per
// SIMD is most of a layer's prologue time (profiles/r02 stamps: 1.6 us -> 0.5 us per highway layer).
static __device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
static __device__ __forceinline__ float fast_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
static __device__ __forceinline__ float fast_act(float x, int act) {
    return act == ACT_RELU ? fmaxf(x, 0.0f) : (act == ACT_SIGMOID ? fast_sigmoid(x) : x);
}
// 16-byte row pieces exchanged with a kernel that is RUNNING (no kernel boundary, so no cache maintenance): two 8-byte
// agent-scope relaxed atomics (sc1: past the CU's L1, coherent across the XCDs' L2s, write-through), never plain accesses.
static __device__ __forceinline__ f32x4 ld_coherent(const float* p) {
    typedef unsigned long long u64_;
    const u64_ lo = __hip_atomic_load((const u64_*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u64_ hi = __hip_atomic_load((const u64_*)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    f32x4 v;
    v[0] = __uint_as_float((unsigned)lo); v[1] = __uint_as_float((unsigned)(lo >> 32));
    v[2] = __uint_as_float((unsigned)hi); v[3] = __uint_as_float((unsigned)(hi >> 32));
    return v;
}
static __device__ __forceinline__ void st_coherent(float* p, const f32x4& v) {
    typedef unsigned long long u64_;
    __hip_atomic_store((u64_*)p, (u64_)__float_as_uint(v[0]) | ((u64_)__float_as_uint(v[1]) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store((u64_*)p + 1, (u64_)__float_as_uint(v[2]) | ((u64_)__float_as_uint(v[3]) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// 16 bytes written by another CU during this launch (write-through sc1 stores, then a counter / granule the reader has seen): ONE
// 16-byte sc1 load (past the CU's L1, coherent across the XCDs) instead of ld_coherent's two 8-byte atomics -- half the requests
// (MI355X_MICROARCH.md: 8-byte accesses run at 0.54-0.70x the 16-byte rate; 16-byte sc1 halves observed untorn)
typedef int i32x4_ __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ f32x4 ld_sc1_b128(const float* base, unsigned byte_off) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    const i32x4_ v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16 /* sc1 */);
    return __builtin_bit_cast(f32x4, v);
}
// ... and the matching 16-byte write-through store (one request per lane; consecutive lanes fill whole cache lines)
static __device__ __forceinline__ void st_sc1_b128(float* base, unsigned byte_off, const f32x4& v) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_, v), r, (int)byte_off, 0, 16 /* sc1 */);
}
static __device__ __forceinline__ bool stopped(const int* stop_after, int t) {
    return stop_after != nullptr && t > *stop_after;
}

// =====================================================================================
// Attention (networks.py:286-325, monotonic synthesis branch).
// With prev_max = p the unmasked keys are [p, min(p+win, N)): keys n<p (key_masks) and
// n>=p+win (reverse_masks, only when N-win-p>0) receive -2**32+1 and their softmax terms
// are exactly 0 in fp32, so only the window is evaluated; outputs are identical.
// One wavefront per (utterance, query row); lane holds d/64 channels.
// =====================================================================================
constexpr int ATT_NV = 2;     // d <= 512
constexpr int ATT_WMAX = 8;   // attention_win_size <= 8

struct AttnOut { float prob[ATT_WMAX]; int nwin; int arg; };

template <int NV>       // NV*256 >= d
static __device__ __forceinline__ AttnOut attend_window(const f32x4 (&q)[NV], const float* Kb, const float* Vb, int ldkv,
                                                 int p, int N, int win, int d, int lane, f32x4 (&ctx)[NV]) {
    AttnOut o;
    o.nwin = min(win, N - p);
    const float scale = 1.0f / sqrtf((float)d);      // tf.rsqrt(tf.to_float(hp.d))  networks.py:300
    float sc[ATT_WMAX];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < ATT_WMAX; ++i) {
        sc[i] = -INFINITY;
        if (i < o.nwin) {
            const float* kr = Kb + (size_t)(p + i) * ldkv;
            float s = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int c = (v * 64 + lane) * 4;
                if (c < d) {
                    const f32x4 kv = *(const f32x4*)(kr + c);
                    s += q[v][0] * kv[0] + q[v][1] * kv[1] + q[v][2] * kv[2] + q[v][3] * kv[3];
                }
            }
            sc[i] = wave_sum(s) * scale;
            mx = fmaxf(mx, sc[i]);
        }
    }
    float den = 0.f;
#pragma unroll
    for (int i = 0; i < ATT_WMAX; ++i) {
        o.prob[i] = i < o.nwin ? expf(sc[i] - mx) : 0.f;
        den += o.prob[i];
    }
    o.arg = 0;
    float best = -1.f;
#pragma unroll
    for (int i = 0; i < ATT_WMAX; ++i) {
        o.prob[i] = o.prob[i] / den;
        if (i < o.nwin && o.prob[i] > best) { best = o.prob[i]; o.arg = i; }   // first max on ties
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        ctx[v] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int c = (v * 64 + lane) * 4;
        if (c < d) {
#pragma unroll
            for (int i = 0; i < ATT_WMAX; ++i)
                if (i < o.nwin) {
                    const f32x4 vv = *(const f32x4*)(Vb + (size_t)(p + i) * ldkv + c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ctx[v][e] += o.prob[i] * vv[e];
                }
        }
    }
    return o;
}

// hp.turn_off_monotonic_for_synthesis (networks.py:307-309): no forcibly-incremental window; the keys past the text
// (n >= text_length + 1) are masked with -2**32+1, i.e. contribute exactly 0.  One wavefront, up to 256 keys: key n's
// logit / probability lives in lane n%64, slot n/64.  arg = first maximum, like tf.argmax.
constexpr int ATT_FULL_SLOTS = 4;
struct AttnFull { float prob[ATT_FULL_SLOTS]; int arg; };

static __device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
static __device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}

static __device__ __forceinline__ AttnFull attend_full(const f32x4 (&q)[ATT_NV], const float* Kb, const float* Vb, int ldkv,
                                               int nkeys, int d, int lane, f32x4 (&ctx)[ATT_NV]) {
    AttnFull o;
    const float scale = 1.0f / sqrtf((float)d);
    float sc[ATT_FULL_SLOTS];
#pragma unroll
    for (int s_ = 0; s_ < ATT_FULL_SLOTS; ++s_) sc[s_] = -INFINITY;
    for (int n = 0; n < nkeys; ++n) {
        const float* kr = Kb + (size_t)n * ldkv;
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < ATT_NV; ++v) {
            const int c = (v * 64 + lane) * 4;
            if (c < d) {
                const f32x4 kv = *(const f32x4*)(kr + c);
                s += q[v][0] * kv[0] + q[v][1] * kv[1] + q[v][2] * kv[2] + q[v][3] * kv[3];
            }
        }
        s = wave_sum(s) * scale;
        if (lane == (n & 63)) {
#pragma unroll
            for (int s_ = 0; s_ < ATT_FULL_SLOTS; ++s_)
                if (s_ == (n >> 6)) sc[s_] = s;
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int s_ = 0; s_ < ATT_FULL_SLOTS; ++s_) mx = fmaxf(mx, sc[s_]);
    mx = wave_max_f(mx);
    float den = 0.f;
#pragma unroll
    for (int s_ = 0; s_ < ATT_FULL_SLOTS; ++s_) {
        o.prob[s_] = (s_ * 64 + lane < nkeys) ? expf(sc[s_] - mx) : 0.f;
        den += o.prob[s_];
    }
    den = wave_sum(den);
    float best = -1.f;
    int bi = 0x7fffffff;
#pragma unroll
    for (int s_ = 0; s_ < ATT_FULL_SLOTS; ++s_) {
        o.prob[s_] = o.prob[s_] / den;
        if (s_ * 64 + lane < nkeys && o.prob[s_] > best) { best = o.prob[s_]; bi = s_ * 64 + lane; }
    }
    const float gbest = wave_max_f(best);
    o.arg = wave_min_i(best == gbest ? bi : 0x7fffffff);
#pragma unroll
    for (int v = 0; v < ATT_NV; ++v) ctx[v] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int n = 0; n < nkeys; ++n) {
        float pn = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < ATT_FULL_SLOTS; ++s_)
            if (s_ == (n >> 6)) pn = __shfl(o.prob[s_], n & 63);
        const float* vr = Vb + (size_t)n * ldkv;
#pragma unroll
        for (int v = 0; v < ATT_NV; ++v) {
            const int c = (v * 64 + lane) * 4;
            if (c < d) {
                const f32x4 vv = *(const f32x4*)(vr + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) ctx[v][e] += pn * vv[e];
            }
        }
    }
    return o;
}


}  // namespace oph
