// cone_gemm_ln: one layer of the AudioDec history cone as ONE launch -- conv-as-GEMM on the fp32 MFMA with the layer's
// LayerNorm (+ highway gate + residual) in the epilogue (gfx950).
//
// The cone (oph_api.hip: launch_cone) re-evaluates AudioDec's highway-layer inputs at 84, 82, 44, 14, 4, 2 history
// positions under the current attention mask every step (networks.py:311, synthesize.py:181-183).  Round 1 ran every
// layer as split-K GEMM partials + a LayerNorm kernel: 13 launches, 3.3x the algorithmic traffic, ~145 us per step.
// Here a workgroup owns a 64- or 32-row x 32-column tile; LayerNorm needs whole rows, so the NT column tiles of a row
// block exchange per-row partial sums {sum, sum of squares} through 8-byte {epoch, value} granules (relaxed agent-scope
// atomics, cdna_hip_programming.md Guideline 16 R2) -- one ~1.5 us hop instead of a kernel boundary and a round trip
// of the raw tile through memory.  For a highway layer the weight columns are packed so that a tile holds channels
// [16j, 16j+16) of BOTH halves (H1 | H2): after the hop every workgroup can finish gate + residual for its channels.
//
// Variance is E[h^2] - mean^2 from the exchanged sums (one hop; the two-pass form would need two).  h is O(1) and
// |mean| <~ std for these layers, so the cancellation costs ~1e-7 relative -- far inside the 1e-4 test tolerance.
#include "oph_internal.h"
#include "oph_device.h"

#include <map>

namespace oph {

typedef unsigned long long u64;
constexpr long long CONE_TIMEOUT_TICKS = 200000000LL;      // 2 s of the 100 MHz clock

// Tile = BM rows x 32 columns, 4 waves: (BM/32) row halves x KS wave groups that split K (one wave per SIMD: a tile's
// fp32-MFMA time is its flops / 614 GFLOP/s whatever the wave arrangement, so small tiles are what spreads a layer over
// the CUs -- a 64x64x768 tile alone is 10 us).  BM = 64, KS = 2 for the large layers; BM = 32, KS = 4 for the small ones.
constexpr int CONE_BN = 32;
template <int BM, int KS>
__global__ __launch_bounds__(256) void cone_gemm_ln(ConeGemmArgs a) {
    static_assert((BM / 32) * KS == 4, "four waves per workgroup");
    if (stopped(a.stop_after, a.t)) return;
    constexpr int BN = CONE_BN, BK = 32, LD = 36, LDH = BN + 1;
    constexpr int GT = 64 * (BM / 32);                 // threads of one K group
    constexpr int AR = BM * 8 / GT, BR = BN * 8 / GT;  // float4 staging loads per thread
    constexpr int RPP = GT / 8;                        // rows staged per pass
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int GROUP_FLOATS = 2 * (BM + BN) * LD;
    int* srow_s = (int*)(smem + KS * GROUP_FLOATS);    // [3][BM] source row per tap (-1 = zeros)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int grp = tid / GT, gt = tid - grp * GT, wr = (gt >> 6);          // K group; thread in group; row half of this wave
    float* As = smem + grp * GROUP_FLOATS;
    float* Bs = As + 2 * BM * LD;
    const int NT = a.NT;
    const int tm = blockIdx.x / NT, tn = blockIdx.x - tm * NT;          // the NT tiles of a row block are neighbours in dispatch order
    const int m0 = tm * BM, n0 = tn * BN;

    for (int i = tid; i < BM * a.ntaps; i += 256) {
        const int tap = i / BM, m = m0 + (i - tap * BM);
        int src = -1;
        if (m < a.M) {
            if (a.dense) src = m;
            else {
                const int ip = m / a.Bpad, b = m - ip * a.Bpad;
                if (a.j >= a.need[tap * a.n_out + ip]) src = a.tab[tap * a.n_out + ip] * a.Bpad + b;
            }
        }
        srow_s[i] = src;
    }
    __syncthreads();

    // ---- K loop of this wave group (prefetch distance 2, LDS double buffer; every group runs the same number of barriers)
    const int lrow = gt >> 3, kq = gt & 7;
    const int kpt = a.kc / BK, nk_all = a.ntaps * kpt;
    const int ks0 = grp * nk_all / KS, nk = (grp + 1) * nk_all / KS - ks0;
    const int nk_max = (nk_all + KS - 1) / KS;
    f32x4 ra0[AR], rb0[BR], ra1[AR], rb1[BR];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto load_global = [&](int sl, f32x4 (&ra)[AR], f32x4 (&rb)[BR]) {
        const int s = ks0 + sl;
        const int tap = s / kpt, ko = (s - tap * kpt) * BK + kq * 4;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int src = srow_s[tap * BM + lrow + RPP * i];
            ra[i] = src >= 0 ? *(const f32x4*)(a.X + (size_t)src * a.ldx + ko) : zero4;
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) rb[i] = *(const f32x4*)(a.Wt + (size_t)(n0 + lrow + RPP * i) * a.ldw + tap * a.kc + ko);
    };
    auto store_lds = [&](int buf, const f32x4 (&ra)[AR], const f32x4 (&rb)[BR]) {
#pragma unroll
        for (int i = 0; i < AR; ++i) *(f32x4*)(As + buf * BM * LD + (lrow + RPP * i) * LD + kq * 4) = ra[i];
#pragma unroll
        for (int i = 0; i < BR; ++i) *(f32x4*)(Bs + buf * BN * LD + (lrow + RPP * i) * LD + kq * 4) = rb[i];
    };
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int r32 = lane & 31, kh = lane >> 5;
    auto compute = [&](int buf) {
        const float* Ab = As + buf * BM * LD + (wr * 32 + r32) * LD + kh * 4;
        const float* Bb = Bs + buf * BN * LD + r32 * LD + kh * 4;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const f32x4 af = *(const f32x4*)(Ab + kk * 8), bf = *(const f32x4*)(Bb + kk * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc, 0, 0, 0);
        }
    };
    if (nk > 0) load_global(0, ra0, rb0);
    if (nk > 1) load_global(1, ra1, rb1);
    if (nk > 0) store_lds(0, ra0, rb0);
    __syncthreads();
    for (int s = 0; s < nk_max; s += 2) {
        if (s + 2 < nk) load_global(s + 2, ra0, rb0);
        if (s < nk) compute(0);
        if (s + 1 < nk) store_lds(1, ra1, rb1);
        __syncthreads();
        if (s + 1 >= nk_max) break;
        if (s + 3 < nk) load_global(s + 3, ra1, rb1);
        if (s + 1 < nk) compute(1);
        if (s + 2 < nk) store_lds(0, ra0, rb0);
        __syncthreads();
    }

    // ---- K groups' partial tiles meet in LDS: Hs[BM][33] = sum + bias
    // C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    float* Hs = smem;                                  // the staging buffers are free now (barrier above)
    float* Ps = smem + BM * LDH;                       // [KS-1][BM][LDH]
    if (grp > 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e)
            Ps[((grp - 1) * BM + wr * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh) * LDH + r32] = acc[e];
    }
    __syncthreads();
    if (grp == 0) {
        const float bv = a.bias[n0 + r32];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int idx = (wr * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh) * LDH + r32;
            float v = acc[e] + bv;
#pragma unroll
            for (int g2 = 0; g2 < KS - 1; ++g2) v += Ps[g2 * BM * LDH + idx];
            Hs[idx] = v;
        }
    }
    __syncthreads();
    if (tid >= BM * 4) return;                         // epilogue: 4 lanes per row

    // ---- per-row partial sums of this tile -> granules -> whole-row statistics
    const int row = tid >> 2, q = tid & 3;             // q: 0 sum / 1 sum of squares of columns 0..15, 2 / 3 of columns 16..31
    {
        const float* hr = Hs + row * LDH + (q >> 1) * 16;
        float s = 0.f;
        if (q & 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s = fmaf(hr[i], hr[i], s);
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) s += hr[i];
        }
        u64* gp = a.stats + (((size_t)tm * NT + tn) * 64 + row) * 4 + q;
        __hip_atomic_store(gp, ((u64)a.epoch << 32) | (u64)__float_as_uint(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    float tot = 0.f;
    {
        const u64* gp = a.stats + ((size_t)tm * NT * 64 + row) * 4 + q;
        long long t0 = 0;
        for (int it = 0;; ++it) {
            bool ok = true;
            float s = 0.f;
            for (int j = 0; j < NT; ++j) {
                const u64 gv = __hip_atomic_load(gp + (size_t)j * 64 * 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = ok && (unsigned)(gv >> 32) == a.epoch;
                s += __uint_as_float((unsigned)gv);
            }
            bool give_up = false;
            if (!__all(ok) && it >= 32 && (it & 31) == 0) {
                const long long now = wall_clock64();
                if (t0 == 0) t0 = now;
                give_up = now - t0 > CONE_TIMEOUT_TICKS || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                if (give_up && lane == 0) __hip_atomic_store(a.err, 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (__all(ok) || give_up) { tot = s; break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    // the four lanes of a row hold {sum_a, sumsq_a, sum_b, sumsq_b}: a = first 16 columns of every tile, b = the others
    const int qb = lane & ~3;
    const float sa = __shfl(tot, qb), qa = __shfl(tot, qb + 1), sb = __shfl(tot, qb + 2), qq = __shfl(tot, qb + 3);
    const float invc = 1.0f / (float)a.C;
    const int m = m0 + row;
    if (m >= a.M) return;
    float* y = a.Y + (size_t)m * a.ldy;
    if (a.hc) {
        // a = H1 (gate), b = H2 (transformation): y = sigmoid(LN1(H1)) * LN2(H2) + (1 - sigmoid) * x   (modules.py:194-203)
        const float m1 = a.nonorm ? 0.f : sa * invc, m2 = a.nonorm ? 0.f : sb * invc;
        const float r1 = a.nonorm ? 1.0f : 1.0f / sqrtf(fmaxf(qa * invc - m1 * m1, 0.f) + LN_EPS);
        const float r2 = a.nonorm ? 1.0f : 1.0f / sqrtf(fmaxf(qq * invc - m2 * m2, 0.f) + LN_EPS);
        const int ch = tn * 16 + q * 4;                // 4 channels per lane
        size_t rrow = m;
        if (a.restab) rrow = (size_t)a.restab[m / a.Bpad] * a.Bpad + (m % a.Bpad);
        const f32x4 xv = *(const f32x4*)(a.Xres + rrow * a.ldres + ch);
        const float* h1 = Hs + row * LDH + q * 4;
        const float* h2 = h1 + 16;
        const f32x4 g1 = *(const f32x4*)(a.g1 + ch), b1 = *(const f32x4*)(a.b1 + ch);
        const f32x4 g2 = *(const f32x4*)(a.g2 + ch), b2 = *(const f32x4*)(a.b2 + ch);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gte = sigmoidf_((h1[e] - m1) * r1 * g1[e] + b1[e]);
            const float u = (h2[e] - m2) * r2 * g2[e] + b2[e];
            o[e] = gte * u + (1.0f - gte) * xv[e];
        }
        *(f32x4*)(y + ch) = o;
    } else {
        const float mean = a.nonorm ? 0.f : (sa + sb) * invc;
        const float rstd = a.nonorm ? 1.0f : 1.0f / sqrtf(fmaxf((qa + qq) * invc - mean * mean, 0.f) + LN_EPS);
        const int cl = q * 8, ch = tn * 32 + cl;       // 8 channels per lane
        const float* hh = Hs + row * LDH + cl;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            if (ch + 4 * v < a.C) {
                const f32x4 g1 = *(const f32x4*)(a.g1 + ch + 4 * v), b1 = *(const f32x4*)(a.b1 + ch + 4 * v);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = apply_act((hh[4 * v + e] - mean) * rstd * g1[e] + b1[e], a.act);
                *(f32x4*)(y + ch + 4 * v) = o;
            }
        }
        if (a.spk_table && tn == 0) {                  // speaker embedding appended to the next layer's input (networks.py:381-387)
            const int id = a.spk_ids[m % a.Bpad];
            for (int c2 = q; c2 < a.spk_dim; c2 += 4) y[a.C + c2] = id == 0 ? 0.f : a.spk_table[(size_t)id * a.spk_dim + c2];
        }
    }
}

template <int BM, int KS>
static void launch_cone_gemm_t(const ConeGemmArgs& a, hipStream_t s) {
    static thread_local std::map<int, bool> done;
    const size_t lds = (size_t)KS * 2 * (BM + CONE_BN) * 36 * 4 + 3 * BM * 4;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!done[dev]) { (void)hipFuncSetAttribute((const void*)cone_gemm_ln<BM, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); done[dev] = true; }
    const int MT = (a.M + BM - 1) / BM;
    hipLaunchKernelGGL((cone_gemm_ln<BM, KS>), dim3(MT * a.NT), dim3(256), lds, s, a);
}
// small_rows: 32-row tiles with K split over all four waves (layers with few rows), else 64-row tiles, K split in two
void launch_cone_gemm(const ConeGemmArgs& a, int small_rows, hipStream_t s) {
    if (small_rows) launch_cone_gemm_t<32, 4>(a, s);
    else launch_cone_gemm_t<64, 2>(a, s);
}
int cone_gemm_tile_cols() { return CONE_BN; }

}  // namespace oph
