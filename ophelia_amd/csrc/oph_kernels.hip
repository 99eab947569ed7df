// HIP kernels for Ophelia's Text2Mel + SSRN synthesis path on MI355X (gfx950, CDNA4).
// Wave = 64 lanes.  All arithmetic fp32; contractions on the fp32-input MFMA
// (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32: exact fp32, bitwise an fmaf chain).
//
// Kernels
//   conv_gemm_f32<BM,BN>  batched conv-as-GEMM (TextEnc, SSRN, AudioDec history cone)
//   ln_rows<NV>           LayerNorm epilogues: conv(LN+act) and highway (2xLN+gate+mix)
//   dec_layer16           fused M=16 decoder layer: previous layer's LN/gate as prologue,
//                         then a 16 x K . K x 16 slice per workgroup, K split over 4 waves
//   row_chain             row-parallel fused run of k=1 layers (+ attention row t, + mel emit)
//   attn_rows             windowed monotonic attention rows (networks.py:286-325)
//   embed_rows, pad_rows
#include "oph_internal.h"
#include "oph_device.h"

#include <map>

namespace oph {

// =====================================================================================
// conv_gemm_f32: H[m][n] = bias[n] + sum_{tap} sum_{c<kc} X[src(m,tap)][c] * Wt[n][tap*kc+c]
// 256 threads = 4 waves (2x2), each wave owns a (BM/2)x(BN/2) block of 32x32 MFMA tiles.
// Both operands are K-contiguous ("TN"): tiles are staged global -> regs -> LDS [row][32+4]
// (pad 4 floats => conflict-free ds_read_b128 fragment reads) and double-buffered.
// K order inside an 8-wide chunk is permuted (lanes<32 take k..k+3, lanes>=32 take k+4..k+7,
// MFMA e pairs k+e with k+4+e) identically for A and B, so the sum is unchanged.
// =====================================================================================
// the fields in which the two phases of a paired launch differ travel by value (scalars: they stay in SGPRs; a modified copy of
// the whole GemmArgs would live in scratch memory because of its indexed off[] member)
struct GemmAlt { const float* Wt; const void* Wh; const void* Wl; float* H; int ldw, ntaps; };
__device__ __forceinline__ GemmAlt gemm_alt(const GemmArgs& a) { return GemmAlt{a.Wt, a.Wh, a.Wl, a.H, a.ldw, a.ntaps}; }
template <int BM, int BN>
static __device__ __forceinline__ void conv_gemm_f32_body(const GemmArgs& a, const GemmAlt ph) {
    if (stopped(a.stop_after, a.t)) return;
    constexpr int BK = 32, LD = 36;
    constexpr int AR = BM / 32, BR = BN / 32;     // float4 staging loads per thread
    constexpr int TM = BM / 64, TN = BN / 64;     // 32x32 tiles per wave
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                      // [2][BM*LD]
    float* Bs = smem + 2 * BM * LD;        // [2][BN*LD]
    int* srow_s = (int*)(smem + 2 * (BM + BN) * LD);   // [3][BM] source row per tap (-1 = zeros)

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int MT = (a.M + BM - 1) / BM, NT = (a.N + BN - 1) / BN;
    const int ntiles = MT * NT;
    // XCD-aware bijective remap (blocks b, b+8, .. share an XCD/L2): contiguous tile chunk per XCD
    int id;
    {
        const int bid = blockIdx.x, q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, loc = bid >> 3;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    // grouped ordering: 8 m-tiles x all n-tiles per group keeps A and W panels L2-resident
    constexpr int GM = 8;
    const int width = GM * NT, g = id / width, first_m = g * GM;
    const int gsz = min(MT - first_m, GM);
    const int tm = first_m + (id % width) % gsz, tn = (id % width) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    for (int i = tid; i < BM * ph.ntaps; i += 256) {
        const int tap = i / BM, m = m0 + (i - tap * BM);
        int src = -1;
        if (m < a.M) {
            if (a.mode == 0) {
                const int b = m / a.T, t = m - b * a.T, tt = t + a.off[tap];
                if (tt >= 0 && tt < a.T) src = m + a.off[tap];
            } else {
                const int ip = m / a.Bpad, b = m - ip * a.Bpad;
                if (a.j >= a.need[tap * a.n_out + ip]) src = a.tab[tap * a.n_out + ip] * a.Bpad + b;
            }
        }
        srow_s[i] = src;
    }
    __syncthreads();

    const int lrow = tid >> 3, kq = tid & 7;
    const int kpt = a.kc / BK, nk_all = ph.ntaps * kpt;
    const int ksplit = a.ksplit > 1 ? a.ksplit : 1, split = blockIdx.y;
    const int ks0 = split * nk_all / ksplit, nk = (split + 1) * nk_all / ksplit - ks0;
    // Software pipeline, prefetch distance 2: while tile s is multiplied out of LDS buffer s&1, tile s+1
    // sits in one register set (written to the other LDS buffer at the end of the step) and the global
    // loads of tile s+2 are already in flight into the second register set.  One K-step of MFMAs
    // (0.45 us for a 64x64 tile) does not cover an L2/MALL round trip; two do.
    f32x4 ra0[AR], rb0[BR], ra1[AR], rb1[BR];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    auto load_global = [&](int sl, f32x4 (&ra)[AR], f32x4 (&rb)[BR]) {
        const int s = ks0 + sl;
        const int tap = s / kpt, ko = (s - tap * kpt) * BK + kq * 4;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int src = srow_s[tap * BM + lrow + 32 * i];
            ra[i] = src >= 0 ? *(const f32x4*)(a.X + (size_t)src * a.ldx + ko) : zero4;
        }
#pragma unroll
        for (int i = 0; i < BR; ++i)
            rb[i] = *(const f32x4*)(ph.Wt + (size_t)(n0 + lrow + 32 * i) * ph.ldw + tap * a.kc + ko);
    };
    auto store_lds = [&](int buf, const f32x4 (&ra)[AR], const f32x4 (&rb)[BR]) {
#pragma unroll
        for (int i = 0; i < AR; ++i) *(f32x4*)(As + buf * BM * LD + (lrow + 32 * i) * LD + kq * 4) = ra[i];
#pragma unroll
        for (int i = 0; i < BR; ++i) *(f32x4*)(Bs + buf * BN * LD + (lrow + 32 * i) * LD + kq * 4) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;

    const int wr = w >> 1, wc = w & 1, r32 = lane & 31, kh = lane >> 5;
    auto compute = [&](int buf) {
        const float* Ab = As + buf * BM * LD + (wr * (BM / 2) + r32) * LD + kh * 4;
        const float* Bb = Bs + buf * BN * LD + (wc * (BN / 2) + r32) * LD + kh * 4;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *(const f32x4*)(Ab + i * 32 * LD + kk * 8);
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) bf[jn] = *(const f32x4*)(Bb + jn * 32 * LD + kk * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[jn][e], acc[i][jn], 0, 0, 0);
        }
    };
    load_global(0, ra0, rb0);
    if (nk > 1) load_global(1, ra1, rb1);
    store_lds(0, ra0, rb0);
    __syncthreads();
    // steps are processed in pairs so that the two register sets are addressed statically
    for (int s = 0; s < nk; s += 2) {
        // even step s: LDS buffer 0; set1 holds tile s+1; set0 is free -> prefetch tile s+2
        if (s + 2 < nk) load_global(s + 2, ra0, rb0);
        compute(0);
        if (s + 1 < nk) store_lds(1, ra1, rb1);
        __syncthreads();
        if (s + 1 >= nk) break;
        // odd step s+1: LDS buffer 1; set0 holds tile s+2; set1 is free -> prefetch tile s+3
        if (s + 3 < nk) load_global(s + 3, ra1, rb1);
        compute(1);
        if (s + 2 < nk) store_lds(0, ra0, rb0);
        __syncthreads();
    }
    // C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            const int col = n0 + wc * (BN / 2) + jn * 32 + r32;
            const float bv = split == 0 ? a.bias[col] : 0.f;
            float* Hs = ph.H + (size_t)split * a.split_stride;
            // (a full tile stores without per-row branches: behind a branch the compiler waits for vmcnt(0) -- the bias load, as far
            //  as it can tell -- before EVERY store, i.e. for the previous store's acknowledgement: 64 serialised round trips per wave)
            if (m0 + BM <= a.M) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = m0 + wr * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                    Hs[(size_t)row * a.ldh + col] = acc[i][jn][e] + bv;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = m0 + wr * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                    if (row < a.M) Hs[(size_t)row * a.ldh + col] = acc[i][jn][e] + bv;
                }
            }
        }
}

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void conv_gemm_f32(GemmArgs a) { conv_gemm_f32_body<BM, BN>(a, gemm_alt(a)); }
// Two contractions over the same rows in ONE launch (grid.z picks): the even- and odd-phase halves of a transposed
// convolution (modules.py:209-258) are each too small to fill the chip, and each launch has a fixed ~5 us floor.
// (The second contraction travels as the handful of fields in which it differs: two whole GemmArgs were 336 bytes of kernel arguments,
// and a launch whose arguments exceed 256 bytes measured several us slower -- see HcFusedArgs in oph_internal.h.)
struct GemmPairArgs {
    GemmArgs a;                                                   // phase 0
    const float* Wt2; const void* Wh2; const void* Wl2; float* H2;      // phase 1: its kernel (fp32 / planes) and its raw rows ...
    int ldw2, ntaps2;                                             // ... their row stride and tap count (off[0] is shared)
};
static_assert(sizeof(GemmPairArgs) <= 256, "kernel arguments: four 64-byte lines");
__device__ __forceinline__ GemmAlt pair_phase(const GemmPairArgs& p) {
    return blockIdx.z != 0 ? GemmAlt{p.Wt2, p.Wh2, p.Wl2, p.H2, p.ldw2, p.ntaps2} : gemm_alt(p.a);
}
template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void conv_gemm_f32_pair(GemmPairArgs p) { conv_gemm_f32_body<BM, BN>(p.a, pair_phase(p)); }

int conv_gemm_tile_m(int M, int N) {
    const long long t128 = (long long)((M + 127) / 128) * ((N + 127) / 128);
    return t128 >= 384 ? 128 : 64;
}

template <int BM, int BN>
static void launch_conv_gemm_t(const GemmArgs& a, hipStream_t s) {
    static bool attr_set[64] = {false};          // function attributes are per device
    const size_t lds = (size_t)(2 * (BM + BN) * 36) * 4 + (size_t)3 * BM * 4;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_set[dev & 63]) {
        (void)hipFuncSetAttribute((const void*)conv_gemm_f32<BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set[dev & 63] = true;
    }
    const int MT = (a.M + BM - 1) / BM, NT = (a.N + BN - 1) / BN;
    hipLaunchKernelGGL((conv_gemm_f32<BM, BN>), dim3(MT * NT, a.ksplit > 1 ? a.ksplit : 1), dim3(256), lds, s, a);
}
void launch_conv_gemm(const GemmArgs& a, hipStream_t s) {
    if (conv_gemm_tile_m(a.M, a.N) == 128) launch_conv_gemm_t<128, 128>(a, s);
    else launch_conv_gemm_t<64, 64>(a, s);
}

// =====================================================================================
// conv_gemm_bf16x3: same contract as conv_gemm_f32, but every fp32 operand is split on the fly into
// hi = bf16(x), lo = bf16(x - hi) while it is staged into LDS, and the contraction runs as
//     a.b ~= ah.bh + ah.bl + al.bh          (dropped al.bl <= 2^-18 |a.b|)
// on v_mfma_f32_32x32x16_bf16 (16x the fp32-MFMA rate => ~5x effective) with fp32 accumulation.
// Used for SSRN only (no argmax feedback there); measured error vs the fp32 oracle is reported by
// tests/test_gpu_model.py.  LDS tile = hi and lo planes [row][32 bf16 + 8 pad] (80-B rows =>
// conflict-free ds_read_b128 of the 8-element MFMA fragments).
// =====================================================================================
// The two 16-bit operand formats of the split contraction.  bf16 (8 significant bits per term: hi + lo reproduce 16 bits,
// ~1e-5 relative per product) and fp16 (11 bits per term: hi + lo reproduce 22 of fp32's 24 bits, 2.4e-7 relative per
// product -- below the rounding of an fp32 accumulation over K >= 512 terms, i.e. the same accuracy class as the fp32 MFMA
// kernel; the lo term of a small value falls into fp16's subnormals, whose absolute spacing 2^-24 is far below the
// accumulator's resolution).  Both run at the same MFMA rate.  fp16's range ends at 65504: the operands here are LayerNorm
// outputs, highway mixes of them, mel frames in [0,1] and weights -- orders of magnitude below it.
template <bool F16> struct SplitT { typedef __bf16 T; };
template <> struct SplitT<true> { typedef _Float16 T; };
template <bool F16>
__device__ __forceinline__ void split16(const f32x4& x, typename SplitT<F16>::T (&hi)[4], typename SplitT<F16>::T (&lo)[4]) {
    typedef typename SplitT<F16>::T H;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const H h = (H)x[e];
        hi[e] = h;
        lo[e] = (H)(x[e] - (float)h);
    }
}

// The weights are split ONCE (launch_split_bf16 at load time) into hi / lo bf16 planes; only the activations are split
// in the kernel, while they are staged.  Software pipeline as conv_gemm_f32 (prefetch distance 2): one K-step of bf16
// MFMAs (~0.3 us for a 128x128 tile) covers nothing of an HBM round trip, two register sets in flight do.
template <bool F16>
__global__ void split16_kernel(const float* w, typename SplitT<F16>::T* hi, typename SplitT<F16>::T* lo, size_t n) {
    typedef typename SplitT<F16>::T H;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const float x = w[i]; const H h = (H)x; hi[i] = h; lo[i] = (H)(x - (float)h); }
}
void launch_split_bf16(const float* w, void* hi, void* lo, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(split16_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, (__bf16*)hi, (__bf16*)lo, n);
}
void launch_split_f16(const float* w, void* hi, void* lo, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(split16_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, (_Float16*)hi, (_Float16*)lo, n);
}

template <bool F16, class V8>
static __device__ __forceinline__ f32x16 mfma16(const V8& x, const V8& y, const f32x16& c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0);
}
template <int BM, int BN, bool TAB, bool F16>      // TAB: table-mapped rows (decoder cone); a separate instance so that the dense one carries no row table.  F16: fp16 terms instead of bf16
static __device__ __forceinline__ void conv_gemm_bf16x3_body(const GemmArgs& a, const GemmAlt ph) {
    typedef typename SplitT<F16>::T H16;
    typedef H16 h16x8 __attribute__((ext_vector_type(8)));
    if (stopped(a.stop_after, a.t)) return;
    // LDS rows are 32 bf16 = four 16-byte chunks, unpadded, with the chunk index XOR-ed by (row >> 2) & 3: the 8-byte staging
    // stores of a wave (4 rows x 8 lanes per pass) then fall into four disjoint 16-bank ranges, and the 16-byte fragment reads
    // of 16 consecutive rows into 16 distinct 4-bank groups -- both conflict-free (the padded [32 + 8] rows had two-way
    // conflicts on every staging store: a third of the LDS cycles, profiles/r02_ssrn_pmc.sh)
    constexpr int BK = 32, LDH = 32;
    constexpr int AR = BM / 32, BR = BN / 32;
    constexpr int TM = BM / 64, TN = BN / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    H16* Ah = (H16*)smem;                          // [2][BM*LDH]
    H16* Al = Ah + 2 * BM * LDH;
    // The weight planes never pass through registers: global_load_lds_dwordx4 copies 16 bytes per lane straight into LDS
    // (lane l's bytes land at the wave's base + 16 l, profiles/glds_probe.hip), two K-steps ahead, into a 3-slot ring; the
    // chunk swizzle is applied on the global side.  A: 2 buffers x 2 planes; B: 3 slots x 2 planes -- for a 128x128 tile
    // exactly half a CU's LDS, so two workgroups stay resident per CU (no source-row table in LDS for the same reason).
    H16* Bh = Al + 2 * BM * LDH;                   // [3][BN*LDH]
    H16* Bl = Bh + 3 * BN * LDH;

    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int MT = (a.M + BM - 1) / BM, NT = (a.N + BN - 1) / BN;
    const int ntiles = MT * NT;
    int id;
    {
        const int bid = blockIdx.x, q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, loc = bid >> 3;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    constexpr int GM = 8;
    const int width = GM * NT, g = id / width, first_m = g * GM;
    const int gsz = min(MT - first_m, GM);
    const int tm = first_m + (id % width) % gsz, tn = (id % width) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    const int lrow = tid >> 3, kq = tid & 7;
    int mt[AR];                                    // dense rows: time index of this thread's staging rows (-1: past the last row)
    int srow[TAB ? 3 : 1][AR];                     // table rows (decoder cone, mode 1): source row per tap (-1 = zeros)
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + lrow + 32 * i;
        mt[i] = (!TAB && m < a.M) ? m % a.T : -1;
        if constexpr (TAB) {
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) {
                srow[tap][i] = -1;
                if (m < a.M && tap < ph.ntaps) {
                    const int ip = m / a.Bpad, b = m - ip * a.Bpad;
                    if (a.j >= a.need[tap * a.n_out + ip]) srow[tap][i] = a.tab[tap * a.n_out + ip] * a.Bpad + b;
                }
            }
        }
    }
    const int kpt = a.kc / BK, nk_all = ph.ntaps * kpt;
    const int ksplit = a.ksplit > 1 ? a.ksplit : 1, split = blockIdx.y;
    const int ks0 = split * nk_all / ksplit, nk = (split + 1) * nk_all / ksplit - ks0;
    const H16* Wh = (const H16*)ph.Wh; const H16* Wl = (const H16*)ph.Wl;
    const int nprod = a.nprod > 0 ? a.nprod : 3;
    f32x4 ra0[AR], ra1[AR];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    constexpr int NDMA = BN * 4 / 256;             // 16-byte chunks of one plane's K-step per thread
    auto dma_b = [&](int sl) {                     // weight planes of K-step sl -> ring slot sl % 3
        const int s = ks0 + sl;
        const int tap = s / kpt, kb = (s - tap * kpt) * BK, slot = sl % 3;
#pragma unroll
        for (int q = 0; q < NDMA; ++q) {
            const int id = q * 256 + tid, row = id >> 2, pos = id & 3;
            const size_t o = (size_t)(n0 + row) * ph.ldw + tap * a.kc + kb + ((pos ^ ((row >> 2) & 3)) << 3);
            const int base = slot * BN * LDH + (q * 256 + w * 64) * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Wh + o),
                                             (__attribute__((address_space(3))) void*)(Bh + base), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Wl + o),
                                             (__attribute__((address_space(3))) void*)(Bl + base), 16, 0, 0);
        }
    };
    auto load_global = [&](int sl, f32x4 (&ra)[AR]) {
        const int s = ks0 + sl;
        const int tap = s / kpt, ko = (s - tap * kpt) * BK + kq * 4;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            if constexpr (TAB) {
                const int src = tap == 0 ? srow[0][i] : (tap == 1 ? srow[1][i] : srow[2][i]);
                ra[i] = src >= 0 ? *(const f32x4*)(a.X + (size_t)src * a.ldx + ko) : zero4;
            } else {
                const int tt = mt[i] + a.off[tap];
                const bool in = mt[i] >= 0 && tt >= 0 && tt < a.T;
                ra[i] = in ? *(const f32x4*)(a.X + (size_t)(m0 + lrow + 32 * i + a.off[tap]) * a.ldx + ko) : zero4;
            }
        }
    };
    auto store_lds = [&](int buf, const f32x4 (&ra)[AR]) {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            H16 hi[4], lo[4];
            split16<F16>(ra[i], hi, lo);
            typedef H16 h16x4 __attribute__((ext_vector_type(4)));
            const int o = buf * BM * LDH + (lrow + 32 * i) * LDH + (((kq >> 1) ^ ((lrow >> 2) & 3)) << 3) + ((kq & 1) << 2);
            *(h16x4*)(Ah + o) = h16x4{hi[0], hi[1], hi[2], hi[3]};
            *(h16x4*)(Al + o) = h16x4{lo[0], lo[1], lo[2], lo[3]};
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;

    // 32x32x16 bf16 fragment: lane l holds row (l&31), k = 8*(l>>5) .. +7 of the 16-wide chunk
    const int wr = w >> 1, wc = w & 1, r32 = lane & 31, kh = lane >> 5;
    auto compute = [&](int buf, int bslot) {
        const int ao = buf * BM * LDH + (wr * (BM / 2) + r32) * LDH;
        const int bo = bslot * BN * LDH + (wc * (BN / 2) + r32) * LDH;
        const int swr = (r32 >> 2) & 3;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            h16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = *(const h16x8*)(Ah + ao + i * 32 * LDH + (((kc * 2 + kh) ^ swr) << 3));
                al[i] = *(const h16x8*)(Al + ao + i * 32 * LDH + (((kc * 2 + kh) ^ swr) << 3));
            }
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) {
                bh[jn] = *(const h16x8*)(Bh + bo + jn * 32 * LDH + (((kc * 2 + kh) ^ swr) << 3));
                bl[jn] = *(const h16x8*)(Bl + bo + jn * 32 * LDH + (((kc * 2 + kh) ^ swr) << 3));
            }
            // product by product over all tiles: with four tiles per wave consecutive MFMAs never share an accumulator (three
            // back-to-back MFMAs on one accumulator serialise on its latency: 30 % of the wave cycles were issue stalls,
            // profiles/r02_ssrn_pmc.sh; own accumulators for the small terms of the single-tile instance were measured slower:
            // 168 instead of 116 registers cost it a workgroup of occupancy)
            // (a.nprod: measurement only -- 2 drops the weights' lo term, 1 the activations' as well)
            if (nprod >= 2) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) {
                    acc[i][jn] = mfma16<F16>(al[i], bh[jn], acc[i][jn]);
                }
            }
            if (nprod >= 3) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) {
                    acc[i][jn] = mfma16<F16>(ah[i], bl[jn], acc[i][jn]);
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn)
                    acc[i][jn] = mfma16<F16>(ah[i], bh[jn], acc[i][jn]);
        }
    };
    // every VMEM operation of a step is issued before its compute; they complete in order, so "at most the operations of
    // THIS step outstanding" means the previous step's (the weight planes the next compute reads) have landed
    auto wait_older = [&](bool issued_now) {
        if (issued_now) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AR + 2 * NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    load_global(0, ra0);
    dma_b(0);
    if (nk > 1) { load_global(1, ra1); dma_b(1); }
    store_lds(0, ra0);
    wait_older(nk > 1);
    __syncthreads();
    for (int s = 0; s < nk; s += 2) {
        if (s + 2 < nk) { load_global(s + 2, ra0); dma_b(s + 2); }
        compute(0, s % 3);
        if (s + 1 < nk) store_lds(1, ra1);
        wait_older(s + 2 < nk);
        __syncthreads();
        if (s + 1 >= nk) break;
        if (s + 3 < nk) { load_global(s + 3, ra1); dma_b(s + 3); }
        compute(1, (s + 1) % 3);
        if (s + 2 < nk) store_lds(0, ra0);
        wait_older(s + 3 < nk);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            const int col = n0 + wc * (BN / 2) + jn * 32 + r32;
            const float bv = split == 0 ? a.bias[col] : 0.f;
            float* Hs = ph.H + (size_t)split * a.split_stride;
            // (a full tile stores without per-row branches: behind a branch the compiler waits for vmcnt(0) -- the bias load, as far
            //  as it can tell -- before EVERY store, i.e. for the previous store's acknowledgement: 64 serialised round trips per wave)
            if (m0 + BM <= a.M) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = m0 + wr * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                    Hs[(size_t)row * a.ldh + col] = acc[i][jn][e] + bv;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = m0 + wr * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                    if (row < a.M) Hs[(size_t)row * a.ldh + col] = acc[i][jn][e] + bv;
                }
            }
        }
}

template <int BM, int BN, bool F16>
__global__ __launch_bounds__(256, 2) void conv_gemm_bf16x3(GemmArgs a) { conv_gemm_bf16x3_body<BM, BN, false, F16>(a, gemm_alt(a)); }
template <int BM, int BN, bool F16>
__global__ __launch_bounds__(256, 2) void conv_gemm_bf16x3_tab(GemmArgs a) { conv_gemm_bf16x3_body<BM, BN, true, F16>(a, gemm_alt(a)); }
template <int BM, int BN, bool F16>
__global__ __launch_bounds__(256, 2) void conv_gemm_bf16x3_pair(GemmPairArgs p) { conv_gemm_bf16x3_body<BM, BN, false, F16>(p.a, pair_phase(p)); }
template <int BM, int BN>
static void launch_conv_gemm_pair_t(const GemmArgs& a0, const GemmArgs& a1, int prec, hipStream_t s) {
    static bool attr_set[3][64] = {{false}};
    const size_t lds = prec ? (size_t)((2 * BM + 3 * BN) * 2 * 32) * 2 : (size_t)(2 * (BM + BN) * 36) * 4 + (size_t)3 * BM * 4;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_set[prec][dev & 63]) {
        if (prec == 2) (void)hipFuncSetAttribute((const void*)conv_gemm_bf16x3_pair<BM, BN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        else if (prec) (void)hipFuncSetAttribute((const void*)conv_gemm_bf16x3_pair<BM, BN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        else (void)hipFuncSetAttribute((const void*)conv_gemm_f32_pair<BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set[prec][dev & 63] = true;
    }
    const int MT = (a0.M + BM - 1) / BM, NT = (a0.N + BN - 1) / BN;
    GemmPairArgs p{};
    p.a = a0; p.Wt2 = a1.Wt; p.Wh2 = a1.Wh; p.Wl2 = a1.Wl; p.H2 = a1.H; p.ldw2 = a1.ldw; p.ntaps2 = a1.ntaps;      // (everything else is shared: same rows, N, kc, off[0], stop word)
    if (prec == 2) hipLaunchKernelGGL((conv_gemm_bf16x3_pair<BM, BN, true>), dim3(MT * NT, 1, 2), dim3(256), lds, s, p);
    else if (prec) hipLaunchKernelGGL((conv_gemm_bf16x3_pair<BM, BN, false>), dim3(MT * NT, 1, 2), dim3(256), lds, s, p);
    else hipLaunchKernelGGL((conv_gemm_f32_pair<BM, BN>), dim3(MT * NT, 1, 2), dim3(256), lds, s, p);
}
// same M and N, no split-K; prec: 0 fp32 MFMA, 1 split-bf16 x3, 2 split-fp16 x3 (both need Wh / Wl in that format)
void launch_conv_gemm_pair(const GemmArgs& a0, const GemmArgs& a1, int prec, hipStream_t s) {
    if (conv_gemm_tile_m(a0.M, a0.N) == 128) launch_conv_gemm_pair_t<128, 128>(a0, a1, prec, s);
    else launch_conv_gemm_pair_t<64, 64>(a0, a1, prec, s);
}
template <int BM, int BN, bool F16>
static void launch_conv_gemm_bf16x3_t(const GemmArgs& a, hipStream_t s) {
    static bool attr_set[64] = {false};
    const size_t lds = (size_t)((2 * BM + 3 * BN) * 2 * 32) * 2;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_set[dev & 63]) {
        (void)hipFuncSetAttribute((const void*)conv_gemm_bf16x3<BM, BN, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set[dev & 63] = true;
    }
    const int MT = (a.M + BM - 1) / BM, NT = (a.N + BN - 1) / BN;
    if (a.mode == 1) {
        static bool tab_set[64] = {false};
        if (!tab_set[dev & 63]) { (void)hipFuncSetAttribute((const void*)conv_gemm_bf16x3_tab<BM, BN, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); tab_set[dev & 63] = true; }
        hipLaunchKernelGGL((conv_gemm_bf16x3_tab<BM, BN, F16>), dim3(MT * NT, a.ksplit > 1 ? a.ksplit : 1), dim3(256), lds, s, a);
    } else
        hipLaunchKernelGGL((conv_gemm_bf16x3<BM, BN, F16>), dim3(MT * NT, a.ksplit > 1 ? a.ksplit : 1), dim3(256), lds, s, a);
}
void launch_conv_gemm_bf16x3(const GemmArgs& a, hipStream_t s) {     // needs a.Wh / a.Wl; a.f16: they are fp16 planes
    const bool big = conv_gemm_tile_m(a.M, a.N) == 128;
    if (a.f16) { if (big) launch_conv_gemm_bf16x3_t<128, 128, true>(a, s); else launch_conv_gemm_bf16x3_t<64, 64, true>(a, s); }
    else { if (big) launch_conv_gemm_bf16x3_t<128, 128, false>(a, s); else launch_conv_gemm_bf16x3_t<64, 64, false>(a, s); }
}

// =====================================================================================
// ln_rows<NV>: one wavefront per row; the row (<= NV*256 channels) lives in registers.
// [TF-sem] tf.contrib.layers.layer_norm: mean, biased variance, eps 1e-12 (modules.py:65).
//   conv : y = act(LN(h))                                     (modules.py:137-139)
//   hc   : g = sigmoid(LN1(h[:C])), u = LN2(h[C:]), y = g*u + (1-g)*x   (modules.py:194-203)
// =====================================================================================
// (hardware exp / rcp / rsq forms, ~1 ulp each: the IEEE expansions made the 1024-channel rows instruction-bound -- 45 % of
//  the wave cycles issuing, profiles/r02_ssrn_pmc.sh)
// PRE: gamma / beta were requested by the caller before the row's statistics (gv / bv hold them): rows of <= 512 channels, where the
// 16 extra registers cost no occupancy and the L2 round trip between the reductions and the stores was a sixth of the kernel
// FULL: the row has exactly NV * 256 channels -- no per-element guards.  Otherwise (NV - 1) * 256 < C < NV * 256 (launch_epilogue
// picks NV that way): only the last 256-channel vector is guarded
template <int NV, bool PRE = false, bool FULL = false>
__device__ __forceinline__ void ln_vec(f32x4 (&x)[NV], int C, int lane, const float* gam, const float* bet, int nonorm,
                                       const f32x4* gvp = nullptr, const f32x4* bvp = nullptr) {
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += (FULL || v < NV - 1 || (v * 64 + lane) * 4 + e < C) ? x[v][e] : 0.f;
    const float mean = nonorm ? 0.f : wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float dlt = (FULL || v < NV - 1 || (v * 64 + lane) * 4 + e < C) ? x[v][e] - mean : 0.f;
            x[v][e] = dlt;
            q += dlt * dlt;
        }
    const float var = wave_sum(q) / (float)C;
    const float rstd = nonorm ? 1.0f : fast_rsqrt(var + LN_EPS);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c = (v * 64 + lane) * 4;
        if ((FULL || v < NV - 1 || c < C)) {
            f32x4 gv, bv;
            if constexpr (PRE) { gv = gvp[v]; bv = bvp[v]; }
            else { gv = *(const f32x4*)(gam + c); bv = *(const f32x4*)(bet + c); }
#pragma unroll
            for (int e = 0; e < 4; ++e) x[v][e] = x[v][e] * rstd * gv[e] + bv[e];
        }
    }
}

template <int NV, bool FULL>
__device__ __forceinline__ void ln_rows_body(const EpiArgs& a) {
    if (stopped(a.stop_after, a.t)) return;
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= a.M) return;
    size_t orow = (size_t)m;               // output row
    if (a.out_T > 0) {
        const int b = m / a.out_T, u = m - b * a.out_T;
        if (u < a.keep_lo || u >= a.keep_hi) return;         // a margin row of a streamed chunk: its value is not valid
        orow = (size_t)b * (size_t)a.out_bs + (size_t)(a.out_t0 + u);
    }
    const int C = a.C;
    const float* h = a.H + (size_t)m * a.ldh;
    f32x4 x[NV];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c = (v * 64 + lane) * 4;
        x[v] = (FULL || v < NV - 1 || c < C) ? *(const f32x4*)(h + c) : zero4;
        for (int sp = 1; sp < a.nsplit; ++sp)
            if ((FULL || v < NV - 1 || c < C)) x[v] += *(const f32x4*)(h + sp * a.split_stride + c);
    }
    // everything the row needs is requested before the first reduction (the highway's second half and residual row; gamma / beta of
    // narrow rows): one memory round trip per row instead of three dependent ones
    constexpr bool PRE = NV <= 2;
    f32x4 gv1[NV], bv1[NV], gv2[NV], bv2[NV], u[NV], xres[NV];
    const bool hc = a.mode == PRE_HC;
    if (hc) {
        size_t rrow = m;
        if (a.restab) rrow = (size_t)a.restab[m / a.Bpad] * a.Bpad + (m % a.Bpad);
        const float* xr = a.Xres + rrow * a.ldres;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = (v * 64 + lane) * 4;
            u[v] = (FULL || v < NV - 1 || c < C) ? *(const f32x4*)(h + C + c) : zero4;
            for (int sp = 1; sp < a.nsplit; ++sp)
                if ((FULL || v < NV - 1 || c < C)) u[v] += *(const f32x4*)(h + sp * a.split_stride + C + c);
            xres[v] = (FULL || v < NV - 1 || c < C) ? *(const f32x4*)(xr + c) : zero4;
        }
    }
    if constexpr (PRE) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = (v * 64 + lane) * 4;
            if ((FULL || v < NV - 1 || c < C)) {
                gv1[v] = *(const f32x4*)(a.g1 + c); bv1[v] = *(const f32x4*)(a.b1 + c);
                if (hc) { gv2[v] = *(const f32x4*)(a.g2 + c); bv2[v] = *(const f32x4*)(a.b2 + c); }
            }
        }
    }
    ln_vec<NV, PRE, FULL>(x, C, lane, a.g1, a.b1, a.nonorm, gv1, bv1);
    const float* lg = nullptr;             // this row's LCC gate vector
    if (a.lcc) lg = a.lcc + (size_t)a.lcc_ids[a.lcc_T > 0 ? m / a.lcc_T : m % a.Bpad] * C;
    if (hc) {
        ln_vec<NV, PRE, FULL>(u, C, lane, a.g2, a.b2, a.nonorm, gv2, bv2);
        if (lg) {
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = (v * 64 + lane) * 4 + e;
                    if ((FULL || v < NV - 1 || c < C)) u[v][e] *= lg[c];
                }
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = (v * 64 + lane) * 4;
            if ((FULL || v < NV - 1 || c < C)) {
                const f32x4 xv = xres[v];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gte = fast_sigmoid(x[v][e]);
                    x[v][e] = gte * u[v][e] + (1.0f - gte) * xv[e];
                }
            }
        }
    } else if (lg) {
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = (v * 64 + lane) * 4 + e;
                const float gt = (FULL || v < NV - 1 || c < C) ? lg[c] : 0.f;
                x[v][e] = a.act == ACT_SIGMOID ? fast_sigmoid(gt * x[v][e]) : gt * fast_act(x[v][e], a.act);
            }
    } else {
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) x[v][e] = fast_act(x[v][e], a.act);
    }
    float* y = a.Y + orow * a.ldy;
    const bool vec_ok = (a.ldy & 3) == 0;
    if (a.planes) {
        // the next layer's contraction reads its operand as fp16 hi / lo planes (plane_gemm, oph_planegemm.hip): the row is split
        // here, once, instead of in that kernel's K loop once per tap; K-blocked [ypad / 32][M][32], pad channels zero
        typedef _Float16 h16x4_ __attribute__((ext_vector_type(4)));
        _Float16* ph = (_Float16*)a.Yh; _Float16* pl = (_Float16*)a.Yl;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = (v * 64 + lane) * 4;
            if (c < a.ypad) {
                h16x4_ hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xv = (FULL || v < NV - 1 || c + e < C) ? x[v][e] : 0.f;
                    hi[e] = (_Float16)xv;
                    lo[e] = (_Float16)(xv - (float)hi[e]);
                }
                const size_t o = ((size_t)(c >> 5) * a.M + m) * 32 + (c & 31);
                *(h16x4_*)(ph + o) = hi;
                *(h16x4_*)(pl + o) = lo;
            }
        }
    }
    const bool sig = !a.planes && a.done_sig;                              // (planes: the two pointers are the planes, EpiArgs)
    const int pos = sig ? m / a.Bpad : -1;
    const bool coh = sig && (pos == a.coh0 || pos == a.coh1);      // rows a running dec_loop reads
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c = (v * 64 + lane) * 4;
        if ((FULL || v < NV - 1 || c + 3 < C) && vec_ok) {
            if (coh) st_coherent(y + c, x[v]);
            else *(f32x4*)(y + c) = x[v];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if ((FULL || v < NV - 1 || c + e < C)) y[c + e] = x[v][e];
        }
    }
    int ctot = C;
    if (a.spk_table) {
        const int u = a.spk_T > 0 ? m / a.spk_T : m % a.Bpad;
        const int id = a.spk_ids[u];
        for (int c = lane; c < a.spk_dim; c += 64)
            y[C + c] = id == 0 ? 0.f : a.spk_table[(size_t)id * a.spk_dim + c];   // row 0 zeroed at lookup (modules.py:38-40)
        ctot += a.spk_dim;
    }
    for (int c = ctot + lane; c < a.ypad; c += 64) y[c] = 0.f;
}

template <int NV, bool FULL = false>
__global__ __launch_bounds__(256) void ln_rows(EpiArgs a) {
    ln_rows_body<NV, FULL>(a);
    const bool sig = !a.planes && a.done_sig;
    const int pos_b = sig ? (int)(blockIdx.x * 4) / a.Bpad : -1;       // Bpad % 4 == 0: a workgroup's 4 rows share a position
    if (sig && (pos_b == a.coh0 || pos_b == a.coh1)) {
        // this launch writes a level of a cone: once the tap rows have left (write-through stores, no fence), one lane
        // raises the word the decoder loop kernel polls for that level (instead of a signalling kernel behind the cone)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned old = atomicAdd(a.done_count, 1u);
            if (old + 1u == a.done_target) {
                __hip_atomic_fetch_max(a.done_sig, a.done_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (a.done_stamp) *a.done_stamp = wall_clock64();      // diagnostics (OPH_RUN_STAMPS): when this level was complete
            }
        }
    }
}

void launch_epilogue(const EpiArgs& a, hipStream_t s) {
    const dim3 grid((a.M + 3) / 4), block(256);
    // rows of exactly 256 / 512 / 1024 channels (every LayerNorm of the nets but the 80- and 1025-channel ends) take the instances
    // without per-element guards: with them the 1024-channel instance needed 308 registers -- one wave per SIMD
    if (a.C == 256) hipLaunchKernelGGL((ln_rows<1, true>), grid, block, 0, s, a);
    else if (a.C == 512) hipLaunchKernelGGL((ln_rows<2, true>), grid, block, 0, s, a);
    else if (a.C == 1024) hipLaunchKernelGGL((ln_rows<4, true>), grid, block, 0, s, a);
    else if (a.C <= 256) hipLaunchKernelGGL(ln_rows<1>, grid, block, 0, s, a);
    else if (a.C <= 512) hipLaunchKernelGGL(ln_rows<2>, grid, block, 0, s, a);
    else if (a.C <= 768) hipLaunchKernelGGL(ln_rows<3>, grid, block, 0, s, a);
    else if (a.C <= 1024) hipLaunchKernelGGL(ln_rows<4>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(ln_rows<5>, grid, block, 0, s, a);   // <= 1280 (full_dim 1025)
}

// =====================================================================================
// dec_layer16: one decoder layer for a tile of 16 utterances at ONE time step.
//   grid = (Npad/16, Bpad/16), 16 waves.  Every workgroup redundantly runs the cheap
//   prologue (the previous layer's LayerNorm / gate / highway mix for its 16 rows) into LDS,
//   gathers the dilated taps x[t-2r], x[t-r] from the layer's history rows, then computes a
//   16x16 output slice on v_mfma_f32_16x16x4_f32 with K split round-robin over the 16 waves;
//   weight fragments (Wt is [n][k], k contiguous) are fetched straight from L2 into
//   registers and are issued BEFORE the prologue so their latency hides behind it.
//   Output = raw conv rows (bias added); the consumer kernel applies this layer's LN.
// =====================================================================================
constexpr int DEC_WAVES = 16;     // waves per workgroup (prologue rows and K-chunks are split over them)
constexpr int DEC_RPW = 16 / DEC_WAVES;   // prologue rows per wave
constexpr int DEC_PF = 48 / DEC_WAVES;    // 16-wide k-chunks prefetched per wave per pass (K <= 768 in one pass)

// Latency engineering (profiles/r01): a step is ~25 dependent launches, so what matters is the
// number of dependent memory round trips inside each one.  Everything that does not depend on
// computed data -- weight fragments, the 16 raw rows, residual rows, dilated-tap rows, LN
// gamma/beta, bias, the stop flag -- is requested up front in ONE batch (rows unrolled 4-wide
// per wave), the LayerNorm reductions of the 4 rows are interleaved, and the stop flag only
// predicates the final stores instead of gating the kernel.
template <int NV, int PRE, int NTAPS, bool LCC>      // LCC: separate instantiation, the default critical-path kernel stays as it was
__global__ __launch_bounds__(64 * DEC_WAVES) void dec_layer16(DecArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Ktot = NTAPS * a.kc, ldxs = Ktot + 4;
    float* xs = smem;                 // [16][ldxs]
    float* part = smem + 16 * ldxs;   // [DEC_WAVES][256]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n0 = blockIdx.x * 16, row0 = blockIdx.y * 16;
    const int r16 = lane & 15, kq = lane >> 4;
    const int nchunks = Ktot >> 4;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    constexpr int KV = 2;             // tap vectors per lane per row: kc <= 512

    // ---- 1. every independent global request, issued back to back
    const float* wrow = a.Wt + (size_t)(n0 + r16) * a.ldw + kq * 4;
    f32x4 bfrag[DEC_PF];
#pragma unroll
    for (int i = 0; i < DEC_PF; ++i) {
        const int c = w + DEC_WAVES * i;
        bfrag[i] = c < nchunks ? *(const f32x4*)(wrow + c * 16) : zero4;
    }
    const int stop_v = a.stop_after ? *a.stop_after : 0x7fffffff;
    const float bias_v = a.bias[n0 + (tid & 15)];
    f32x4 x[DEC_RPW][NV], u[DEC_RPW][NV], xr[DEC_RPW][NV], tp0[DEC_RPW][KV], tp1[DEC_RPW][KV];
    f32x4 g1v[NV], b1v[NV], g2v[NV], b2v[NV];
#pragma unroll
    for (int rr = 0; rr < DEC_RPW; ++rr) {
        const int grow = row0 + DEC_RPW * w + rr;
        const float* sp = a.src + (size_t)grow * a.ldsrc;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = (v * 64 + lane) * 4;
            x[rr][v] = c < a.cin ? *(const f32x4*)(sp + c) : zero4;
            if (PRE == PRE_HC) {
                u[rr][v] = c < a.cin ? *(const f32x4*)(sp + a.cin + c) : zero4;
                xr[rr][v] = c < a.cin ? *(const f32x4*)(a.xres + (size_t)grow * a.ldres + c) : zero4;
            }
        }
        if (NTAPS == 3) {
#pragma unroll
            for (int v = 0; v < KV; ++v) {
                const int c = (v * 64 + lane) * 4;
                tp0[rr][v] = (a.tap0 && c < a.kc) ? *(const f32x4*)(a.tap0 + (size_t)grow * a.ldtap + c) : zero4;
                tp1[rr][v] = (a.tap1 && c < a.kc) ? *(const f32x4*)(a.tap1 + (size_t)grow * a.ldtap + c) : zero4;
            }
        }
    }
    if (PRE != PRE_COPY) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = (v * 64 + lane) * 4;
            g1v[v] = c < a.cin ? *(const f32x4*)(a.g1 + c) : zero4;
            b1v[v] = c < a.cin ? *(const f32x4*)(a.b1 + c) : zero4;
            if (PRE == PRE_HC) {
                g2v[v] = c < a.cin ? *(const f32x4*)(a.g2 + c) : zero4;
                b2v[v] = c < a.cin ? *(const f32x4*)(a.b2 + c) : zero4;
            }
        }
    }
    int cat_id[DEC_RPW] = {};
    if (a.ccat > 0) {
#pragma unroll
        for (int rr = 0; rr < DEC_RPW; ++rr) {
            const int grow = row0 + DEC_RPW * w + rr;
            cat_id[rr] = a.cat_ids[grow < a.B ? grow : 0];
        }
    }

    // ---- 2. prologue math: LayerNorm(s) of 4 rows interleaved, gate / activation
    if (PRE != PRE_COPY) {
        const float invc = 1.0f / (float)a.cin;
        auto ln4 = [&](f32x4 (&z)[DEC_RPW][NV], const f32x4 (&gv)[NV], const f32x4 (&bv)[NV]) {
            float s[DEC_RPW], q[DEC_RPW];
#pragma unroll
            for (int rr = 0; rr < DEC_RPW; ++rr) {
                s[rr] = 0.f;
#pragma unroll
                for (int v = 0; v < NV; ++v) s[rr] += z[rr][v][0] + z[rr][v][1] + z[rr][v][2] + z[rr][v][3];
            }
#pragma unroll
            for (int rr = 0; rr < DEC_RPW; ++rr) s[rr] = wave_sum(s[rr]);
#pragma unroll
            for (int rr = 0; rr < DEC_RPW; ++rr) {
                const float mean = a.nonorm ? 0.f : s[rr] * invc;
                q[rr] = 0.f;
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float dlt = ((v * 64 + lane) * 4 + e < a.cin) ? z[rr][v][e] - mean : 0.f;
                        z[rr][v][e] = dlt;
                        q[rr] += dlt * dlt;
                    }
            }
#pragma unroll
            for (int rr = 0; rr < DEC_RPW; ++rr) q[rr] = wave_sum(q[rr]);
#pragma unroll
            for (int rr = 0; rr < DEC_RPW; ++rr) {
                const float rstd = a.nonorm ? 1.0f : 1.0f / sqrtf(q[rr] * invc + LN_EPS);
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int e = 0; e < 4; ++e) z[rr][v][e] = z[rr][v][e] * rstd * gv[v][e] + bv[v][e];
            }
        };
        ln4(x, g1v, b1v);
        f32x4 lgv[DEC_RPW][NV];          // LCC gate of the previous layer for this wave's rows
        if (LCC) {
#pragma unroll
            for (int rr = 0; rr < DEC_RPW; ++rr) {
                const int grow = row0 + DEC_RPW * w + rr;
                const float* lgp = a.lcc + (size_t)a.lcc_ids[grow < a.B ? grow : 0] * a.cin;
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const int c = (v * 64 + lane) * 4;
                    lgv[rr][v] = c < a.cin ? *(const f32x4*)(lgp + c) : zero4;
                }
            }
        }
        if (PRE == PRE_HC) {
            ln4(u, g2v, b2v);
#pragma unroll
            for (int rr = 0; rr < DEC_RPW; ++rr)
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float gte = sigmoidf_(x[rr][v][e]);
                        const float tr = LCC ? lgv[rr][v][e] * u[rr][v][e] : u[rr][v][e];
                        x[rr][v][e] = gte * tr + (1.0f - gte) * xr[rr][v][e];
                    }
        } else {
#pragma unroll
            for (int rr = 0; rr < DEC_RPW; ++rr)
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float av = apply_act(x[rr][v][e], a.act);
                        x[rr][v][e] = LCC ? lgv[rr][v][e] * av : av;
                    }
        }
    }
    // ---- 3. stage the 16 x Ktot operand in LDS: [taps (oldest first) | current]
    const int cur = (NTAPS - 1) * a.kc;
#pragma unroll
    for (int rr = 0; rr < DEC_RPW; ++rr) {
        float* xrow = xs + (DEC_RPW * w + rr) * ldxs;
#pragma unroll
        for (int v = 0; v < KV; ++v) {
            const int c = (v * 64 + lane) * 4;
            if (c < a.kc) {
                f32x4 val = zero4;
                if (v < NV) {
                    val = x[rr][v < NV ? v : 0];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (c + e >= a.cin) val[e] = 0.f;
                }
                *(f32x4*)(xrow + cur + c) = val;
                if (NTAPS == 3) {
                    *(f32x4*)(xrow + c) = tp0[rr][v];
                    *(f32x4*)(xrow + a.kc + c) = tp1[rr][v];
                }
            }
        }
        if (a.ccat > 0) {
            for (int c = lane; c < a.ccat; c += 64)
                xrow[cur + a.cin + c] = cat_id[rr] == 0 ? 0.f : a.cat_table[(size_t)cat_id[rr] * a.ccat + c];
        }
    }
    __syncthreads();
    const bool live = a.t <= stop_v;
    if (blockIdx.x == 0 && a.xstore && live) {     // publish x[t] (this layer's input) for later steps / residual
        for (int i = tid * 4; i < 16 * a.kc; i += 256 * DEC_WAVES) {
            const int row = i / a.kc, c = i - row * a.kc;
            *(f32x4*)(a.xstore + (size_t)(row0 + row) * a.ldstore + c) = *(const f32x4*)(xs + row * ldxs + cur + c);
        }
    }

    // ---- 4. 16x16 slice, K split over waves
    f32x4 acc0 = zero4, acc1 = zero4;
    const float* xa = xs + r16 * ldxs + kq * 4;
    for (int base = 0; base < nchunks; base += DEC_WAVES * DEC_PF) {
        if (base > 0) {
#pragma unroll
            for (int i = 0; i < DEC_PF; ++i) {
                const int c = base + w + DEC_WAVES * i;
                bfrag[i] = c < nchunks ? *(const f32x4*)(wrow + c * 16) : zero4;
            }
        }
#pragma unroll
        for (int i = 0; i < DEC_PF; ++i) {
            const int c = base + w + DEC_WAVES * i;
            if (c < nchunks) {
                const f32x4 av = *(const f32x4*)(xa + c * 16);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bfrag[i][0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bfrag[i][1], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bfrag[i][2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bfrag[i][3], acc1, 0, 0, 0);
            }
        }
    }
    // C/D layout of 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
    for (int e = 0; e < 4; ++e) part[w * 256 + (kq * 4 + e) * 16 + r16] = acc0[e] + acc1[e];
    __syncthreads();
    if (live && tid < 256) {
        const int row = tid >> 4, col = tid & 15;
        float v = bias_v;
#pragma unroll
        for (int ww = 0; ww < DEC_WAVES; ++ww) v += part[ww * 256 + tid];
        a.H[(size_t)(row0 + row) * a.ldh + n0 + col] = v;
    }
}

template <int NV, int PRE>
static void launch_dec_t(const DecArgs& a, dim3 grid, size_t lds, hipStream_t s) {
    auto set = [&](const void* f) {
        // per (function, device); thread_local: handles are confined to one host thread each, different threads may
        // run different handles at the same time, and setting the attribute twice is harmless
        static thread_local std::map<std::pair<const void*, int>, size_t> done;
        int dev = 0;
        (void)hipGetDevice(&dev);
        size_t& d = done[{f, dev}];
        if (d < lds) { (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); d = lds; }
    };
    const bool lcc = a.lcc != nullptr && PRE != PRE_COPY;
    if (a.ntaps == 3 && !lcc)      { set((const void*)dec_layer16<NV, PRE, 3, false>); hipLaunchKernelGGL((dec_layer16<NV, PRE, 3, false>), grid, dim3(64 * DEC_WAVES), lds, s, a); }
    else if (a.ntaps == 3)         { set((const void*)dec_layer16<NV, PRE, 3, true>);  hipLaunchKernelGGL((dec_layer16<NV, PRE, 3, true>), grid, dim3(64 * DEC_WAVES), lds, s, a); }
    else if (!lcc)                 { set((const void*)dec_layer16<NV, PRE, 1, false>); hipLaunchKernelGGL((dec_layer16<NV, PRE, 1, false>), grid, dim3(64 * DEC_WAVES), lds, s, a); }
    else                           { set((const void*)dec_layer16<NV, PRE, 1, true>);  hipLaunchKernelGGL((dec_layer16<NV, PRE, 1, true>), grid, dim3(64 * DEC_WAVES), lds, s, a); }
}

void launch_dec_layer(const DecArgs& a, int Npad16, hipStream_t s) {
    const int Ktot = a.ntaps * a.kc;
    const size_t lds = (size_t)(16 * (Ktot + 4) + 256 * DEC_WAVES) * 4;
    const int Bpad = round_up(a.B, 16);
    const dim3 grid(Npad16 / 16, Bpad / 16);
    const bool wide = a.cin > 256;          // NV = 2 (cin <= 512)
    if (a.pre == PRE_COPY) { if (wide) launch_dec_t<2, PRE_COPY>(a, grid, lds, s); else launch_dec_t<1, PRE_COPY>(a, grid, lds, s); }
    else if (a.pre == PRE_CONV) { if (wide) launch_dec_t<2, PRE_CONV>(a, grid, lds, s); else launch_dec_t<1, PRE_CONV>(a, grid, lds, s); }
    else { if (wide) launch_dec_t<2, PRE_HC>(a, grid, lds, s); else launch_dec_t<1, PRE_HC>(a, grid, lds, s); }
}

// =====================================================================================
// cone_fc16 (ConeFcArgs): see oph_internal.h.  grid = (Npad/16, (n_out + n_extra) * Bpad/16), 16 waves; wave w owns
// utterance row w of the group for the three gathered positions.
// =====================================================================================
// FC_CT: 16-column MFMA tiles per workgroup.  The prologue is redone per column slice, so slices are wide (64 columns) when
// there are many row groups and 32 columns when few (more CUs share the contraction).
template <int NS, int FC_CT>       // NS: split-K partials of the producing GEMM (1: a cone_fc16 or an unsplit GEMM, 2, 4)
__global__ __launch_bounds__(64 * DEC_WAVES) void cone_fc16(ConeFcArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Ktot = 3 * a.kc, ldxs = Ktot + 4;
    float* xs = smem;                 // [16][ldxs]
    float* part = smem + 16 * ldxs;   // [DEC_WAVES][FC_CT][256]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int UB = a.Bpad >> 4;
    const int grp = blockIdx.y / UB, ub = blockIdx.y - grp * UB;
    const bool is_extra = grp >= a.n_out;
    const int n0 = blockIdx.x * 16 * FC_CT;
    const int r16 = lane & 15, kq = lane >> 4;
    const int nchunks = Ktot >> 4;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const bool live = !stopped(a.stop_after, a.t);
    const bool store_blk = blockIdx.x == 0;
    if (live && !(is_extra && !store_blk)) {
        // ---- independent requests: weight fragments and the raw / residual rows of this wave's three positions.  With
        //      split-K partials to sum the fragments are requested after the sums (128 registers per lane at 16 waves)
        f32x4 bfrag[FC_CT][DEC_PF];
        auto fetch_w = [&]() {
#pragma unroll
            for (int ct = 0; ct < FC_CT; ++ct) {
                const float* wrow = a.Wt + (size_t)(n0 + ct * 16 + r16) * a.ldw + kq * 4;
                const bool cols = !is_extra && n0 + ct * 16 < a.N;
#pragma unroll
                for (int i = 0; i < DEC_PF; ++i) {
                    const int ch = w + DEC_WAVES * i;
                    bfrag[ct][i] = (cols && ch < nchunks) ? *(const f32x4*)(wrow + ch * 16) : zero4;
                }
            }
        };
        if (NS == 1) fetch_w();
        const int ocol = n0 + (tid >> 8) * 16 + (tid & 15);          // reducer role: thread -> (tile tid >> 8, row (tid >> 4) & 15, column tid & 15)
        const bool reducer = (tid >> 8) < FC_CT;
        const float bias_v = (!is_extra && reducer && ocol < a.N) ? a.bias[ocol] : 0.f;
        const int C = a.C, c = lane * 4;
        const bool cok = c < C;
        const int urow = ub * 16 + w;
        int pos[3], rpos[3]; bool ok[3];
#pragma unroll
        for (int tp = 0; tp < 3; ++tp) {
            if (is_extra) { pos[tp] = a.extra[grp - a.n_out]; rpos[tp] = a.extra_res[grp - a.n_out]; ok[tp] = tp == 2; }
            else { pos[tp] = a.tab[tp][grp]; rpos[tp] = a.res[tp][grp]; ok[tp] = a.j >= a.need[tp][grp]; }
        }
        // every request of the three rows (and of all split-K partials) before the first use
        f32x4 h1[3], h2[3], xr[3], p1[3][NS], p2[3][NS];        // ([NS - 1] used)
#pragma unroll
        for (int tp = 0; tp < 3; ++tp) {
            h1[tp] = zero4; h2[tp] = zero4; xr[tp] = zero4;
            const bool in = ok[tp] && cok;
            const float* hp = a.rawp + ((size_t)pos[tp] * a.Bpad + urow) * a.ldrawp;
            if (in) { h1[tp] = *(const f32x4*)(hp + c); h2[tp] = *(const f32x4*)(hp + C + c); }
#pragma unroll
            for (int sp = 1; sp < NS; ++sp) {
                const bool ins = in && sp < a.nsplit;
                p1[tp][sp - 1] = ins ? *(const f32x4*)(hp + sp * a.split_stride + c) : zero4;
                p2[tp][sp - 1] = ins ? *(const f32x4*)(hp + sp * a.split_stride + C + c) : zero4;
            }
            if (in) xr[tp] = *(const f32x4*)(a.xres + ((size_t)rpos[tp] * a.Bpad + urow) * a.ldres + c);
        }
#pragma unroll
        for (int tp = 0; tp < 3; ++tp)
#pragma unroll
            for (int sp = 0; sp + 1 < NS; ++sp) { h1[tp] += p1[tp][sp]; h2[tp] += p2[tp][sp]; }      // same order as ln_rows: partial 0 + 1 + 2 + 3
        if (NS > 1) { asm volatile("" ::: "memory"); fetch_w(); }
        f32x4 g1v = zero4, b1v = zero4, g2v = zero4, b2v = zero4;
        if (cok) { g1v = *(const f32x4*)(a.g1 + c); b1v = *(const f32x4*)(a.b1 + c); g2v = *(const f32x4*)(a.g2 + c); b2v = *(const f32x4*)(a.b2 + c); }
        // ---- prologue: x = sigmoid(LN1(h1)) * LN2(h2) + (1 - sigmoid) * residual  (modules.py:194-203); the three rows'
        //      reductions are independent and interleave
        const float invc = 1.0f / (float)C;
        float s1[3], s2[3], q1[3], q2[3];
#pragma unroll
        for (int tp = 0; tp < 3; ++tp) {
            s1[tp] = wave_sum(h1[tp][0] + h1[tp][1] + h1[tp][2] + h1[tp][3]);
            s2[tp] = wave_sum(h2[tp][0] + h2[tp][1] + h2[tp][2] + h2[tp][3]);
        }
#pragma unroll
        for (int tp = 0; tp < 3; ++tp) {
            const float m1 = a.nonorm ? 0.f : s1[tp] * invc, m2 = a.nonorm ? 0.f : s2[tp] * invc;
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d1 = cok ? h1[tp][e] - m1 : 0.f, d2 = cok ? h2[tp][e] - m2 : 0.f;
                h1[tp][e] = d1; h2[tp][e] = d2; a1 += d1 * d1; a2 += d2 * d2;
            }
            q1[tp] = wave_sum(a1); q2[tp] = wave_sum(a2);
        }
#pragma unroll
        for (int tp = 0; tp < 3; ++tp) {
            const float r1 = a.nonorm ? 1.0f : fast_rsqrt(q1[tp] * invc + LN_EPS), r2 = a.nonorm ? 1.0f : fast_rsqrt(q2[tp] * invc + LN_EPS);
            f32x4 x = zero4;
            if (ok[tp] && cok) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gte = fast_sigmoid(h1[tp][e] * r1 * g1v[e] + b1v[e]);
                    x[e] = gte * (h2[tp][e] * r2 * g2v[e] + b2v[e]) + (1.0f - gte) * xr[tp][e];
                }
            }
            if (c < a.kc) *(f32x4*)(xs + w * ldxs + tp * a.kc + c) = x;
            for (int c2 = 256 + c; c2 < a.kc; c2 += 256) *(f32x4*)(xs + w * ldxs + tp * a.kc + c2) = zero4;
            // level k of the cone: column slice 0 keeps the rows of the CURRENT position (next level's residual, the
            // loop kernel's taps where the position is one of them)
            if (store_blk && tp == 2 && ok[tp] && c < a.ldx) {
                float* y = a.xstore + ((size_t)pos[tp] * a.Bpad + urow) * a.ldx + c;
                if (a.done_sig && (pos[tp] == a.coh0 || pos[tp] == a.coh1)) st_coherent(y, x);
                else *(f32x4*)y = x;
            }
        }
        if (!is_extra) {
            __syncthreads();
            // ---- 16 x 64 slice (FC_CT MFMA tiles), K split over the waves (dec_layer16)
            f32x4 acc[FC_CT][2];
#pragma unroll
            for (int ct = 0; ct < FC_CT; ++ct) { acc[ct][0] = zero4; acc[ct][1] = zero4; }
            const float* xa = xs + r16 * ldxs + kq * 4;
#pragma unroll
            for (int i = 0; i < DEC_PF; ++i) {
                const int ch = w + DEC_WAVES * i;
                if (ch < nchunks) {
                    const f32x4 av = *(const f32x4*)(xa + ch * 16);
#pragma unroll
                    for (int ct = 0; ct < FC_CT; ++ct) {
                        acc[ct][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bfrag[ct][i][0], acc[ct][0], 0, 0, 0);
                        acc[ct][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bfrag[ct][i][1], acc[ct][1], 0, 0, 0);
                        acc[ct][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bfrag[ct][i][2], acc[ct][0], 0, 0, 0);
                        acc[ct][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bfrag[ct][i][3], acc[ct][1], 0, 0, 0);
                    }
                }
            }
            // C/D layout of the 16x16 MFMA: column = lane & 15, row = 4 (lane >> 4) + register
#pragma unroll
            for (int ct = 0; ct < FC_CT; ++ct)
#pragma unroll
                for (int e = 0; e < 4; ++e) part[(w * FC_CT + ct) * 256 + (kq * 4 + e) * 16 + r16] = acc[ct][0][e] + acc[ct][1][e];
            __syncthreads();
            if (reducer) {
                const int ct = tid >> 8, rc = tid & 255, row = rc >> 4;
                float pv[DEC_WAVES];
#pragma unroll
                for (int ww = 0; ww < DEC_WAVES; ++ww) pv[ww] = part[(ww * FC_CT + ct) * 256 + rc];
                float v = bias_v;
#pragma unroll
                for (int ww = 0; ww < DEC_WAVES; ++ww) v += pv[ww];
                if (ocol < a.N) a.H[((size_t)grp * a.Bpad + ub * 16 + row) * a.ldh + ocol] = v;
            }
        }
    }
    if (a.done_sig && store_blk) {      // every storing workgroup arrives (block-uniform)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned old = atomicAdd(a.done_count, 1u);
            if (old + 1u == a.done_target) {
                __hip_atomic_fetch_max(a.done_sig, a.done_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (a.done_stamp) *a.done_stamp = wall_clock64();      // diagnostics (OPH_RUN_STAMPS): when this level was complete
            }
        }
    }
}
template <int NS, int CT>
static void launch_cone_fc16_t(const ConeFcArgs& a, hipStream_t s) {
    const size_t lds = (size_t)(16 * (3 * a.kc + 4) + 256 * DEC_WAVES * CT) * 4;
    static thread_local std::map<int, size_t> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    size_t& d = done[dev];
    if (d < lds) { (void)hipFuncSetAttribute((const void*)cone_fc16<NS, CT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); d = lds; }
    const dim3 grid(round_up(a.N, 16 * CT) / (16 * CT), (a.n_out + a.n_extra) * (a.Bpad / 16));
    hipLaunchKernelGGL((cone_fc16<NS, CT>), grid, dim3(64 * DEC_WAVES), lds, s, a);
}
void launch_cone_fc16(const ConeFcArgs& a, hipStream_t s) {
    // 32-column slices while that still leaves at most ~one workgroup per CU of the cone's partition
    const bool narrow = (a.n_out + a.n_extra) * (a.Bpad / 16) * (round_up(a.N, 32) / 32) <= 128;
    if (a.nsplit <= 1) { if (narrow) launch_cone_fc16_t<1, 2>(a, s); else launch_cone_fc16_t<1, 4>(a, s); }
    else if (a.nsplit == 2) { if (narrow) launch_cone_fc16_t<2, 2>(a, s); else launch_cone_fc16_t<2, 4>(a, s); }
    else { if (narrow) launch_cone_fc16_t<4, 2>(a, s); else launch_cone_fc16_t<4, 4>(a, s); }
}

// attn_rows: generic rows.  mode 0 = decoder history rows (position-major, current mask p);
// mode 1 = batched operator over (b,t) with alignments + argmax outputs.
__global__ __launch_bounds__(256) void attn_rows(AttnRowsArgs a) {
    if (a.wait_sig) {
        // first launch of a cone: its inputs (prev_max, Q[t-1]) exist once the decoder loop kernel has raised this word
        if (threadIdx.x == 0) {
            long long t0 = 0;
            for (int it = 0; (int)(__hip_atomic_load(a.wait_sig, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - a.wait_val) < 0; ++it) {
                __builtin_amdgcn_s_sleep(8);
                if ((it & 255) == 255) {
                    const long long now = wall_clock64();
                    if (t0 == 0) t0 = now;
                    if (now - t0 > 200000000LL || __hip_atomic_load(a.wait_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                        __hip_atomic_store(a.wait_err, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // the loop kernel wrote Q and prev_max through; drop stale L1 lines
        }
        __syncthreads();
    }
    if (stopped(a.stop_after, a.t)) return;
    const int lane = threadIdx.x & 63;
    const int rid = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (rid >= a.nrows) return;
    int b, tq = 0;
    const float* qp;
    if (a.mode == 0) {
        const int i = rid / a.Bpad;
        b = rid - i * a.Bpad;
        const int t = a.j - a.off[i];
        if (b >= a.B || t < 0) return;
        qp = a.Q + ((size_t)t * a.Bpad + b) * a.ldq;
    } else {
        b = rid / a.T;
        tq = rid - b * a.T;
        qp = a.Q + (size_t)rid * a.ldq;
    }
    const int d = a.d;
    f32x4 q[ATT_NV], ctx[ATT_NV];
#pragma unroll
    for (int v = 0; v < ATT_NV; ++v) {
        const int c = (v * 64 + lane) * 4;
        q[v] = c < d ? *(const f32x4*)(qp + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int p = a.p[b];
    const float* Kb = a.K + (size_t)b * a.N * a.ldkv;
    const float* Vb = a.V + (size_t)b * a.N * a.ldkv;
    float* rr = a.R + (size_t)rid * a.ldr;
    if (a.ptab) {                     // FixedAttention (external durations): R = [V[key of this query's time], Q]
        const int tt = a.mode == 0 ? a.j - a.off[rid / a.Bpad] : tq;
        const int pf = a.ptab[(size_t)tt * a.Bpad + b];
#pragma unroll
        for (int v = 0; v < ATT_NV; ++v) {
            const int c = (v * 64 + lane) * 4;
            if (c < d) {
                *(f32x4*)(rr + c) = pf >= 0 ? *(const f32x4*)(Vb + (size_t)pf * a.ldkv + c) : f32x4{0.f, 0.f, 0.f, 0.f};
                *(f32x4*)(rr + d + c) = q[v];
            }
        }
        return;
    }
    if (a.ends) {                     // non-monotonic synthesis: every key of the text (+1) takes part
        const int nkeys = min(a.N, a.ends[b] + 1);
        const AttnFull of = attend_full(q, Kb, Vb, a.ldkv, nkeys, d, lane, ctx);
#pragma unroll
        for (int v = 0; v < ATT_NV; ++v) {
            const int c = (v * 64 + lane) * 4;
            if (c < d) {
                *(f32x4*)(rr + c) = ctx[v];
                *(f32x4*)(rr + d + c) = q[v];
            }
        }
        if (a.mode == 1) {
            float* al = a.align + (size_t)b * a.N * a.T + tq;
#pragma unroll
            for (int s_ = 0; s_ < ATT_FULL_SLOTS; ++s_) {
                const int n = s_ * 64 + lane;
                if (n < a.N) al[(size_t)n * a.T] = n < nkeys ? of.prob[s_] : 0.f;
            }
            if (lane == 0) a.amax[rid] = of.arg;
        }
        return;
    }
    const AttnOut o = attend_window(q, Kb, Vb, a.ldkv, p, a.N, a.win, d, lane, ctx);
#pragma unroll
    for (int v = 0; v < ATT_NV; ++v) {
        const int c = (v * 64 + lane) * 4;
        if (c < d) {
            *(f32x4*)(rr + c) = ctx[v];
            *(f32x4*)(rr + d + c) = q[v];
        }
    }
    if (a.mode == 1) {
        float* al = a.align + (size_t)b * a.N * a.T + tq;
        for (int n = lane; n < a.N; n += 64) {
            float pv = 0.f;
#pragma unroll
            for (int i = 0; i < ATT_WMAX; ++i)
                if (i < o.nwin && n == p + i) pv = o.prob[i];
            al[(size_t)n * a.T] = pv;
        }
        if (lane == 0) a.amax[rid] = p + o.arg;
    }
}

void launch_attn_rows(const AttnRowsArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(attn_rows, dim3((a.nrows + 3) / 4), dim3(256), 0, s, a);
}

// =====================================================================================
// row_chain: grid = B workgroups (one utterance each), 1024 threads = 16 waves.
// Stage = one k=1 conv + LayerNorm + activation on ONE row: wave w takes K/16 of the reduction,
// lane l the 4 output columns 4l..4l+3 (coalesced 1 KB weight rows, all 16 loads of a pass in
// flight), partials meet in LDS, wave 0 normalises and writes the next stage's input.
// Prologues: copy a row / highway gate of the previous layer / gate + attention + bookkeeping.
// =====================================================================================
constexpr int RC_WAVES = 16;
constexpr int RC_XMAX = 1024;

template <int ATT, bool NONORM, bool LCC>   // ATT: 0 monotonic window, 1 every key of the text, 2 fixed (external durations).
                                            // Option variants are separate instantiations: the default kernel's register
                                      // allocation (at the 128-VGPR cap, no scratch) must stay untouched
__global__ __launch_bounds__(64 * RC_WAVES) void row_chain(RowChainArgs a) {
    __shared__ __attribute__((aligned(16))) float xs[2][RC_XMAX];
    __shared__ __attribute__((aligned(16))) float part[RC_WAVES][256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, b = blockIdx.x;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int stop_v = a.stop_after ? *a.stop_after : 0x7fffffff;
    const bool live = a.t <= stop_v;
    int cur = 0;
    // Stage weights do not depend on activations: all 16 rows of a wave's K-slice are requested one stage ahead
    // (stage 0: before the prologue; stage l+1: right after stage l's FMAs), so their L2/MALL round trip hides
    // behind the prologue or behind the previous stage's reduce + LayerNorm barriers instead of being paid twice
    // per stage (micro-benchmark: a cold 256 KB block costs ~1.3 us per exposed round trip).
    f32x4 wA[8], wB[8];
    const int col = lane * 4;
    const int wu = __builtin_amdgcn_readfirstlane(w);      // wave index as a scalar: row pointers live in SGPRs
#define RC_FETCH_STAGE(LREF, BASE)   /* rows BASE..BASE+15 of this wave's K slice -> wA (8), wB (8) */   \
    {                                                                                                 \
        const int kr_ = (LREF).kc / RC_WAVES, ldn_ = (LREF).ldn;                                      \
        const float* rowp_ = (LREF).W + (size_t)(wu * kr_ + (BASE)) * ldn_ + col;                     \
        const bool colok_ = col < (LREF).N;                                                           \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                            \
            wA[i_] = (colok_ && (BASE) + i_ < kr_) ? *(const f32x4*)(rowp_ + (size_t)i_ * ldn_) : zero4;        \
            wB[i_] = (colok_ && (BASE) + 8 + i_ < kr_) ? *(const f32x4*)(rowp_ + (size_t)(8 + i_) * ldn_) : zero4; \
        }                                                                                             \
    }
    // ---------------- prologue -> xs[0][0 .. kc0)
    const int kc0 = a.L[0].kc;
    if (a.pro == ROW_COPY) {
        const float* sp = a.src + (size_t)b * a.ldsrc;
        for (int c = tid; c < kc0; c += 64 * RC_WAVES) xs[0][c] = c < a.cin ? sp[c] : 0.f;
    } else if (w == 0) {
        const int d = a.cin;
        f32x4 q[ATT_NV], u[ATT_NV];
        const float* h = a.src + (size_t)b * a.ldsrc;
        const float* xr = a.xres + (size_t)b * a.ldres;
        f32x4 xv[ATT_NV], g1v[ATT_NV], b1v[ATT_NV], g2v[ATT_NV], b2v[ATT_NV];
        const int p = a.pro == ROW_ATTN ? a.pcur[b] : 0;
#pragma unroll
        for (int v = 0; v < ATT_NV; ++v) {
            const int c = (v * 64 + lane) * 4;
            const bool ok = c < d;
            q[v] = ok ? *(const f32x4*)(h + c) : zero4;
            u[v] = ok ? *(const f32x4*)(h + d + c) : zero4;
            xv[v] = ok ? *(const f32x4*)(xr + c) : zero4;
            g1v[v] = ok ? *(const f32x4*)(a.g1 + c) : zero4; b1v[v] = ok ? *(const f32x4*)(a.b1 + c) : zero4;
            g2v[v] = ok ? *(const f32x4*)(a.g2 + c) : zero4; b2v[v] = ok ? *(const f32x4*)(a.b2 + c) : zero4;
        }
        auto ln = [&](f32x4 (&z)[ATT_NV], const f32x4 (&gv)[ATT_NV], const f32x4 (&bv)[ATT_NV]) {
            float s = 0.f;
#pragma unroll
            for (int v = 0; v < ATT_NV; ++v) s += z[v][0] + z[v][1] + z[v][2] + z[v][3];
            const float mean = NONORM ? 0.f : wave_sum(s) / (float)d;
            float qq = 0.f;
#pragma unroll
            for (int v = 0; v < ATT_NV; ++v)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dlt = ((v * 64 + lane) * 4 < d) ? z[v][e] - mean : 0.f;
                    z[v][e] = dlt;
                    qq += dlt * dlt;
                }
            const float rstd = NONORM ? 1.0f : 1.0f / sqrtf(wave_sum(qq) / (float)d + LN_EPS);
#pragma unroll
            for (int v = 0; v < ATT_NV; ++v)
#pragma unroll
                for (int e = 0; e < 4; ++e) z[v][e] = z[v][e] * rstd * gv[v][e] + bv[v][e];
        };
        ln(q, g1v, b1v);
        ln(u, g2v, b2v);
        if (LCC && a.lcc_pro) {
            const float* lgp = a.lcc_pro + (size_t)a.cat_ids[b] * d;
#pragma unroll
            for (int v = 0; v < ATT_NV; ++v) {
                const int c = (v * 64 + lane) * 4;
                if (c < d) {
                    const f32x4 lgv = *(const f32x4*)(lgp + c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) u[v][e] *= lgv[e];
                }
            }
        }
#pragma unroll
        for (int v = 0; v < ATT_NV; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gte = sigmoidf_(q[v][e]);
                q[v][e] = gte * u[v][e] + (1.0f - gte) * xv[v][e];
            }
        if (a.pro == ROW_HC) {
#pragma unroll
            for (int v = 0; v < ATT_NV; ++v) {
                const int c = (v * 64 + lane) * 4;
                if (c < kc0) *(f32x4*)(&xs[0][c]) = c < d ? q[v] : zero4;
            }
        } else {
            // R' = concat(softmax(QK^T/sqrt(d)) V, Q) for row t under the current mask
            const float* KVb = a.KV + (size_t)b * a.N_keys * 2 * d;
            f32x4 ctx[ATT_NV];
            AttnOut o;
            AttnFull of;
            constexpr bool NOMONO = ATT == 1;
            const int nkeys = NOMONO ? min(a.N_keys, a.ends[b] + 1) : 0;
            const int pf = ATT == 2 ? a.ptab[(size_t)a.t * a.Bpad + b] : 0;
            if (ATT == 2) {
#pragma unroll
                for (int v = 0; v < ATT_NV; ++v) {
                    const int c = (v * 64 + lane) * 4;
                    ctx[v] = (pf >= 0 && c < d) ? *(const f32x4*)(KVb + d + (size_t)pf * 2 * d + c) : zero4;
                }
            } else if (NOMONO) of = attend_full(q, KVb, KVb + d, 2 * d, nkeys, d, lane, ctx);
            else o = attend_window(q, KVb, KVb + d, 2 * d, p, a.N_keys, a.win, d, lane, ctx);
            float* qh = a.Qhist + ((size_t)a.t * a.Bpad + b) * d;
#pragma unroll
            for (int v = 0; v < ATT_NV; ++v) {
                const int c = (v * 64 + lane) * 4;
                if (c < d) {
                    *(f32x4*)(&xs[0][c]) = ctx[v];
                    *(f32x4*)(&xs[0][d + c]) = q[v];
                    if (live) *(f32x4*)(qh + c) = q[v];
                }
            }
            if (NOMONO && live) {
                float* al = a.align + (size_t)b * a.N_keys * a.max_T + a.t;
#pragma unroll
                for (int s_ = 0; s_ < ATT_FULL_SLOTS; ++s_) {
                    const int n = s_ * 64 + lane;
                    if (n < nkeys) al[(size_t)n * a.max_T] = of.prob[s_];
                }
            }
            if (ATT == 2) {
                // the selection matrix itself is the alignment; utterance ends come from the durations (host side)
                if (lane == 0 && live) {
                    if (pf >= 0) a.align[(size_t)b * a.N_keys * a.max_T + (size_t)pf * a.max_T + a.t] = 1.0f;
                    a.pnext[b] = pf >= 0 ? pf : 0;
                }
            } else if (lane == 0 && live) {
                float* al = a.align + (size_t)b * a.N_keys * a.max_T + a.t;
                if (!NOMONO) {
#pragma unroll
                    for (int i = 0; i < ATT_WMAX; ++i)
                        if (i < o.nwin) al[(size_t)(p + i) * a.max_T] = o.prob[i];
                }
                const int m = NOMONO ? of.arg : p + o.arg;
                a.pnext[b] = m;
                if (a.t_ends[b] == a.max_T && m >= a.ends[b]) {
                    a.t_ends[b] = a.t;
                    const int old = atomicAdd(a.n_ended, 1);
                    if (old + 1 == a.B && a.stop_mode == 0) *a.stop_flag = a.t;
                }
            }
        }
    }
    RC_FETCH_STAGE(a.L[0], 0);      // (holding 16 rows across the prologue would spill at the 128-VGPR cap of 1024 threads)
    __syncthreads();
    // ---------------- stages
    for (int li = 0; li < a.nlayers; ++li) {
        const RowLayer& L = a.L[li];
        const int kr = L.kc / RC_WAVES, k0 = w * kr;
        // bias / gamma / beta are needed only after the barriers below: request them now so that their
        // global round trips are not exposed between the barriers
        f32x4 gv = zero4, bv = zero4, biasv = zero4;
        if (w == 0 && col < L.N) { gv = *(const f32x4*)(L.g + col); bv = *(const f32x4*)(L.b + col); biasv = *(const f32x4*)(L.bias + col); }
        f32x4 acc = zero4;
        for (int base = 0; base < kr; base += 16) {
            if (base > 0) RC_FETCH_STAGE(L, base);            // K slices longer than 16 rows (kc > 256): exposed fetch
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float xa = base + i < kr ? xs[cur][k0 + base + i] : 0.f;
                const float xb = base + 8 + i < kr ? xs[cur][k0 + base + 8 + i] : 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += xa * wA[i][e] + xb * wB[i][e];
            }
        }
        if (col < L.N) *(f32x4*)(&part[w][col]) = acc;
        if (li + 1 < a.nlayers) RC_FETCH_STAGE(a.L[li + 1], 0);   // prefetch: lands during the barriers below
        __syncthreads();
        const int nxt = cur ^ 1;
        const bool last = li + 1 == a.nlayers;
        const int kc_next = last ? (a.xout ? a.ldout : L.N) : a.L[li + 1].kc;
        if (w == 0) {
            // wave 0 reduces the 16 K-slice partials of its own 4 columns and goes straight into the LayerNorm
            f32x4 v = zero4;
            if (col < L.N) {
                v = biasv;
#pragma unroll
                for (int ww = 0; ww < RC_WAVES; ++ww) v += *(const f32x4*)(&part[ww][col]);
#pragma unroll
                for (int e = 0; e < 4; ++e) if (col + e >= L.N) v[e] = 0.f;
            }
            float s = v[0] + v[1] + v[2] + v[3];
            const float mean = NONORM ? 0.f : wave_sum(s) / (float)L.N;
            float qq = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float dlt = col + e < L.N ? v[e] - mean : 0.f; v[e] = dlt; qq += dlt * dlt; }
            const float rstd = NONORM ? 1.0f : 1.0f / sqrtf(wave_sum(qq) / (float)L.N + LN_EPS);
            if (LCC && L.lcc) {
                // y = gate * act(LN(h)); the squash sigmoid of the last decoder layer (outside conv1d) comes after the gate
                f32x4 lgv = zero4;
                if (col < L.N) lgv = *(const f32x4*)(L.lcc + (size_t)a.cat_ids[b] * L.N + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float hn = v[e] * rstd * gv[e] + bv[e];
                    v[e] = col + e < L.N ? (L.act == ACT_SIGMOID ? sigmoidf_(lgv[e] * hn) : lgv[e] * apply_act(hn, L.act)) : 0.f;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[e] = col + e < L.N ? apply_act(v[e] * rstd * gv[e] + bv[e], L.act) : 0.f;
            }
            if (col < RC_XMAX && col < ((kc_next + 3) & ~3)) *(f32x4*)(&xs[nxt][col]) = v;
            for (int c = 256 + lane; c < kc_next; c += 64) xs[nxt][c] = 0.f;
            if (!last && a.L[li + 1].ccat > 0) {     // speaker embedding appended to the next layer's input
                const int id = a.cat_ids[b];
                for (int c = lane; c < a.L[li + 1].ccat; c += 64)
                    xs[nxt][L.N + c] = id == 0 ? 0.f : a.cat_table[(size_t)id * a.L[li + 1].ccat + c];
            }
        }
        __syncthreads();
        cur = nxt;
    }
    // ---------------- epilogue
    if (!live) return;
    if (a.xout) {
        float* xo = a.xout + (size_t)b * a.ldout;
        for (int c = tid; c < a.ldout; c += 64 * RC_WAVES) xo[c] = xs[cur][c];
    }
    if (a.emit) {
        const int nm = a.L[a.nlayers - 1].N;
        float* yo = a.Yout + ((size_t)b * a.max_T + a.t) * a.ldy;
        float* yt = a.Ytm + ((size_t)(a.t + 1) * a.Bpad + b) * a.ldtm;
        for (int c = tid; c < a.ldy; c += 64 * RC_WAVES) {
            const float v = c < nm ? xs[cur][c] : 0.f;
            yo[c] = v;
            if (c < a.ldtm) yt[c] = v;
        }
    }
}

void launch_row_chain(const RowChainArgs& a, hipStream_t s) {
    const int att = a.pro != ROW_ATTN ? 0 : (a.ptab ? 2 : (a.nomono ? 1 : 0));
    const dim3 grid(a.B), block(64 * RC_WAVES);
#define RC_LAUNCH(AT, NN, LC) hipLaunchKernelGGL((row_chain<AT, NN, LC>), grid, block, 0, s, a)
    const int sel = att * 4 + (a.nonorm ? 2 : 0) + (a.has_lcc ? 1 : 0);
    switch (sel) {
        case 0: RC_LAUNCH(0, false, false); break;
        case 1: RC_LAUNCH(0, false, true); break;
        case 2: RC_LAUNCH(0, true, false); break;
        case 3: RC_LAUNCH(0, true, true); break;
        case 4: RC_LAUNCH(1, false, false); break;
        case 5: RC_LAUNCH(1, false, true); break;
        case 6: RC_LAUNCH(1, true, false); break;
        case 7: RC_LAUNCH(1, true, true); break;
        case 8: RC_LAUNCH(2, false, false); break;
        case 9: RC_LAUNCH(2, false, true); break;
        case 10: RC_LAUNCH(2, true, false); break;
        default: RC_LAUNCH(2, true, true); break;
    }
#undef RC_LAUNCH
}

// embed_rows: modules.py:15-44 (row 0 replaced by zeros at lookup time); pads to ldo with zeros
__global__ __launch_bounds__(256) void embed_rows(const int* ids, long long n, const float* table, int units, float* out, int ldo) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int per = ldo / 4;
    if (i >= n * per) return;
    const long long row = i / per;
    const int c = (int)(i - row * per) * 4;
    const int id = ids[row];
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (id != 0 && c + e < units) v[e] = table[(size_t)id * units + c + e];
    *(f32x4*)(out + row * ldo + c) = v;
}
void launch_embed(const int* ids, long long n, const float* table, int units, float* out, int ldo, hipStream_t s) {
    const long long tot = n * (ldo / 4);
    hipLaunchKernelGGL(embed_rows, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, ids, n, table, units, out, ldo);
}

// spk_append_rows: out[row][col0 : col0+dim) = table[ids[row / T]] (row 0 of the table reads as zeros, modules.py:38-40)
// -- tf.tile(speaker_codes, [1, T]) -> embed -> concat on the channel axis (networks.py:139-144)
__global__ __launch_bounds__(256) void spk_append_rows_k(float* out, int ldo, long long rows, int T, int col0, const float* table, const int* ids, int dim) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * dim) return;
    const long long row = i / dim;
    const int c = (int)(i - row * dim);
    const int id = ids[row / T];
    out[row * ldo + col0 + c] = id == 0 ? 0.f : table[(size_t)id * dim + c];
}
void launch_spk_append_rows(float* out, int ldo, long long rows, int T, int col0, const float* table, const int* ids, int dim, hipStream_t s) {
    const long long tot = rows * dim;
    hipLaunchKernelGGL(spk_append_rows_k, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, out, ldo, rows, T, col0, table, ids, dim);
}

// pad_rows: dst[r][0:ldd) = src[r][0:C) then zeros
__global__ __launch_bounds__(256) void pad_rows_k(const float* src, int lds_, float* dst, int ldd, long long rows, int C) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * ldd) return;
    const long long r = i / ldd;
    const int c = (int)(i - r * ldd);
    dst[i] = c < C ? src[r * lds_ + c] : 0.f;
}
__global__ void copy_rows_strided_k(const float* src, long long src_bs, int ld, float* dst, int B, int T, int C4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * T * C4) return;
    const int c = (int)(i % C4); const long long r = i / C4; const int t = (int)(r % T), b = (int)(r / T);
    *(f32x4*)(dst + r * ld + 4 * c) = *(const f32x4*)(src + b * src_bs + (long long)t * ld + 4 * c);
}
// a [B][T] window of time-strided rows as one dense batch (ld = row stride of both, C a multiple of 4)
void launch_copy_rows_strided(const float* src, long long src_bs, int ld, float* dst, int B, int T, int C, hipStream_t s) {
    const long long n = (long long)B * T * (C / 4);
    hipLaunchKernelGGL(copy_rows_strided_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, src_bs, ld, dst, B, T, C / 4);
}
void launch_pad_rows(const float* src, int lds_, float* dst, int ldd, long long rows, int C, hipStream_t s) {
    const long long tot = rows * ldd;
    hipLaunchKernelGGL(pad_rows_k, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, src, lds_, dst, ldd, rows, C);
}

__global__ __launch_bounds__(256) void fill_int_k(int* p, int v, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}
// A tile's decode state back to "nothing decoded yet" in ONE launch (synthesize.py:157-166: Y, alignments = zeros, prev_max = 0,
// t_ends = max_T; plus this library's control words): until round 5 four memsets, a fill kernel and a 16-byte copy from the host's
// stack -- six dependent stream operations of ~7 us each in front of every decode, and a host synchronisation to keep the stack alive.
__global__ __launch_bounds__(256) void reset_tile_k(ResetTileArgs a) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = tid; i < a.n4[0]; i += nth) ((f32x4*)a.buf[0])[i] = z;
    for (size_t i = tid; i < a.n4[1]; i += nth) ((f32x4*)a.buf[1])[i] = z;
    for (size_t i = tid; i < a.n4[2]; i += nth) ((f32x4*)a.buf[2])[i] = z;
    if (tid < (size_t)a.n_p) a.p[tid] = 0;
    if (tid < (size_t)a.n_tends) a.tends[tid] = a.max_T;
    if (tid < 4) a.ctl[tid] = tid == 1 ? INT_MAX : 0;
}
void launch_reset_tile(const ResetTileArgs& a, hipStream_t s) { hipLaunchKernelGGL(reset_tile_k, dim3(1024), dim3(256), 0, s, a); }
void launch_fill_int(int* p, int v, int n, hipStream_t s) {
    hipLaunchKernelGGL(fill_int_k, dim3((n + 255) / 256), dim3(256), 0, s, p, v, n);
}

}  // namespace oph
