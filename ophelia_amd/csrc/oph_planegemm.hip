// plane_gemm: the batched nets' conv-as-GEMM on operands that ARRIVE as fp16 hi / lo planes (modules.conv1d 91-146, hc 148-207,
// conv1d_transpose 209-258 behind networks.SSRN 437-537 / TextEnc 121-212).
//
// conv_gemm_bf16x3 (oph_kernels.hip) reads fp32 activation rows and splits them into 16-bit terms by VALU work + ds_write inside
// its K loop, once per tap; its 64x64 instance (what D_4 / D_7 ran on) reads four LDS fragments for three MFMAs.  Here the
// producing LayerNorm launch (ln_rows) has already written its rows as planes, K-blocked [channel / 32][row][32] (one K block of
// one row = 64 contiguous bytes, one K block of a tile = one contiguous run; the weights' planes alike, launch_kblock_planes), so
//   * no conversion instruction in the K loop: a 16-byte piece per lane travels global -> register -> LDS as it is;
//   * the activation tile is loaded ONCE per 32-channel block with a halo of 16 rows each side and every tap of a k = 3 layer
//     (or x[t], x[t-1] of the transposed convolution) reads it at a row offset -- a row whose tap falls outside the utterance is
//     AND-ed to zero in registers (no branch in the loop);
//   * a workgroup is 128 rows x 128 columns, a wave 64 x 64 (four accumulator tiles): 8 fragment reads per 12 MFMAs;
//   * conv1d_transpose is ONE problem: a workgroup owns 128 input rows x 64 channels of BOTH phases (columns = 64 even-phase +
//     64 odd-phase channels).  x[t] multiplies [Kt0 | Kt1] as a 128-column step, x[t-1] multiplies Kt2 into the even half only:
//     every workgroup does the same 3 units of work (the paired launch ran 2-unit and 1-unit workgroups side by side);
//   * the K loop's instruction order is written out (one LDS read or one piece behind each MFMA, sched_barrier between them), the
//     barrier that releases the next block stands before the last slice so that no block starts with an exposed LDS round trip;
//   * one workgroup per CU (the ring and the staged output tile take 64 - 136 KB).
// Measured (round 4, profiles/r04_planes.sh, r04_ssrn_pmc.sh): 43 - 49 cycles per MFMA and wave in the loop (32 = back to back), no
// LDS bank conflicts, the same time with the pieces by global_load_lds (3 - 4 stage ring, first version) and with no operand
// stream at all -- the loop is bound by its own issue stream (MFMA + fragment read pairs of ONE wave per SIMD), a workgroup's fixed
// cost (first blocks, output tile, dispatch) is ~8 us beside 12 - 55 us of loop.
// Arithmetic: a.b = ah.bh + al.bh + ah.bl on v_mfma_f32_32x32x16_f16, fp32 accumulate -- the three products of
// conv_gemm_bf16x3<.., F16 = true>; the K order is (channel block, tap) instead of (tap, channel block), so the two kernels differ
// in the last bits of the fp32 sums.  The order does not depend on where a row sits in its tile or on the number of rows, so a
// streamed SSRN chunk reproduces the one-piece evaluation bit for bit as before.
#include "oph_internal.h"
#include "oph_device.h"

#include <type_traits>

namespace oph {

typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef h16 h16x4 __attribute__((ext_vector_type(4)));

// WV = waves per workgroup.  4: 2 x 2 waves of 64 x 64 (transposed conv: 64 rows x [32 even- | 32 odd-phase] channels, 64 channels
// per workgroup).  8: 2 x 4 waves -- two per SIMD, which fill each other's issue gaps (one wave per SIMD runs the loop at 43-49
// cycles per MFMA) -- of 64 x 32 on the same 128 x 128 tile (transposed conv: the same wave tile, 128
// channels of both phases per workgroup: half the workgroups, one round for D_7).
template <int NT, bool CONVT, int WV> struct PlaneGemmCfg {
    static constexpr int WC = WV / 2;                      // wave columns
    static constexpr int BNC = CONVT ? 32 * WC : 128;      // output channels per workgroup
    static constexpr int NCOLS = CONVT ? 2 * BNC : 128;    // columns of the workgroup's output tile (transposed conv: both phases)
    static constexpr int TNW = CONVT ? 2 : 128 / WC / 32;  // 32-column accumulator tiles per wave (transposed conv: even, odd)
    static constexpr int HALO = NT == 1 ? 0 : PLANE_GEMM_HALO;
    static constexpr int AROWS = 128 + 2 * HALO;           // activation rows per stage
    static constexpr int A_WCH = AROWS / 16;               // 1 KB wave pieces (16 rows x 64 B) per plane
    static constexpr int A_PIECES = 2 * A_WCH;             // hi and lo planes together: 20 (halo) / 16
    static constexpr int A_OPS = (A_PIECES + WV - 1) / WV; // per wave (a surplus slot repeats the last piece: the same bytes to the same place)
    static constexpr int A_BUF = 2 * AROWS * 32;           // halves: [hi plane | lo plane]
    static constexpr int b_rows(int tap) { return CONVT ? (tap == 0 ? 2 * BNC : BNC) : 128; }       // weight rows of a tap's step
    static constexpr int b_pl(int tap) { return b_rows(tap) * 32; }                                  // halves of one weight plane of a tap
    static constexpr int b_ops(int tap) { return 2 * b_rows(tap) / 16 / WV; }                        // pieces per wave
    static constexpr int b_at(int tap) { return tap == 0 ? A_BUF : b_at(tap - 1) + 2 * b_pl(tap - 1); }   // where a tap's planes start in a stage
    static constexpr int op_at(int tap) { return tap == 0 ? A_OPS : op_at(tap - 1) + b_ops(tap - 1); }     // first piece index of a tap
    static constexpr int STAGE = b_at(NT);                 // halves per stage: activations + every tap's weights of one 32-channel block
    static constexpr int STAGES = 2;                       // LDS stages: the block being multiplied and the one being written (the blocks
                                                           // further ahead are in registers / in flight): 64 - 136 KB
    static constexpr int NOPS = op_at(NT);                 // 1 KB pieces (16 bytes per lane) per wave and stage
    static constexpr int LDO = NCOLS + 4;                  // floats per row of the staged output tile
    static constexpr size_t LDS_BYTES = (size_t)STAGES * STAGE * 2 > (size_t)128 * LDO * 4 ? (size_t)STAGES * STAGE * 2 : (size_t)128 * LDO * 4;
};

// LNF (round 6, transposed convolution only): the layer's LayerNorm (modules.py:47-75, 252-256) inside the launch.  Until round 5 D_4 / D_7
// were two launches -- this kernel writing raw rows, ln_rows re-reading them (10 of D_4's 32 us).  Here the column
// tiles of a row tile (8 x 64 channels, or 4 x 128) sit on ONE XCD with consecutive ids (workgroup b -> XCD b % 8), exchange per-row
// partial statistics of their 64-channel groups as 16-byte {mean, M2, epoch} granules (write-through stores, sc1 loads, the data is
// its own flag: hc_fused's exchange), pool them exactly (Chan et al.) in a fixed order, normalise their own channels in the staged
// output tile and store fp32 rows + the next layer's planes.  The partials are always per 64-channel group, whatever the form
// (4 or 8 waves): a streamed SSRN chunk (few rows: 4 waves) reproduces the one-piece evaluation (8 waves) bit for bit.
constexpr long long PG_LN_TIMEOUT_TICKS = 200000000LL;     // 2 s of the 100 MHz clock
template <int NT, bool CONVT, int WV = 4, int DBG = 0, bool LNF = false>        // DBG (measurement only): 1 = no MFMAs, 2 = no operand stream in the loop, 4 = no stores, 8 = three K blocks
__global__ __launch_bounds__(64 * WV, 1) void plane_gemm(PlaneGemmArgs a) {      // (one workgroup per CU: two of a k = 1 layer's -- 68 KB each -- measured 115 us against 103)
    static_assert(!CONVT || NT == 2, "transposed convolution: taps x[t], x[t-1]");
    typedef PlaneGemmCfg<NT, CONVT, WV> Cfg;
    constexpr int HALO = Cfg::HALO, AROWS = Cfg::AROWS, A_WCH = Cfg::A_WCH, A_OPS = Cfg::A_OPS, STAGE = Cfg::STAGE, STAGES = Cfg::STAGES;
    constexpr int WC = Cfg::WC, BNC = Cfg::BNC, TNW = Cfg::TNW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    h16* Ss = (h16*)smem;                            // [STAGES][STAGE]

    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int MT = (a.M + 127) / 128, NTL = (a.N + BNC - 1) / BNC;
    const int ntiles = MT * NTL;
    int tm, tn;
    if constexpr (LNF) {
        // the NTL column tiles of a row tile exchange statistics: ids {8 NTL q + 8 k + x : k} = row tile 8 q + x -- one XCD, consecutive
        // in that XCD's dispatch order, so the partners of a waiting workgroup are resident or next in line (no dead-lock as long as an
        // XCD holds NTL workgroups of this launch at a time: plane_gemm_ln_fits)
        const int bid = blockIdx.x, xcd = bid & 7;
        tn = (bid >> 3) % NTL; tm = (bid >> 3) / NTL * 8 + xcd;
        if (tm >= MT) return;                        // (grid padded to whole groups of 8 row tiles)
    } else {
        int id;
        {   // XCD-aware bijective remap (workgroups b, b + 8, .. share an XCD / L2): a contiguous chunk of tiles per XCD
            const int bid = blockIdx.x, q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, loc = bid >> 3;
            id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        }
        constexpr int GM = 8;                            // 8 row tiles x all column tiles per group: both panels stay in the L2
        const int width = GM * NTL, g = id / width, first_m = g * GM;
        const int gsz = min(MT - first_m, GM);
        tm = first_m + (id % width) % gsz; tn = (id % width) / gsz;
    }
    const int m0 = tm * 128, n0 = tn * BNC;

    // ---- what each lane copies.  A piece = 64 lanes x 16 B = 16 LDS rows of 64 B; lane -> (row lane >> 2, 16-byte position
    // lane & 3); the position a lane FETCHES is swizzled with the LDS row's (row >> 2) & 3 (= (lane >> 4) & 3: pieces start at
    // multiples of 16 rows), so a fragment read of 16 consecutive rows touches 16 distinct 4-bank groups
    const int lrow = lane >> 2, sp = ((lane & 3) ^ ((lane >> 4) & 3)) << 3;
    const size_t w_kstride = (size_t)a.nalloc * 32;  // halves between two K blocks of a weight plane
    const h16* pp[Cfg::NOPS];                        // this lane's 16 bytes of every piece of block 0
    int pdst[Cfg::NOPS];                             // where the piece starts in a stage (halves)
#pragma unroll
    for (int j = 0; j < A_OPS; ++j) {
        const int wq = min(j * WV + w, Cfg::A_PIECES - 1), plane = wq / A_WCH, i = (wq % A_WCH) * 16 + lrow;
        const int gr = min(max(m0 - HALO + i, 0), a.M - 1);         // (halo / tail rows outside the problem: clamped, never used unmasked)
        pp[j] = (plane ? a.Al : a.Ah) + (size_t)gr * 32 + sp;
        pdst[j] = wq * 512;
    }
#pragma unroll
    for (int tap = 0; tap < NT; ++tap) {
        const int rows = Cfg::b_rows(tap), ppl = rows / 16;          // pieces per plane
#pragma unroll
        for (int j = 0; j < Cfg::b_ops(tap); ++j) {
            const int wq = j * WV + w, plane = wq / ppl, rr = (wq % ppl) * 16 + lrow;
            const h16* src;
            if (CONVT && tap == 0 && rr >= BNC) src = (plane ? a.Wl2 : a.Wh2) + (size_t)(n0 + rr - BNC) * 32;        // odd phase: Kt1
            else src = (plane ? a.Wl : a.Wh) + (size_t)(tap * (a.kc >> 5)) * w_kstride + (size_t)(n0 + rr) * 32;    // (transposed conv, tap 1: Kt2 behind Kt0)
            pp[Cfg::op_at(tap) + j] = src + sp;
            pdst[Cfg::op_at(tap) + j] = Cfg::b_at(tap) + wq * 512;
        }
    }
    const size_t a_kstride = (size_t)a.M * 32;       // halves between two K blocks of the activation planes
    // one stage = the activation rows and every tap's weights of one 32-channel block: NOPS 1 KB pieces per wave, 16 bytes per lane.
    // A piece travels global -> register -> LDS: plain 16-byte loads stream 2-3 x what the LDS-DMA path (global_load_lds) takes in
    // per CU (measured round 4, here and in hc_fused: ~40 GB/s per CU by DMA whatever the ring depth -- the first version of this
    // kernel was bound by exactly that, profiles/r04_ssrn_pmc.sh); the piece requested during block kb is written to LDS during
    // block kb + 1 and multiplied in block kb + 2.
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 stg[Cfg::NOPS];
    auto load_op = [&](auto op_c, int kb) {          // piece `op` of block kb -> its staging register
        constexpr int op = decltype(op_c)::value;
        stg[op] = *(const i32x4*)(pp[op] + (size_t)kb * (op < A_OPS ? a_kstride : w_kstride));
    };
    auto store_op = [&](auto op_c, int stage) {      // staging register -> its place in ring stage `stage` (lane l: byte 16 l of the piece)
        constexpr int op = decltype(op_c)::value;
        *(i32x4*)(Ss + stage * STAGE + pdst[op] + lane * 8) = stg[op];
    };
    auto for_ops = [&](auto&& f) {
        auto go = [&](auto self, auto op_c) {
            constexpr int op = decltype(op_c)::value;
            if constexpr (op < Cfg::NOPS) { f(op_c); self(self, std::integral_constant<int, op + 1>{}); }
        };
        go(go, std::integral_constant<int, 0>{});
    };

    // ---- fragments.  32x32x16 operand: lane l holds row (l & 31), k = 8 (l >> 5) .. + 7 of the 16-wide slice
    const int wr = w / WC, wc = w % WC, r32 = lane & 31, kh = lane >> 5;
    int aoff[NT][2];                                 // halves into a stage's hi plane, K slice 0 (slice 1: ^ 16)
    unsigned keep[NT][2];                            // all ones, or 0 where this lane's row has no source for the tap (utterance boundary):
                                                     // the fragment is AND-ed with it -- no branch in the K loop
#pragma unroll
    for (int tap = 0; tap < NT; ++tap) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int lr = HALO + wr * 64 + i * 32 + r32 + a.off[tap];
            aoff[tap][i] = lr * 32 + ((kh ^ ((lr >> 2) & 3)) << 3);
            const int m = m0 + wr * 64 + i * 32 + r32, tt = m % a.T + a.off[tap];
            keep[tap][i] = (tt < 0 || tt >= a.T) ? 0u : 0xffffffffu;
        }
    }
    int boff[TNW];
#pragma unroll
    for (int jn = 0; jn < TNW; ++jn) {
        const int br = (CONVT ? jn * BNC + wc * 32 : wc * (128 / WC) + jn * 32) + r32;
        boff[jn] = br * 32 + ((kh ^ ((br >> 2) & 3)) << 3);
    }

    float bias_v[TNW];                               // (requested before the K loop: no round trip between the loop and the stores)
#pragma unroll
    for (int jn = 0; jn < TNW; ++jn) bias_v[jn] = a.bias[CONVT ? n0 + wc * 32 + r32 : n0 + wc * (128 / WC) + jn * 32 + r32];
    // fused LayerNorm: this thread's 8 channels of the store pass (the same in every iteration) -- gamma / beta requested here, with the bias
    f32x4 ln_g[2], ln_b[2];
    if constexpr (LNF) {
        constexpr int C8 = Cfg::NCOLS / 8, PC8 = BNC / 8;
        const int ch = n0 + ((tid % C8) % PC8) * 8;
        ln_g[0] = *(const f32x4*)(a.ln_gamma + ch); ln_g[1] = *(const f32x4*)(a.ln_gamma + ch + 4);
        ln_b[0] = *(const f32x4*)(a.ln_beta + ch); ln_b[1] = *(const f32x4*)(a.ln_beta + ch + 4);
    }
    f32x16 acc[2][TNW];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < TNW; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;

    h16x8 fr[2][8];
    constexpr int NQ = 2 * NT;
    // The barrier that releases block kb + 1 stands BEFORE block kb's last slice, and that slice reads block kb + 1's first fragments
    // behind its MFMAs: no exposed LDS round trip at the top of a block.  So every piece of block kb + 1 is written to LDS during
    // the slices before the last one.
    struct Sl {       // per slice: MFMAs, LDS reads issued behind them (the next slice's fragments), pieces moved before it
        static constexpr int tn(int q) { return CONVT ? (((q % (2 * NT)) >> 1) == 1 ? 1 : 2) : TNW; }
        static constexpr int nm(int q) { return 6 * tn(q); }
        static constexpr int nr(int q) { return 4 + 2 * tn(q + 1); }
        // a piece (LDS write of block kb + 1's, then the load of block kb + 2's into the same register) rides behind every MFMA that
        // has no LDS read behind it; a k = 1 layer has a single slice before the barrier: all of its pairs carry one
        static constexpr int first(int q) { return NT == 1 ? 0 : (nr(q) < nm(q) ? nr(q) : nm(q)); }
        static constexpr int base(int q) { return q == 0 ? 0 : base(q - 1) + 12 - first(q - 1); }
    };
    static_assert(Sl::base(NQ - 1) >= Cfg::NOPS, "every operand piece of a stage has a slot before the barrier");
    auto read_frag = [&](auto q_c, auto f_c, const h16* st) {
        constexpr int q = decltype(q_c)::value % NQ, f = decltype(f_c)::value, tap = q >> 1, ks = q & 1;
        constexpr int TNt = Sl::tn(q);
        if constexpr (f < 4) {
            constexpr int i = f & 1, lo = f < 2;
            fr[q & 1][f] = *(const h16x8*)(st + (lo ? AROWS * 32 : 0) + (aoff[tap][i] ^ (ks * 16)));
        } else {
            constexpr int g = f - 4, lo = g >= TNt, jn = g % TNt;
            fr[q & 1][f] = *(const h16x8*)(st + Cfg::b_at(tap) + (lo ? Cfg::b_pl(tap) : 0) + (boff[jn] ^ (ks * 16)));
        }
    };
    auto mfma_k = [&](auto q_c, auto k_c) {
        constexpr int q = decltype(q_c)::value, k = decltype(k_c)::value;
        constexpr int TNt = Sl::tn(q);
        // product by product over the tiles (al.bh, ah.bl, ah.bh): consecutive MFMAs never share an accumulator
        constexpr int p = k / (2 * TNt), i = (k / TNt) & 1, jn = k % TNt;
        constexpr int fa = p == 0 ? i : 2 + i, fb = 4 + (p == 1 ? TNt : 0) + jn;
        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[q & 1][fa], fr[q & 1][fb], acc[i][jn], 0, 0, 0);
    };
    // A K block is 2 NT slices (tap, 16-wide K slice); a slice = its fragment reads (8, or 6 for the x[t-1] step of the transposed
    // convolution), the boundary masks and 12 (6) MFMAs.  Software-pipelined by hand: slice q + 1's fragments are read into the
    // other register set behind slice q's first MFMAs, one read per MFMA, the operand pieces of block kb + D are requested behind
    // the others, and the order is pinned with sched_barrier -- left alone the compiler reads two fragments, waits, issues two
    // MFMAs, waits, ... (13 exposed LDS round trips per block with one wave per SIMD).
    // Fragment f of a slice: 0..3 = A rows (al0, al1, ah0, ah1), 4.. = B columns (bh0, [bh1], bl0, [bl1]).
    // st: this block's stage; stn: where slice q + 1 lives (the same stage, or the next block's for the last slice)
    auto slice = [&](auto q_c, const h16* stn, int kbn, int stage_w) {
        constexpr int q = decltype(q_c)::value, tap = q >> 1;
        constexpr int nm = Sl::nm(q), nr = Sl::nr(q);
        constexpr bool masked = CONVT ? tap == 1 : (NT == 3 && tap != 1);                              // taps at an offset can leave the utterance
        if constexpr (masked) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int f = 0; f < 4; ++f) fr[q & 1][f] = __builtin_bit_cast(h16x8, __builtin_bit_cast(u32x4, fr[q & 1][f]) & keep[tap][f & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        auto pair = [&](auto k_c) {
            constexpr int k = decltype(k_c)::value;
            if constexpr (k < nm) mfma_k(q_c, k_c);
            // read order: what the next slice's first MFMAs need first (al0, bh.., al1, ah0, bl.., ah1)
            if constexpr (k < nr) {
                constexpr int TNn = (nr - 4) / 2;
                constexpr int order2[8] = {0, 4, 5, 1, 2, 6, 7, 3}, order1[6] = {0, 4, 1, 2, 5, 3};
                constexpr int f = TNn == 2 ? order2[k] : order1[k];
                read_frag(std::integral_constant<int, q + 1>{}, std::integral_constant<int, f>{}, stn);
            }
            if constexpr (q < NQ - 1 && k >= Sl::first(q) && !(DBG & 2)) {
                constexpr int op = Sl::base(q) + k - Sl::first(q);
                if constexpr (op < Cfg::NOPS) { store_op(std::integral_constant<int, op>{}, stage_w); load_op(std::integral_constant<int, op>{}, kbn); }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        pair(std::integral_constant<int, 0>{}); pair(std::integral_constant<int, 1>{}); pair(std::integral_constant<int, 2>{});
        pair(std::integral_constant<int, 3>{}); pair(std::integral_constant<int, 4>{}); pair(std::integral_constant<int, 5>{});
        pair(std::integral_constant<int, 6>{}); pair(std::integral_constant<int, 7>{}); pair(std::integral_constant<int, 8>{});
        pair(std::integral_constant<int, 9>{}); pair(std::integral_constant<int, 10>{}); pair(std::integral_constant<int, 11>{});
    };
    auto read_slice0 = [&](const h16* st) {
        constexpr int nf = 4 + 2 * Sl::tn(0);
        auto go = [&](auto self, auto f_c) {
            constexpr int f = decltype(f_c)::value;
            if constexpr (f < nf) { read_frag(std::integral_constant<int, 0>{}, f_c, st); self(self, std::integral_constant<int, f + 1>{}); }
        };
        go(go, std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- K loop over 32-channel blocks.  Block kb is multiplied out of LDS stage kb & 1 while block kb + 1's pieces (loaded during
    // block kb - 1) are written to the other stage and block kb + 2's are requested into the registers they leave.
    const int nkb = (DBG & 8) ? 3 : a.kc / 32;       // (DBG 8: three blocks only -- what is left is the fixed cost of a workgroup)
    for_ops([&](auto op_c) { load_op(op_c, 0); });
    for_ops([&](auto op_c) { store_op(op_c, 0); load_op(op_c, nkb > 1 ? 1 : 0); });
    // (s_barrier directly: __syncthreads() carries a workgroup-scope fence, which the compiler turns into vmcnt(0) -- the loads in
    //  flight would have to land before every barrier)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (!(DBG & 1)) read_slice0(Ss);
    for (int kb = 0; kb < nkb; ++kb) {
        // no branch in the loop body: past the end the last block is loaded again and written to a stage nobody reads
        const int kbn = kb + 2 < nkb ? kb + 2 : nkb - 1;
        const int stage = kb & 1, stage_w = stage ^ 1;
        const h16* st = Ss + stage * STAGE;
        if constexpr (DBG & 1) {
            for_ops([&](auto op_c) { store_op(op_c, stage_w); load_op(op_c, kbn); });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else {
            slice(std::integral_constant<int, 0>{}, st, kbn, stage_w);
            if constexpr (NQ > 2) { slice(std::integral_constant<int, 1>{}, st, kbn, stage_w); slice(std::integral_constant<int, 2>{}, st, kbn, stage_w); }
            if constexpr (NQ > 4) { slice(std::integral_constant<int, 3>{}, st, kbn, stage_w); slice(std::integral_constant<int, 4>{}, st, kbn, stage_w); }
            // block kb + 1 is in LDS once every wave's writes have completed (and this wave's fragment reads of stage kb & 1, which
            // the next iteration's writes overwrite)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            slice(std::integral_constant<int, NQ - 1>{}, Ss + stage_w * STAGE, kbn, stage_w);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- raw rows (bias added), through LDS: the accumulators' layout (C/D of the 32x32 MFMA: column = lane & 31, row = (reg & 3)
    // + 8 (reg >> 2) + 4 (lane >> 5)) would leave as 64 four-byte stores per lane in 128-byte runs (measured 4 us of D_4's 22);
    // as a [128][128] tile in the ring's LDS it leaves as 16 sixteen-byte stores per lane in 512-byte (transposed conv: 256-byte) runs
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // every wave is done with the ring
    constexpr int LDO = Cfg::LDO, NCOLS = Cfg::NCOLS;
    float* Os = smem;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < TNW; ++jn) {
            const int cl = CONVT ? jn * BNC + wc * 32 + r32 : wc * (128 / WC) + jn * 32 + r32;      // (transposed: [even-phase | odd-phase] channels)
            const float bv = bias_v[jn];
#pragma unroll
            for (int e = 0; e < 16; ++e) Os[(wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh) * LDO + cl] = acc[i][jn][e] + bv;
        }
    __syncthreads();
    if constexpr (LNF) {
        static_assert(CONVT, "the fused LayerNorm serves the transposed convolution");
        // ---- LayerNorm of the 256 output rows of this tile (input row r, phase p -> output row 2 (m0 + r) + p) over all a.N channels
        constexpr int NG = BNC / 64;                               // 64-channel groups per phase in this workgroup (1: 4 waves, 2: 8 waves)
        static_assert(64 * WV == 256 * NG, "one (row, phase, group) task per thread");
        __shared__ float srow[256][2];                             // mean, rstd per (row, phase)
        __shared__ float spart[256][NG][2];                        // this workgroup's own partials (its partners' come from memory)
        const int rp = tid / NG, gq = tid % NG, r = rp >> 1, p = rp & 1;
        const int ngt = a.N >> 6, g0 = n0 >> 6, gt = g0 + gq;      // groups of a row; this workgroup's first; this thread's
        float* const xs = Os + r * LDO + p * BNC + gq * 64;
        float mean_g, m2_g;
        {
            f32x4 v[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) v[c] = *(const f32x4*)(xs + 4 * c);
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c) s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
            mean_g = s * (1.0f / 64.0f);
            float q = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float dl = v[c][e] - mean_g; q = fmaf(dl, dl, q); }
            m2_g = q;
        }
        float* const region = a.ln_stats + (size_t)tm * ngt * 256 * 4;
        {
            f32x4 g;
            g[0] = mean_g; g[1] = m2_g; g[2] = __uint_as_float(a.ln_epoch); g[3] = 0.f;
            st_sc1_b128(region, (unsigned)((gt * 256 + rp) * 16), g);
        }
        spart[rp][gq][0] = mean_g; spart[rp][gq][1] = m2_g;
        __syncthreads();
        // gather the row's ngt partials: this workgroup's own from LDS (the floats it stored: the same bits), the partners' from memory
        // (sc1 loads past the L1, until every granule carries this launch's tag)
        float mean = 0.f, rstd = 0.f;
        {
            long long t0 = 0;
            for (int it = 0;; ++it) {
                float gm[16], gq2[16];                       // (N <= 1024: at most 16 groups; constant trip counts keep them in registers)
                bool ok = true;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const bool own = j >= g0 && j < g0 + NG;
                    if (j < ngt && !own) {
                        const f32x4 g = ld_sc1_b128(region, (unsigned)((j * 256 + rp) * 16));
                        gm[j] = g[0]; gq2[j] = g[1];
                        ok = ok && __float_as_uint(g[2]) == a.ln_epoch;
                    } else if (own) { gm[j] = spart[rp][(j - g0) & (NG - 1)][0]; gq2[j] = spart[rp][(j - g0) & (NG - 1)][1]; }
                    else { gm[j] = 0.f; gq2[j] = 0.f; }
                }
                bool give_up = false;
                if (!__all(ok) && it >= 64 && (it & 63) == 0) {
                    const long long now = wall_clock64();
                    if (t0 == 0) t0 = now;
                    give_up = now - t0 > PG_LN_TIMEOUT_TICKS || __hip_atomic_load(a.ln_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
                    if (give_up && lane == 0) __hip_atomic_store(a.ln_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // (pinned host word)
                }
#ifdef OPH_ABLATE
                if (a.dbg & 16) give_up = true;           // measurement: do not wait for the partners (wrong statistics)
#endif
                if (__all(ok) || give_up) {
                    float sm = 0.f;
#pragma unroll
                    for (int j = 0; j < 16; ++j) sm += gm[j];
                    mean = sm / (float)ngt;
                    float q = 0.f;
#pragma unroll
                    for (int j = 0; j < 16; ++j) q += gq2[j];
#pragma unroll
                    for (int j = 0; j < 16; ++j) if (j < ngt) { const float dl = gm[j] - mean; q = fmaf(64.0f * dl, dl, q); }
                    rstd = fast_rsqrt(q / (float)a.N + LN_EPS);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
#ifdef OPH_ABLATE
        if ((a.dbg & 64) && blockIdx.x == 0 && tid == 0) __hip_atomic_store(a.ln_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // fault injection
#endif
        if (gq == 0) { srow[rp][0] = mean; srow[rp][1] = rstd; }
        __syncthreads();
        // ---- normalise, store fp32 rows and the next layer's planes: thread <-> (row, 8 consecutive channels of one phase): two 16-byte
        //      fp32 stores and one 16-byte store per plane
        constexpr int C8 = NCOLS / 8, ITS = 128 * C8 / (64 * WV), PC8 = BNC / 8;
        const int c8 = tid % C8;                                   // (64 WV is a multiple of C8: the same columns in every iteration)
        const int ph = c8 / PC8, ch = n0 + (c8 % PC8) * 8;
        const size_t rows2 = (size_t)2 * a.M;
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
            const int idx = it * 64 * WV + tid, row = idx / C8;
            if (m0 + row >= a.M) continue;
            f32x4 v0 = *(const f32x4*)(Os + row * LDO + c8 * 8), v1 = *(const f32x4*)(Os + row * LDO + c8 * 8 + 4);
            const float mu = srow[row * 2 + ph][0], rs = srow[row * 2 + ph][1];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v0[e] = (v0[e] - mu) * rs * ln_g[0][e] + ln_b[0][e]; v1[e] = (v1[e] - mu) * rs * ln_g[1][e] + ln_b[1][e]; }
            const size_t orow = (size_t)2 * (m0 + row) + ph;
            *(f32x4*)(a.Y + orow * a.ldy + ch) = v0;
            *(f32x4*)(a.Y + orow * a.ldy + ch + 4) = v1;
#ifdef OPH_ABLATE
            if (a.dbg & 32) continue;                     // measurement: no plane stores
#endif
            if (a.Yh) {
                h16x8 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    hi[e] = (h16)v0[e]; lo[e] = (h16)(v0[e] - (float)hi[e]);
                    hi[4 + e] = (h16)v1[e]; lo[4 + e] = (h16)(v1[e] - (float)hi[4 + e]);
                }
                const size_t o = ((size_t)(ch >> 5) * rows2 + orow) * 32 + (ch & 31);
                *(h16x8*)(a.Yh + o) = hi;
                *(h16x8*)(a.Yl + o) = lo;
            }
        }
        return;
    }
    if constexpr (!(DBG & 4)) {
        constexpr int C4 = NCOLS / 4, ITS = 128 * C4 / (64 * WV);      // 16-byte pieces per row; per thread
        auto out = [&](auto guarded) {
            f32x4 v[ITS];
#pragma unroll
            for (int it = 0; it < ITS; ++it) { const int idx = it * 64 * WV + tid; v[it] = *(const f32x4*)(Os + (idx / C4) * LDO + (idx % C4) * 4); }
#pragma unroll
            for (int it = 0; it < ITS; ++it) {
                const int idx = it * 64 * WV + tid, row = idx / C4, c4 = idx % C4;
                float* dst = a.H + (size_t)(m0 + row) * a.ldh + (CONVT ? (size_t)(c4 / (BNC / 4)) * (a.ldh >> 1) + n0 + (c4 % (BNC / 4)) * 4 : (size_t)n0 + c4 * 4);
                if (!decltype(guarded)::value || m0 + row < a.M) *(f32x4*)dst = v[it];
            }
        };
        // (a full tile stores without per-row branches, see conv_gemm_f32_body)
        if (m0 + 128 <= a.M) out(std::false_type{}); else out(std::true_type{});
    }
}

template <int NT, bool CONVT, int WV = 4, int DBG = 0, bool LNF = false>
static void launch_plane_gemm_t(const PlaneGemmArgs& a, hipStream_t s) {
    typedef PlaneGemmCfg<NT, CONVT, WV> Cfg;
    constexpr size_t lds = Cfg::LDS_BYTES;
    static bool attr_set[64] = {false};          // function attributes are per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_set[dev & 63]) {
        (void)hipFuncSetAttribute((const void*)plane_gemm<NT, CONVT, WV, DBG, LNF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set[dev & 63] = true;
    }
    const int MT = (a.M + 127) / 128, NTL = (a.N + Cfg::BNC - 1) / Cfg::BNC;
    const int grid = LNF ? (MT + 7) / 8 * 8 * NTL : MT * NTL;          // (fused LayerNorm: whole groups of 8 row tiles, see the kernel)
    hipLaunchKernelGGL((plane_gemm<NT, CONVT, WV, DBG, LNF>), dim3(grid), dim3(64 * WV), lds, s, a);
}
size_t plane_gemm_ln_stats_bytes(int M, int N) { return (size_t)((M + 127) / 128) * (size_t)(N >> 6) * 256 * 16; }
// a.convt: conv1d_transpose (taps x[t], x[t-1] on the even phase's planes Wh / Wl, x[t] on the odd phase's Wh2 / Wl2; raw rows
// interleaved, a.ldh = 2 Nalloc); otherwise a.ntaps = 1 or 3 taps at offsets a.off[] (|offset| <= PLANE_GEMM_HALO).
// a.waves: 0 = chosen here, 4 / 8 forced (measurement)
void launch_plane_gemm(const PlaneGemmArgs& a, hipStream_t s) {
    const int MT = (a.M + 127) / 128;
    if (a.convt) {
#ifdef OPH_ABLATE      // the ablation builds of the kernel behind DESIGN.md section 10.7 (oph_bench_conv1d_transpose precisions 6..9)
        if (a.dbg == 1) return launch_plane_gemm_t<2, true, 4, 1>(a, s);
        if (a.dbg == 2) return launch_plane_gemm_t<2, true, 4, 2>(a, s);
        if (a.dbg == 4) return launch_plane_gemm_t<2, true, 4, 4>(a, s);
        if (a.dbg == 8) return launch_plane_gemm_t<2, true, 4, 8>(a, s);
#endif
        // more 64-channel workgroups than CUs would take two rounds at one per CU: 128 channels per workgroup, 8 waves
        const bool wide = a.waves ? a.waves == 8 : (MT * ((a.N + 63) / 64) > 256 && a.N % 128 == 0);
        if (a.ln_gamma) {          // LayerNorm inside the launch (the caller has checked N % 64 == 0, N <= 1024 and the exchange region)
            if (wide) launch_plane_gemm_t<2, true, 8, 0, true>(a, s); else launch_plane_gemm_t<2, true, 4, 0, true>(a, s);
            return;
        }
        if (wide) launch_plane_gemm_t<2, true, 8>(a, s); else launch_plane_gemm_t<2, true, 4>(a, s);
    } else if (a.ntaps == 1) {
#ifdef OPH_ABLATE      // (the 8-wave forms of the k = 1 / 3-tap layers measured slower, DESIGN.md section 10.7: measurement builds only)
        if (a.waves == 8) return launch_plane_gemm_t<1, false, 8>(a, s);
#endif
        launch_plane_gemm_t<1, false, 4>(a, s);
    } else {
#ifdef OPH_ABLATE
        if (a.waves == 8) return launch_plane_gemm_t<3, false, 8>(a, s);
#endif
        launch_plane_gemm_t<3, false, 4>(a, s);
    }
}
bool plane_gemm_ok(int ntaps, const int* off, int kc, bool convt) {
    if (kc % 32) return false;
    if (convt) return true;
    if (ntaps != 1 && ntaps != 3) return false;
    for (int t = 0; t < ntaps; ++t) if (off[t] > PLANE_GEMM_HALO || off[t] < -PLANE_GEMM_HALO) return false;
    if (off[ntaps >> 1] != 0) return false;         // the centre tap is never masked ('same' padding; the kernel relies on it)
    return true;
}

// a 2-byte plane [rows][ld] (k contiguous) -> K-blocked [ld / 32][rows][32]; 16 bytes per thread
__global__ __launch_bounds__(256) void kblock_planes_k(const uint4* src, uint4* dst, int rows, int ld) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int per_row = ld >> 3;
    const size_t r = i / per_row;
    if (r >= (size_t)rows) return;
    const int c8 = (int)(i - r * per_row);          // 8-element piece of the row
    dst[((size_t)(c8 >> 2) * rows + r) * 4 + (c8 & 3)] = src[i];
}
void launch_kblock_planes(const void* src, void* dst, int rows, int ld, hipStream_t s) {
    const size_t n = (size_t)rows * (ld >> 3);
    hipLaunchKernelGGL(kblock_planes_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const uint4*)src, (uint4*)dst, rows, ld);
}

// fp32 rows [M][ld] -> fp16 hi / lo planes, K-blocked [kc / 32][M][32] (what ln_rows writes for the layers behind it; this kernel
// serves inputs that no LayerNorm launch produced: the per-operator entry points and the timing of a single layer)
__global__ __launch_bounds__(256) void rows_to_planes_k(const float* x, int ld, int M, int kc, h16* ph, h16* pl) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;          // one thread = 4 channels of one row
    const int per_row = kc >> 2;
    const size_t m = i / per_row;
    if (m >= (size_t)M) return;
    const int c = (int)(i - m * per_row) * 4;
    const f32x4 v = *(const f32x4*)(x + m * ld + c);
    h16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) { hi[e] = (h16)v[e]; lo[e] = (h16)(v[e] - (float)hi[e]); }
    const size_t o = ((size_t)(c >> 5) * M + m) * 32 + (c & 31);
    *(h16x4*)(ph + o) = hi;
    *(h16x4*)(pl + o) = lo;
}
void launch_rows_to_planes(const float* x, int ld, int M, int kc, void* ph, void* pl, hipStream_t s) {
    const size_t n = (size_t)M * (kc >> 2);
    hipLaunchKernelGGL(rows_to_planes_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, ld, M, kc, (h16*)ph, (h16*)pl);
}

}  // namespace oph
