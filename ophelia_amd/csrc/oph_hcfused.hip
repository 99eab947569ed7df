// hc_fused: one gated highway conv layer (modules.hc, modules.py:148-207) over many rows in ONE launch -- contraction, the two
// LayerNorms, gate and highway mix -- for the levels of the AudioDec history cone (networks.py:360-435 re-evaluated under the
// current attention mask).
//
//   raw[m][n] = bias[n] + sum_tap sum_c x[src(m, tap)][c] . W[n][tap, c]        n in [0, 2C), C = 256, 3 taps
//   out[m]    = sigmoid(LN1(raw[m][:C])) * LN2(raw[m][C:]) + (1 - sigmoid(...)) * x[res(m)]
//
// Until round 4 a level was two launches (conv_gemm_bf16x3 writing raw rows, ln_rows re-reading them) with the activations split
// into fp16 hi / lo terms by VALU work inside the contraction's staging.  Here
//   * both operands arrive as fp16 hi / lo PLANES (the producing launch -- this kernel, or cone_head -- writes its output rows as
//     fp32 AND as planes): no conversion instructions in the K loop.  The planes are K-blocked -- activations [channel / 64][row][64],
//     weights [column tile][K-step][64 columns][64] -- so that the 8 rows of a wave's request are 1 KB contiguous: with row-major
//     planes they sat 512 or 1536 bytes apart, on a quarter of the L2 channels (K loop 39-49 GB/s per CU);
//   * the weights' columns are permuted so that a 64-column tile holds 32 H1 columns and the SAME 32 channels of H2: gate and mix
//     are local to the tile;
//   * LayerNorm needs whole-row statistics: the 8 column tiles of a 64-row block exchange per-row (mean, M2) partials of their 32
//     columns as 8-byte {epoch, value} granules (pooled exactly: Chan et al.), every workgroup then normalises its own columns.
//     The 8 workgroups of a row block have consecutive ids on one XCD (b % 8 equal) and are resident together; every spin is bounded.
// Arithmetic: a.b = ah.bh + al.bh + ah.bl on v_mfma_f32_32x32x16_f16, fp32 accumulation (the fp32 class of DESIGN.md section 9.1).
#include "oph_internal.h"
#include "oph_device.h"

#include <map>

namespace oph {

namespace {
typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef unsigned long long u64_t;
constexpr int HF_BM = 64, HF_BN = 64, HF_BK = 64, HF_C = 256, HF_K = 3 * HF_C, HF_NT = 2 * HF_C / HF_BN;      // 8 column tiles
constexpr int HF_AHEAD = 2;                           // K-steps of operand planes requested ahead (register sets)
constexpr long long HF_TIMEOUT_TICKS = 200000000LL;      // 2 s of the 100 MHz clock
}  // namespace

// (Round 4 also ran the small levels 3..5 on a second, unmasked stream -- level 2 handed over through a device word, write-through
// rows, sc1 loads -- so that they would overlap the next step's large levels: with a fourth queue busy beside the three CU-masked
// ones the queues time-slice, 50 ms per decode instead of 20.4; removed.)
static __device__ __forceinline__ void hc_fused_body(const HcFusedArgs& a) {
    // Every output row must get the SAME arithmetic whatever register or lane holds it (an utterance's result may not depend on its
    // row of the tile: tests/test_gpu_properties.py).  With contraction left to the compiler the unrolled epilogue got v_fma for some
    // register pairs and mul + add for others -- rows with (b & 1) ^ (b >> 3) set rounded differently (profiles/r04_alone.py).
    // Contraction off; the fused operations are spelled out with fmaf.
#pragma clang fp contract(off)
    constexpr int SLOTS = 2;                               // LDS buffers of the operand planes (32 KB each)
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    // A K-step is 64 wide: a row's 128 bytes of a plane are ONE cache line (32-wide steps fetched half of every line they touched:
    // the K loop ran at 39 GB/s per CU, profiles/r04).  [SLOTS][64 rows][64] halves, 16-byte chunks XOR-swizzled by (row >> 1) & 7
    h16* const Ah = (h16*)smem_f;
    h16* const Al = Ah + SLOTS * HF_BM * HF_BK;
    h16* const Bh = Al + SLOTS * HF_BM * HF_BK;
    h16* const Bl = Bh + SLOTS * HF_BN * HF_BK;
    // the epilogue's buffers alias the ring (free once the K loop is through)
    float* const h2s = smem_f;                             // [64 rows][32] LN2(H2) rows of this tile
    float* const ys = h2s + HF_BM * 32;                    // [64 rows][36] output rows of this tile (pad: the stores' 16-byte reads stay conflict-free)
    float* const srow = ys + HF_BM * 36;                   // [4 waves][32 rows][2] mean, rstd of the wave's half
    float* const red = smem_f + 8192;                      // [4 wave tiles][16][64 lanes] the second K-half's accumulators (32 KB in: clear of the above)

    // 8 waves: wave tile (wr, wc) = w & 3 as before, and the K-step's four 16-wide chunks split between waves 0-3 (chunks 0, 1) and
    // 4-7 (chunks 2, 3): two waves per SIMD overlap each other's LDS / MFMA / load-issue phases (a 4-wave workgroup alone on its CU
    // ran the K loop at 0.65 us per step whatever was in flight); the halves meet in LDS after the loop
    const int tid = threadIdx.x, lane = tid & 63, w8 = __builtin_amdgcn_readfirstlane(tid >> 6), w = w8 & 3, kpart = w8 >> 2;
    // workgroup -> (row block tm, column tile jt): the 8 tiles of a row block are the ids {64 q + 8 k + x : k} -- one XCD, one dispatch wave
    const int bid = blockIdx.x, xcd = bid & 7, jt = (bid >> 3) & 7, tm = (bid >> 6) * 8 + xcd;
    const int MT = (a.M + HF_BM - 1) / HF_BM;
    const bool active = tm < MT;
    // the stop word is written by the running decode kernel at any time: ONE thread reads it (past the L1), the workgroup shares
    // the answer -- every barrier below sits under a workgroup-uniform condition
    const int m0 = tm * HF_BM, n0 = jt * HF_BN;
    // The source-position tables of this wave's rows and the residual positions depend on the arguments alone: requested here, so that
    // they are one round trip TOGETHER with the stop word (round 5: the prologue was four dependent round trips -- stop word, tap
    // tables, residual table, first operands -- ~0.7 us each; now two)
    int needv[3], tabv[3], rapos = 0, rbpos = 0;
    {
        const int ipc = min((m0 + 8 * w8) >> 4, a.n_out - 1);
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) { needv[tap] = a.need[tap * a.n_out + ipc]; tabv[tap] = a.tab[tap * a.n_out + ipc]; }
        const int wr_ = (w8 & 3) >> 1;
        rapos = a.restab[min((m0 >> 4) + 2 * wr_, a.n_out - 1)]; rbpos = a.restab[min((m0 >> 4) + 2 * wr_ + 1, a.n_out - 1)];
    }
    __shared__ int live_s;
    if (tid == 0) live_s = !(a.t > __hip_atomic_load(a.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    __syncthreads();
    const bool live = live_s != 0;
    long long* const dbg = (a.dbg && bid == 0 && tid == 0) ? a.dbg : nullptr;      // diagnostics (OPH_RUN_STAMPS): phase stamps of workgroup 0
    if (dbg) dbg[0] = wall_clock64();
    if (active && live) {
        // ---- operand addressing: one request of a wave = 8 rows x 8 chunks of 16 bytes (1 KB contiguous in the K-blocked planes);
        //      wave w8 fills rows 8 w8 .. 8 w8 + 7 of a plane buffer.  Bpad = 16.
        const int pos = lane & 7, rq0 = 8 * w8 + (lane >> 3);       // this lane's chunk position and its row
        const h16* const Xh = (const h16*)a.Xh; const h16* const Xl = (const h16*)a.Xl;
        const h16* const Wh = (const h16*)a.Wh; const h16* const Wl = (const h16*)a.Wl;
        const h16* const zrow = (const h16*)a.zeros;
        // element offset of the source row per tap, or -1 (a tap outside the utterance: zeros).  A request's 8 rows lie within one
        // position (16 utterance rows), so the table entries are wave-uniform and all six loads are in flight together (as per-lane
        // loads under their row conditions they were six dependent round trips before the first operand request)
        const int gp = (pos ^ ((rq0 >> 1) & 7)) * 8;       // the chunk of the row this lane fetches (the swizzle, applied on the global side)
        int asrc[3];
        {
            const int ipw = (m0 + 8 * w8) >> 4, bq = rq0 & 15;
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) asrc[tap] = (ipw < a.n_out && a.j >= needv[tap]) ? (tabv[tap] * 16 + bq) * HF_BK + gp : -1;
        }
        // 32x32x16 fragment: lane l holds row (l & 31), k = 8 (l >> 5) .. +7 of the 16-wide chunk
        const int wr = w >> 1, wc = w & 1, r32 = lane & 31, kh = lane >> 5;
        const int colh = jt * 32 + r32;                    // channel of this lane within its half (wc: 0 = H1, 1 = H2)
        // the residual rows x[res(m)] of this lane's 16 output rows (H1 waves): requested first, used after the exchange.  The wave's
        // 32 rows are two positions (16 utterances each): rows (e & 3) + 4 kh + 8 ((e >> 2) & 1) of position 2 wr + (e >> 3)
        float xres[16];
        auto load_xres = [&]() {
            if (wc == 0 && kpart == 0) {
                const int ra = rapos * 16, rb = rbpos * 16;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int b = (e & 3) + 4 * kh + 8 * ((e >> 2) & 1);
                    const float* px = a.Xres + (size_t)((e >> 3) ? rb + b : ra + b) * HF_C + colh;
                    xres[e] = *px;
                }
            }
        };
        load_xres();
        // Operand planes: global -> registers -> LDS (16 bytes per lane and request; a wave's request covers 8 whole cache lines).
        // (Measured first, round 4: the same chunks through global_load_lds -- 39 GB/s per CU whatever the ring depth or the K-step
        // width, the LDS-DMA path's own limit; plain loads stream 2-3x that from L2 / MALL.)  Software pipeline, prefetch distance 2:
        // while step s is multiplied out of LDS buffer s & 1, step s + 1 sits in one register set and step s + 2 is in flight.
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        struct Regs { i32x4 ah[1], al[1], bh[1], bl[1]; };
        auto load_a = [&](int s, Regs& q) {
            const int tap = s >> 2;
            const size_t kb = (size_t)(s & 3) * a.in_rows * HF_BK;        // planes are K-blocked: [channel / 64][row][64]
            const int so = asrc[tap];
            const h16* gh = so >= 0 ? Xh + so + kb : zrow;
            const h16* gl = so >= 0 ? Xl + so + kb : zrow;
            q.ah[0] = *(const i32x4*)gh;
            q.al[0] = *(const i32x4*)gl;
        };
        auto load_b = [&](int s, Regs& q) {
            const size_t bo = (((size_t)jt * (HF_K / HF_BK) + s) * HF_BN + rq0) * HF_BK + gp;      // [column tile][K-step][64 columns][64]
            q.bh[0] = *(const i32x4*)(Wh + bo);
            q.bl[0] = *(const i32x4*)(Wl + bo);
        };
        auto load_step = [&](int s, Regs& q) { load_a(s, q); load_b(s, q); };
        auto store_step = [&](int buf, const Regs& q) {
#pragma unroll
            for (int i = 0; i < 1; ++i) {
                const int o = buf * HF_BM * HF_BK + (rq0 + 8 * i) * HF_BK + pos * 8;
                *(i32x4*)(Ah + o) = q.ah[i];
                *(i32x4*)(Al + o) = q.al[i];
                *(i32x4*)(Bh + o) = q.bh[i];
                *(i32x4*)(Bl + o) = q.bl[i];
            }
        };
        f32x16 acc, accs;                                  // main products / the two small terms: two independent MFMA chains
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[e] = 0.f; accs[e] = 0.f; }
        constexpr int NK = HF_K / HF_BK;                   // 12 K-steps
        auto compute = [&](int buf) {
            const int ao = buf * HF_BM * HF_BK + (wr * 32 + r32) * HF_BK;
            const int bo = buf * HF_BN * HF_BK + (wc * 32 + r32) * HF_BK;
            const int swr = (r32 >> 1) & 7;                // (wr * 32 and wc * 32 do not change (row >> 1) & 7)
#pragma unroll
            for (int kq = 0; kq < 2; ++kq) {
                const int kc = 2 * kpart + kq;
                const int ch = ((kc * 2 + kh) ^ swr) << 3;
                const h16x8 ah = *(const h16x8*)(Ah + ao + ch), al = *(const h16x8*)(Al + ao + ch);
                const h16x8 bh = *(const h16x8*)(Bh + bo + ch), bl = *(const h16x8*)(Bl + bo + ch);
                accs = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accs, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
                accs = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accs, 0, 0, 0);
            }
        };
        // prefetch distance HF_AHEAD: step s + 1 waits in registers, steps s + 2 .. s + HF_AHEAD are in flight (the loop is fully
        // unrolled: every register-set index is a constant)
        Regs q[HF_AHEAD];
#pragma unroll
        for (int s = 0; s < HF_AHEAD; ++s) load_step(s, q[s]);
        store_step(0, q[0]);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < NK; ++s) {
            if (s + HF_AHEAD < NK) load_step(s + HF_AHEAD, q[s % HF_AHEAD]);      // (its previous content, step s, is in LDS)
            compute(s & 1);
            if (s + 1 < NK) store_step((s + 1) & 1, q[(s + 1) % HF_AHEAD]);
            __syncthreads();
        }
        if (dbg) dbg[1] = wall_clock64();
        // the two K-halves of every wave tile meet in LDS
        if (kpart == 1) {
#pragma unroll
            for (int e = 0; e < 16; ++e) red[(w * 16 + e) * 64 + lane] = acc[e] + accs[e];
        }
        __syncthreads();
        // ---- epilogue (waves 0-3; the others only keep the barriers company).  C/D layout of the 32x32 MFMA: col = lane & 31,
        //      row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
        if (kpart == 0) {
        const float bv = a.bias[n0 + wc * 32 + r32];
        const float gamma = (wc ? a.g2 : a.g1)[colh], beta = (wc ? a.b2 : a.b1)[colh];
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = ((acc[e] + accs[e]) + red[(w * 16 + e) * 64 + lane]) + bv;
        // per-row (mean, M2) of this wave's 32 columns: 16-lane DPP row totals, the two 16-lane rows of a k-half added through one
        // cross-lane exchange per value (all 16 in flight together)
        auto row16 = [&](float v) -> float {
            v += dpp_mov<0xB1>(v);
            v += dpp_mov<0x4E>(v);
            v += dpp_mov<0x141>(v);
            v += dpp_mov<0x140>(v);
            return v;
        };
        float pm[16], pq[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) pm[e] = row16(acc[e]);
#pragma unroll
        for (int e = 0; e < 16; ++e) pm[e] = (pm[e] + __shfl_xor(pm[e], 16)) * (1.0f / 32.0f);
#pragma unroll
        for (int e = 0; e < 16; ++e) { const float dl = acc[e] - pm[e]; pq[e] = row16(dl * dl); }
#pragma unroll
        for (int e = 0; e < 16; ++e) pq[e] += __shfl_xor(pq[e], 16);
        // publish: region (tm, wr, half wc) = [tile jt][row 0..31][{mean, M2}] granules -- a tile's 32 rows are 512 contiguous bytes, written
        // by ONE store instruction (lane (r32, kh) holds row r32 when kh == (r32 >> 2) & 1, in accumulator slot (r32 & 3) + 4 (r32 >> 3)):
        // whole cache lines per writer.  (Laid out [row][tile] every 128-byte line collected eight 16-byte partial write-throughs
        // from eight CUs.)
        u64_t* const sg = a.stats + ((((size_t)tm * 2 + wr) * 2 + wc) * 32) * (HF_NT * 2);
        const u64_t tag = (u64_t)a.epoch << 32;
        {
            const int esel = (r32 & 3) + 4 * (r32 >> 3);
            float mv = pm[0], qv = pq[0];
#pragma unroll
            for (int e = 1; e < 16; ++e) { mv = esel == e ? pm[e] : mv; qv = esel == e ? pq[e] : qv; }
            if (kh == ((r32 >> 2) & 1)) {
                f32x4 g;
                g[0] = mv; g[1] = __uint_as_float(a.epoch); g[2] = qv; g[3] = __uint_as_float(a.epoch);
                st_sc1_b128((float*)sg, (unsigned)((jt * 32 + r32) * 16), g);
            }
        }
        if (dbg) dbg[2] = wall_clock64();
        // gather: lane l reads row (l >> 1) of tiles 4 (l & 1) .. +3 (16 bytes each); re-read until all carry the epoch
        {
            const float* gbase = (const float*)sg;
            const unsigned goff = (unsigned)((((lane & 1) * 4) * 32 + (lane >> 1)) * 16);
            float gm[4], gq[4];
            long long t0 = 0;
            for (int it = 0;; ++it) {
                u64_t gv[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 g = ld_sc1_b128(gbase, goff + (unsigned)i * 32 * 16);
                    gv[2 * i] = ((u64_t)__float_as_uint(g[1]) << 32) | __float_as_uint(g[0]);
                    gv[2 * i + 1] = ((u64_t)__float_as_uint(g[3]) << 32) | __float_as_uint(g[2]);
                }
                bool ok = true;
#pragma unroll
                for (int i = 0; i < 8; ++i) ok = ok && (unsigned)(gv[i] >> 32) == a.epoch;
                bool give_up = false;
                if (!__all(ok) && it >= 64 && (it & 63) == 0) {
                    const long long now = wall_clock64();
                    if (t0 == 0) t0 = now;
                    give_up = now - t0 > HF_TIMEOUT_TICKS || __hip_atomic_load(a.ctl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                    if (give_up && lane == 0) __hip_atomic_store(a.ctl + 2, 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (__all(ok) || give_up) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { gm[i] = __uint_as_float((unsigned)gv[2 * i]); gq[i] = __uint_as_float((unsigned)gv[2 * i + 1]); }
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            // pooled statistics of the row's 8 x 32 columns: mean = avg of the tile means, M2 = sum M2_j + 32 sum (mean_j - mean)^2
            float sm = (gm[0] + gm[1]) + (gm[2] + gm[3]);
            sm += dpp_mov<0xB1>(sm);                       // + the other four tiles (lane ^ 1)
            const float mean = sm * (1.0f / HF_NT);
            float q = (gq[0] + gq[1]) + (gq[2] + gq[3]);
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float dl = gm[i] - mean; q = fmaf(32.0f * dl, dl, q); }
            q += dpp_mov<0xB1>(q);
            const float rstd = fast_rsqrt(q * (1.0f / HF_C) + LN_EPS);
            if ((lane & 1) == 0) { srow[(w * 32 + (lane >> 1)) * 2] = mean; srow[(w * 32 + (lane >> 1)) * 2 + 1] = rstd; }
        }
        if (dbg) dbg[3] = wall_clock64();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the wave reads back what it wrote itself
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * kh;
            const float mean = srow[(w * 32 + row) * 2], rstd = srow[(w * 32 + row) * 2 + 1];
            acc[e] = fmaf((acc[e] - mean) * rstd, gamma, beta);
        }
        if (wc == 1) {
#pragma unroll
            for (int e = 0; e < 16; ++e) h2s[(wr * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh) * 32 + r32] = acc[e];
        }
        }       // kpart == 0
        __syncthreads();
        if (wc == 0 && kpart == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rt = wr * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                const float h2 = h2s[rt * 32 + r32];
                const float gte = fast_sigmoid(acc[e]);
                ys[rt * 36 + r32] = fmaf(gte, h2, (1.0f - gte) * xres[e]);
            }
        }
        __syncthreads();
        // ---- stores: thread <-> (row tid >> 2, 8 consecutive channels): 2 x 16 bytes of fp32, 16 bytes per plane
        if (tid < 256) {
            const int rt = tid >> 2, m = m0 + rt, c8 = (tid & 3) * 8;
            if (m < a.M) {
                const f32x4 y0 = *(const f32x4*)(ys + rt * 36 + c8), y1 = *(const f32x4*)(ys + rt * 36 + c8 + 4);
                const size_t o = (size_t)m * HF_C + jt * 32 + c8;
                const int pos_m = m >> 4;
                if (a.done_sig && (pos_m == a.coh0 || pos_m == a.coh1)) { st_coherent(a.Y + o, y0); st_coherent(a.Y + o + 4, y1); }     // a row a RUNNING kernel reads
                else { *(f32x4*)(a.Y + o) = y0; *(f32x4*)(a.Y + o + 4) = y1; }
                h16x8 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    hi[e] = (h16)y0[e]; lo[e] = (h16)(y0[e] - (float)hi[e]);
                    hi[4 + e] = (h16)y1[e]; lo[4 + e] = (h16)(y1[e] - (float)hi[4 + e]);
                }
                const int ch = jt * 32 + c8;
                const size_t po = ((size_t)(ch >> 6) * a.M + m) * HF_BK + (ch & 63);
                *(h16x8*)((h16*)a.Yh + po) = hi;
                *(h16x8*)((h16*)a.Yl + po) = lo;
            }
        }
    }
    if (dbg) dbg[4] = wall_clock64();
    // ---- completion of a cone level: the tap rows have left (write-through), one lane raises the level's word (as ln_rows does)
    if (a.done_sig && active) {
        const int p0 = m0 / a.Bpad, p1 = (min(m0 + HF_BM, a.M) - 1) / a.Bpad;
        const bool holds = (a.coh0 >= p0 && a.coh0 <= p1) || (a.coh1 >= p0 && a.coh1 <= p1);
        if (holds) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                const unsigned old = atomicAdd(a.done_count, 1u);
                if (old + 1u == a.done_target) {
                    __hip_atomic_fetch_max(a.done_sig, a.done_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (a.done_stamp) *a.done_stamp = wall_clock64();
                }
            }
        }
    }
}
__global__ __launch_bounds__(512) void hc_fused(HcFusedArgs a) { hc_fused_body(a); }

size_t hc_fused_lds_bytes() { return (size_t)2 * (2 * HF_BM + 2 * HF_BN) * HF_BK * 2; }      // (the epilogue's 18.2 KB alias the operand buffers)

// number of workgroups the launch will have / of those that do work (one per (row block, column tile))
int hc_fused_grid(int M) { const int MT = (M + HF_BM - 1) / HF_BM; return ((MT + 7) / 8) * 64; }
int hc_fused_active(int M) { return ((M + HF_BM - 1) / HF_BM) * HF_NT; }
// how many of the launch's workgroups hold rows of position `pos` or `pos2` (they arrive at the level's counter)
int hc_fused_holders(int M, int Bpad, int pos, int pos2) {
    const int MT = (M + HF_BM - 1) / HF_BM;
    int n = 0;
    for (int tm = 0; tm < MT; ++tm) {
        const int p0 = tm * HF_BM / Bpad, p1 = (std::min(tm * HF_BM + HF_BM, M) - 1) / Bpad;
        if ((pos >= p0 && pos <= p1) || (pos2 >= p0 && pos2 <= p1)) n += HF_NT;
    }
    return n;
}
void launch_hc_fused(const HcFusedArgs& a, hipStream_t s) {
    static thread_local std::map<int, bool> done;
    const size_t lds = hc_fused_lds_bytes();
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!done[dev]) { (void)hipFuncSetAttribute((const void*)hc_fused, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); done[dev] = true; }
    hipLaunchKernelGGL(hc_fused, dim3(hc_fused_grid(a.M)), dim3(512), lds, s, a);
}
// workgroups of a launch that can be resident per CU
int hc_fused_blocks_per_cu(int M) {
    (void)M;
    const size_t lds = hc_fused_lds_bytes();
    (void)hipFuncSetAttribute((const void*)hc_fused, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)hc_fused, 512, lds) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

}  // namespace oph
