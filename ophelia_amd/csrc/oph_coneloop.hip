// cone_loop: the AudioDec history cone of EVERY decode step in ONE persistent launch (gfx950), beside dec_loop.
//
// Why a cone at all: the reference re-evaluates AudioDec's whole history under the CURRENT attention mask every step
// (networks.py:311 tiles one mask over max_T; synthesize.py:181-183 re-runs the graph), so row t of AudioDec needs its
// receptive cone recomputed: the highway-layer inputs at the history offsets Hset[k] (84, 82, 44, 14, 4, 2 positions for
// rates 1,3,9,27,1,1) x 16 utterances -- 2.27 GFLOP per step, networks.py:360-435, modules.py:148-207.
//
// Round 2 ran it as nine dependent launches per step (cone_head, split-K GEMM + ln_rows pairs, cone_fc16): every launch has
// a ~5 us floor, split-K partials are written and re-read, LayerNorm is a separate pass.  Here the cone is a task graph
// executed by resident workgroups:
//   level 0 task (position)          : attention rows under the current mask through the cached V.Wc / Q.Wq terms + LayerNorm
//                                      (what cone_head does), 16 rows
//   level k task (position, cg)      : highway layer k-1 at one position, 16 rows x 768 K, for the 64 raw columns of column
//                                      group cg = 32 channels of H1 and the SAME 32 channels of H2, on v_mfma_f32_16x16x4_f32
//                                      (exact fp32); bias; the row statistics of the two LayerNorms are exchanged between the
//                                      eight column groups of the position as {epoch, value} granules (mean and M2 of 32
//                                      columns each, combined exactly: M2 = sum M2_i + n sum (mean_i - mean)^2); then
//                                      LayerNorm x 2 + sigmoid gate + highway mix in registers, rows stored write-through.
// No raw rows, no partials and no separate LayerNorm pass ever reach memory.  Tasks are handed over through per-(level,
// position) counters (producers: write-through stores, s_waitcnt vmcnt(0), one relaxed atomic add; consumers: poll the three
// counters of their taps, then coherent loads): a level-k task starts as soon as ITS three input positions are written, so
// the levels overlap, and the cone of step t+1 may start (after the attention of step t) while the last tasks of step t are
// still running -- buffers, counters and statistics are double-buffered by step parity.
//
// Placement: workgroup b serves column group b % 8 (on this chip workgroup b runs on XCD b % 8, so the 192 KB weight slice
// of a column group stays in ONE XCD's L2 -- for speed only: any placement is correct).  Within a column group the tasks
// are dealt round-robin to its workgroups in the global order (step, level, position); every workgroup runs its tasks in
// that order, every dependency points to an earlier task and the eight siblings of a statistics exchange sit on eight
// different workgroups, so with all workgroups resident (the launcher sizes the grid by the occupancy query) the graph
// cannot dead-lock; all spins are bounded anyway and set the decode's error word.
//
// The two ends towards dec_loop are the words it already uses: sig[0] (attention of step t-1 done => cone(t) may start) and
// one word per level (its two tap positions written => the AudioDec layer reading them may go on).
#include "oph_internal.h"
#include "oph_device.h"

#include <map>

namespace oph {

typedef unsigned long long u64;
constexpr long long CL_TIMEOUT_TICKS = 200000000LL;     // 2 s of the 100 MHz clock
constexpr int CL_LDX = 772;                             // xs row stride: 768 + 4 -> row m starts at bank 4 m; the four k-quarters
                                                        // are 192 floats = 0 banks apart: every ds_read_b128 group is conflict-free
constexpr int CL_NPF = 16;                              // weight fragments in flight per wave: a fragment comes from L2 or (evicted between steps) from the Infinity Cache, ~1.1 us; 24 x 55 ns of MFMAs cover it

static __device__ __forceinline__ float row16_sum(float v) {      // total over the 16 lanes of a DPP row, in every lane of the row
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return v;
}
static __device__ __forceinline__ unsigned ld_u32(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// one thread spins until *p >= want (wrap-safe); false on time-out / error elsewhere
static __device__ bool spin_ge(const unsigned* p, unsigned want, int* err, int code, int scope_system) {
    long long t0 = 0;
    for (int it = 0;; ++it) {
        const unsigned v = scope_system ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : ld_u32(p);
        if ((int)(v - want) >= 0) return true;
        __builtin_amdgcn_s_sleep(2);
        if ((it & 127) == 127) {
            const long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            if (now - t0 > CL_TIMEOUT_TICKS || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
}

__global__ __launch_bounds__(256, 2) void cone_loop(ConeLoopArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                               // [16][CL_LDX] operand rows: [tap x[o+2r] | tap x[o+r] | current x[o]]
    float* part_s = smem + 16 * CL_LDX;             // [4 waves][16 rows] row sums of a wave's 16 columns
    float* part_q = part_s + 64;                    // [4][16] sums of squared deviations
    float* stl = part_q + 64;                       // [8 cg][64] the position's statistics granules, landed
    float* rs = stl + 512;                          // [2 halves][16 rows][2] mean, rstd
    float* h2t = rs + 64;                           // [16][32] LN2(H2) of this column group's channels
    float* yt = h2t + 512;                          // [16][32] the task's output rows
    int* tapi = (int*)(yt + 512);                   // [0..2] source position per tap (-1: before the utterance's start), [3] go-on flag, [4..5] claimed task indices
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = blockIdx.x & 7;
    const int r16 = lane & 15, kq = lane >> 4;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    int* const stop_word = a.ctl + 1;
    int* const err = a.ctl + 2;
    const int d = a.d;                               // 256
    if (a.stamps && tid == 0) {                      // diagnostics: the XCD every workgroup runs on ([stamps 8 (2 max_T + 4)] onwards)
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
        a.stamps[(size_t)(2 * a.t_end + 4) * 8 + blockIdx.x] = (long long)(xcc & 15u);
    }
    // ---- this column group's task list of ONE step, in dependency order: its level-0 positions (pos = cg, cg + 8, ...), then
    //      every position of level 1, level 2, ...  The list is dealt round-robin to the column group's workgroups (task
    //      index i -> workgroup slot i % nslots; the index runs on over the steps), and every workgroup runs its tasks in
    //      that order.  Every dependency of a task has a smaller index in its own or a sibling list and the eight siblings of a
    //      statistics exchange sit at (nearly) the same index of eight different lists, so with all workgroups resident the
    //      graph cannot dead-lock.  (Measured alternatives: pulling tasks from a per-group counter, 30.5 vs 25.5 ms per batch --
    //      the siblings of an exchange drift apart; pulling with one task claimed ahead, 150 ms -- a claimed but not
    //      started task makes the exchanges convoy.)
    const int n0 = (a.npos0 - cg + 7) >> 3;
    int per_step = n0;
    for (int k = 1; k < a.nlevels; ++k) per_step += a.L[k].npos;
    const int slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    auto level_of = [&](int idx, int& t, int& k, int& pos) {        // task index -> (step, level, position)
        t = a.t_begin + idx / per_step;
        int o = idx % per_step;
        if (o < n0) { k = 0; pos = cg + 8 * o; return; }
        o -= n0;
        for (k = 1; k < a.nlevels - 1 && o >= a.L[k].npos; ++k) o -= a.L[k].npos;
        pos = o;
    };
    f32x4 bfrag[CL_NPF];
    int pf_level = -1;                               // level whose first CL_NPF weight fragments are already in (or on their way to) bfrag
    int t_live = a.t_begin - 1;                      // last step known to be released (attention of step t-1 done) and not stopped
    for (int cur = slot;; cur += nslots) {
        int t, k, pos;
        level_of(cur, t, k, pos);
        if (t >= a.t_end) break;
        if (t != t_live) {
            // ---- the cone of step t may start once the attention of step t-1 is done (p_t, Q[t-1], QW[t-1] written through)
            if (tid == 0) {
                const bool ok = spin_ge(a.sig, a.sig_base + (unsigned)t, err, 3, 1);
                const int stop_v = __hip_atomic_load(stop_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // dec_loop breaks at step t when the stop step is <= t - 2 (step stop+1 still runs and polls its cone)
                tapi[3] = (ok && stop_v > t - 2 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) ? 1 : 0;
                if (a.stamps && cur % per_step == 0 && cg == 0) a.stamps[t * 8] = wall_clock64();
            }
            __syncthreads();
            if (!tapi[3]) break;
            t_live = t;
        }
        const int par = t & 1;
        const unsigned done_par = (unsigned)((t - a.t_begin) / 2 + 1);      // steps of this parity so far, this one included
        const unsigned want_sig = a.sig_base + (unsigned)t;

        if (k == 0) {
            // ================= level 0: the cone head at one position, 16 rows =================
            const int tq = t - a.off0[pos];
            if (tq >= 0) {
                const int c = lane * 4;
                const float scale = 1.0f / sqrtf((float)d);          // tf.rsqrt(tf.to_float(hp.d))  networks.py:300
                const f32x4 g0 = *(const f32x4*)(a.gamma0 + c), be0 = *(const f32x4*)(a.beta0 + c);
                // 4 rows per wave in two passes of two, every request of a round issued before the first use
#pragma unroll 1
                for (int half = 0; half < 2; ++half) {
                    int b[2], p[2];
                    f32x4 q[2], qw[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        b[i] = w + 4 * (2 * half + i);
                        p[i] = __hip_atomic_load(a.p + par * 16 + b[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        q[i] = ld_coherent(a.Q + ((size_t)tq * 16 + b[i]) * d + c);
                        qw[i] = ld_coherent(a.QW + ((size_t)tq * 16 + b[i]) * d + c);
                    }
                    constexpr int AW = 4;
                    f32x4 kv[2][AW], vw[2][AW];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const float* Kb = a.KV + (size_t)b[i] * a.N_keys * 2 * d;
                        const float* VWb = a.VW + (size_t)b[i] * a.N_keys * a.ldvw;
#pragma unroll
                        for (int ww = 0; ww < AW; ++ww) {
                            const bool in = ww < a.win && p[i] + ww < a.N_keys && b[i] < a.B;
                            kv[i][ww] = in ? *(const f32x4*)(Kb + (size_t)(p[i] + ww) * 2 * d + c) : zero4;
                            vw[i][ww] = in ? *(const f32x4*)(VWb + (size_t)(p[i] + ww) * a.ldvw + c) : zero4;
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        // attention window [p, p+win) under the CURRENT mask (networks.py:300-315), as cone_head_row
                        const int nwin = min(a.win, a.N_keys - p[i]);
                        float sc[AW], mx = -INFINITY;
#pragma unroll
                        for (int ww = 0; ww < AW; ++ww) {
                            sc[ww] = -INFINITY;
                            if (ww < nwin) {
                                sc[ww] = wave_sum(q[i][0] * kv[i][ww][0] + q[i][1] * kv[i][ww][1] + q[i][2] * kv[i][ww][2] + q[i][3] * kv[i][ww][3]) * scale;
                                mx = fmaxf(mx, sc[ww]);
                            }
                        }
                        float den = 0.f, pr[AW];
#pragma unroll
                        for (int ww = 0; ww < AW; ++ww) { pr[ww] = ww < nwin ? expf(sc[ww] - mx) : 0.f; den += pr[ww]; }
                        f32x4 hh = qw[i];
#pragma unroll
                        for (int ww = 0; ww < AW; ++ww)
                            if (ww < nwin) {
                                const float pw = pr[ww] / den;
#pragma unroll
                                for (int n = 0; n < 4; ++n) hh[n] = fmaf(pw, vw[i][ww][n], hh[n]);
                            }
                        // LayerNorm (modules.py:137-139; C_1 has no activation)
                        const float invd = 1.0f / (float)d;
                        const float mean = wave_sum(hh[0] + hh[1] + hh[2] + hh[3]) * invd;
                        float qq = 0.f;
#pragma unroll
                        for (int n = 0; n < 4; ++n) { const float dl = hh[n] - mean; hh[n] = dl; qq += dl * dl; }
                        const float rstd = 1.0f / sqrtf(wave_sum(qq) * invd + LN_EPS);
                        f32x4 o;
#pragma unroll
                        for (int n = 0; n < 4; ++n) o[n] = hh[n] * rstd * g0[n] + be0[n];
                        st_coherent(a.rows0[par] + ((size_t)pos * 16 + b[i]) * d + c, o);
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(a.flags + (par * CL_MAX_LEVELS + 0) * CL_MAX_POS + pos, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (pos == a.sig0_pos0 || pos == a.sig0_pos1) {
                    const unsigned ntap = a.sig0_pos0 == a.sig0_pos1 ? 1u : 2u;
                    const unsigned old = __hip_atomic_fetch_add(a.levelcnt + par * CL_MAX_LEVELS + 0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (old + 1u == ntap * done_par) {
                        __hip_atomic_fetch_max(a.sig + LOOP_SIG_LEVEL0, want_sig, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        if (a.stamps) a.stamps[t * 8 + 1] = wall_clock64();
                    }
                }
            }
        } else {
            // ================= level k >= 1: highway layer k-1 at one position, this column group's 64 raw columns =================
            const ConeLoopLevel& L = a.L[k];
            // diagnostics: the phases of ONE sample task per step (level 4, first position, column group 0)
            long long* const tst = (a.stamps && k == 4 && pos == 0 && cg == 0 && tid == 0) ? a.stamps + (size_t)(a.t_end + 1 + t) * 8 : nullptr;
#define CL_STAMP(K) do { if (tst) tst[K] = wall_clock64(); } while (0)
            CL_STAMP(0);
            // ---- taps: source positions in level k-1 (oldest first; the third is this position itself = the residual)
            if (tid < 3) {
                int src = L.tab[tid * L.npos + pos];
                if (t < L.need[tid * L.npos + pos]) src = -1;                 // before the utterance's start: zeros (causal padding)
                else {
                    const unsigned per = k == 1 ? 1u : 8u;                    // tasks per position of the producing level
                    if (!spin_ge(a.flags + (par * CL_MAX_LEVELS + (k - 1)) * CL_MAX_POS + src, per * done_par, err, 2, 0)) src = -1;
                }
                tapi[tid] = src;
            }
            // weight fragments of this wave, in the lanes' order: [cg][wave][chunk][lane][4] (1 KB per request); the first CL_NPF
            // were requested when the previous task's contraction ended, if this task was known then
            const f32x4* wsw = (const f32x4*)L.Wsw + ((size_t)(cg * 4 + w) * CL_NCH) * 64 + lane;
            if (pf_level != k) {
#pragma unroll
                for (int i = 0; i < CL_NPF; ++i) bfrag[i] = wsw[i * 64];
            }
            pf_level = -1;
            __syncthreads();
            CL_STAMP(1);
            const bool live = tapi[2] >= 0;                                   // the position exists (block-uniform)
            f32x4 acc0 = zero4, acc1 = zero4;
            if (live) {
                // ---- gather the operand rows: 3 taps x 16 rows x 256 channels, coherent loads (written by other CUs in this launch)
                const float* prev = k == 1 ? a.rows0[par] : a.L[k - 1].rows[par];
                f32x4 g[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    const int tap = i >> 2, e = (i & 3) * 256 + tid, row = e >> 6, c4 = (e & 63) * 4;
                    const int src = tapi[tap];
                    if (src < 0) g[i] = zero4;
                    else if (a.dbg & 1) g[i] = ld_coherent(prev + ((size_t)src * 16 + row) * 256 + c4);
                    else g[i] = ld_sc1_b128(prev, (unsigned)((((size_t)src * 16 + row) * 256 + c4) * 4));
                }
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    const int tap = i >> 2, e = (i & 3) * 256 + tid, row = e >> 6, c4 = (e & 63) * 4;
                    *(f32x4*)(xs + row * CL_LDX + tap * 256 + c4) = g[i];
                }
                __syncthreads();
                CL_STAMP(2);
                // ---- 16 x 16 slice per wave on the 16x16x4 MFMA: lane (m = lane & 15, kq = lane >> 4) supplies
                //      x[m][192 kq + 4 i + e] and W[column][192 kq + 4 i + e] in step (i, e): any assignment of the K
                //      indices to the MFMA's four k slots sums the same products
                const float* xa = xs + r16 * CL_LDX + kq * (CL_NCH * 4);
#pragma unroll
                for (int i = 0; i < CL_NCH; ++i) {
                    const f32x4 av = *(const f32x4*)(xa + 4 * i);
                    const f32x4 bv = bfrag[i % CL_NPF];
                    if (i + CL_NPF < CL_NCH) bfrag[i % CL_NPF] = wsw[(i + CL_NPF) * 64];
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[0], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[1], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv[2], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv[3], acc1, 0, 0, 0);
                }
                CL_STAMP(3);
            }
            // ---- this workgroup's next task: request its first weight fragments now -- they travel while this task's statistics
            //      are exchanged, its rows stored, and the next task's taps awaited and gathered
            {
                int tn, kn, posn;
                level_of(cur + nslots, tn, kn, posn);
                if (kn >= 1 && tn < a.t_end && !(a.dbg & 2)) {
                    const f32x4* wn = (const f32x4*)a.L[kn].Wsw + ((size_t)(cg * 4 + w) * CL_NCH) * 64 + lane;
#pragma unroll
                    for (int i = 0; i < CL_NPF; ++i) bfrag[i] = wn[i * 64];
                    pf_level = kn;
                }
            }
            if (live) {
                // C/D layout: column = lane & 15, row = 4 (lane >> 4) + register.  Wave w: half = w >> 1 (0: H1, 1: H2),
                // channel = 32 cg + 16 (w & 1) + (lane & 15)
                const int half = w >> 1, ch = 32 * cg + 16 * (w & 1) + r16;
                const float bias = L.bias[half * 256 + ch];
                f32x4 x;
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = (acc0[e] + acc1[e]) + bias;
                // ---- statistics of this column group's 32 columns per row and half: mean and M2 (two passes, local)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float s = row16_sum(x[e]);
                    if (r16 == 0) part_s[w * 16 + 4 * kq + e] = s;
                }
                __syncthreads();
                float mloc[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    mloc[e] = (part_s[w * 16 + 4 * kq + e] + part_s[(w ^ 1) * 16 + 4 * kq + e]) * (1.0f / 32.0f);
                    const float dl = x[e] - mloc[e];
                    const float qv = row16_sum(dl * dl);
                    if (r16 == 0) part_q[w * 16 + 4 * kq + e] = qv;
                }
                __syncthreads();
                CL_STAMP(4);
                const unsigned ep = a.epoch0 + (unsigned)(t * CL_MAX_LEVELS + k);
                u64* sg = a.stats + ((size_t)((par * CL_MAX_LEVELS + k) * CL_MAX_POS + pos) * 8) * 64;
                if (w == 0) {       // lane l: row l >> 2, value l & 3 = {mean H1, M2 H1, mean H2, M2 H2}
                    const int row = lane >> 2, which = lane & 3, hf = which >> 1;
                    const float v = (which & 1) ? part_q[(2 * hf) * 16 + row] + part_q[(2 * hf + 1) * 16 + row]
                                                : (part_s[(2 * hf) * 16 + row] + part_s[(2 * hf + 1) * 16 + row]) * (1.0f / 32.0f);
                    __hip_atomic_store(sg + cg * 64 + lane, ((u64)ep << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                // ---- the eight column groups' statistics: 512 granules, two per thread, until every tag is this task's
                {
                    long long t0 = 0;
                    bool got0 = false, got1 = false;
                    for (int itp = 0;; ++itp) {
                        if (!got0) { const u64 v = __hip_atomic_load(sg + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if ((unsigned)(v >> 32) == ep) { stl[tid] = __uint_as_float((unsigned)v); got0 = true; } }
                        if (!got1) { const u64 v = __hip_atomic_load(sg + 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if ((unsigned)(v >> 32) == ep) { stl[256 + tid] = __uint_as_float((unsigned)v); got1 = true; } }
                        if (__all(got0 && got1)) break;
                        __builtin_amdgcn_s_sleep(1);
                        if ((itp & 127) == 127) {
                            const long long now = wall_clock64();
                            if (t0 == 0) t0 = now;
                            if (now - t0 > CL_TIMEOUT_TICKS || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                                if (lane == 0) __hip_atomic_store(err, 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                break;
                            }
                        }
                    }
                }
                __syncthreads();
                CL_STAMP(5);
                if (tid < 32) {     // (half, row): combine the eight (mean, M2) pairs of 32 columns each -- exact pooling
                    const int row = tid & 15, hf = tid >> 4;
                    float mu[8], m = 0.f, M2 = 0.f;
#pragma unroll
                    for (int c8 = 0; c8 < 8; ++c8) { mu[c8] = stl[c8 * 64 + row * 4 + 2 * hf]; m += mu[c8]; M2 += stl[c8 * 64 + row * 4 + 2 * hf + 1]; }
                    m *= 0.125f;
                    float dd = 0.f;
#pragma unroll
                    for (int c8 = 0; c8 < 8; ++c8) { const float dl = mu[c8] - m; dd += dl * dl; }
                    M2 += 32.0f * dd;
                    rs[(hf * 16 + row) * 2] = m;
                    rs[(hf * 16 + row) * 2 + 1] = fast_rsqrt(M2 * (1.0f / 256.0f) + LN_EPS);      // [TF-sem] biased variance, eps 1e-12
                }
                __syncthreads();
                // ---- g = sigmoid(LN1(H1)), u = LN2(H2), y = g u + (1 - g) x   (modules.py:194-203)
                const float gam = (half ? L.g2 : L.g1)[ch], bet = (half ? L.b2 : L.b1)[ch];
                f32x4 hn;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = 4 * kq + e;
                    hn[e] = (x[e] - rs[(half * 16 + row) * 2]) * rs[(half * 16 + row) * 2 + 1] * gam + bet;
                }
                if (half) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) h2t[(4 * kq + e) * 32 + 16 * (w & 1) + r16] = hn[e];
                }
                __syncthreads();
                if (!half) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int row = 4 * kq + e;
                        const float gte = fast_sigmoid(hn[e]);
                        yt[row * 32 + 16 * (w & 1) + r16] = gte * h2t[row * 32 + 16 * (w & 1) + r16] + (1.0f - gte) * xs[row * CL_LDX + 512 + ch];
                    }
                }
                __syncthreads();
                CL_STAMP(6);
                if (tid < 128) {    // 16 rows x 32 channels as 16-byte write-through stores
                    const int row = tid >> 3, c4 = (tid & 7) * 4;
                    st_coherent(L.rows[par] + ((size_t)pos * 16 + row) * 256 + 32 * cg + c4, *(const f32x4*)(yt + row * 32 + c4));
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the rows have left (write-through) before the counter moves
            __syncthreads();
            CL_STAMP(7);
            if (tid == 0) {
                __hip_atomic_fetch_add(a.flags + (par * CL_MAX_LEVELS + k) * CL_MAX_POS + pos, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (pos == L.sig_pos0 || pos == L.sig_pos1) {
                    const unsigned ntap = L.sig_pos0 == L.sig_pos1 ? 1u : 2u;
                    const unsigned old = __hip_atomic_fetch_add(a.levelcnt + par * CL_MAX_LEVELS + k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (old + 1u == 8u * ntap * done_par) {
                        __hip_atomic_fetch_max(a.sig + LOOP_SIG_LEVEL0 + 16 * k, want_sig, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        if (a.stamps && k < 7) a.stamps[t * 8 + 1 + k] = wall_clock64();
                    }
                }
            }
        }
    }
}

static size_t cone_loop_lds() { return (size_t)(16 * CL_LDX + 64 + 64 + 512 + 64 + 512 + 512 + 8) * 4; }

// workgroups of cone_loop that fit on one CU at once
int cone_loop_blocks_per_cu() {
    const size_t lds = cone_loop_lds();
    (void)hipFuncSetAttribute((const void*)cone_loop, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)cone_loop, 256, lds) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}
void launch_cone_loop(const ConeLoopArgs& a, int nwg, hipStream_t s) {
    static thread_local std::map<int, bool> done;
    const size_t lds = cone_loop_lds();
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!done[dev]) { (void)hipFuncSetAttribute((const void*)cone_loop, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); done[dev] = true; }
    hipLaunchKernelGGL(cone_loop, dim3(nwg), dim3(256), lds, s, a);
}

}  // namespace oph
