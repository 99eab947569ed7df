// dec_run: a RUN of decoder layers of one time step in ONE launch (gfx950).
//
// The decode step of synth_codedtext2mel (synthesize.py:181-209) is a chain of ~25 dependent layers over the batch's
// rows (AudioEnc networks.py:214-284 -> Attention 286-325 -> AudioDec 360-435), each needing all channels of the
// previous one.  As one launch per layer the chain costs ~5-7 us per layer of launch boundary + cold loads (profiles/r01).
// Here the workgroups stay resident over the whole run.  Workgroup (g, rg) owns output columns [16g, 16g+16) of every
// layer for the R rows (utterances) of row group rg: one wave per row runs the cheap row-local prologue (LayerNorm /
// gate / activation / attention), then the R x 16 slice is contracted with VALU FMAs (fp32 FMA and the fp32-input MFMA
// run at the same rate on gfx950, and without the MFMA's 16-row tile a workgroup can be R = 4 rows thin: 128 small
// workgroups, one wave per SIMD, instead of 32 fat ones whose 16 waves queue on 4 SIMDs -- profiles/r02 stamps).
// The raw slices travel between workgroups as 8-byte {epoch, value} granules written and polled with relaxed
// agent-scope atomics (cdna_hip_programming.md Guideline 16, recipe R2: the data is its own flag, so no fence and no
// separate flag round trip).  Rows are independent: a row group only ever waits for its own 32 column slices.
// Weights, LayerNorm parameters and the dilated-tap rows of layer l+1 do not depend on activations and are requested
// while layer l's partial sums are still being reduced.
//
// Arithmetic: fp32 throughout; every dot product is an fmaf chain over 4-wide k groups, K split round-robin over the
// R waves and the 4 k-quarters of a 16-wide chunk, partials summed in a fixed order.
#include "oph_internal.h"
#include "oph_device.h"
#include "oph_loopdev.h"

#include <map>

namespace oph {

template <int R>       // rows (= waves) per workgroup
__global__ __launch_bounds__(64 * R) void dec_run(RunArgs a) {
    constexpr int PF = (RUN_KMAX / 16 + R - 1) / R;     // 16-wide k-chunks per wave
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* part = smem;                          // [R waves][4 k-quarters][16 columns][R rows] partial sums
    float* xs = smem + R * 4 * 16 * R;           // [R][ldxs] this layer's R x Ktot operand: [taps (oldest first) | current]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave index as a scalar: row pointers live in SGPRs
    const int g = blockIdx.x, n0 = g * 16, row0 = blockIdx.y * R, grow = row0 + w;    // wave w <-> row w of the group
    const int r16 = lane & 15, kq = lane >> 4, c = lane * 4;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int stop_v = a.stop_after ? *a.stop_after : 0x7fffffff;
    const bool live = a.t <= stop_v;
    const int spk = a.spk_ids ? a.spk_ids[grow < a.B ? grow : 0] : 0;

    f32x4 xprev = zero4;                         // the previous layer's input row = highway residual of this prologue
    f32x4 bfrag[PF], tp0 = zero4, tp1 = zero4;
    float bias_v = 0.f;
    // everything of a layer that does not depend on activations: weight fragments (Wt is [n][k], k contiguous; L2 ->
    // registers), dilated-tap rows, bias
    auto fetch_layer = [&](const RunLayer& L) {
        const int nch = (L.ntaps * L.kc) >> 4;
        const bool cols = n0 < L.N;
        const float* wrow = L.Wt + (size_t)(n0 + r16) * L.ldw + kq * 4;
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int ch = w + R * i;
            bfrag[i] = (cols && ch < nch) ? *(const f32x4*)(wrow + ch * 16) : zero4;
        }
        bias_v = cols ? L.bias[n0 + (tid & 15)] : 0.f;
        if (L.ntaps == 3) {
            tp0 = (L.tap0 && c < L.kc) ? *(const f32x4*)(L.tap0 + (size_t)grow * L.ldtap + c) : zero4;
            tp1 = (L.tap1 && c < L.kc) ? *(const f32x4*)(L.tap1 + (size_t)grow * L.ldtap + c) : zero4;
        }
    };
    fetch_layer(a.L[0]);
    // diagnostics: wave 0 of every column slice stamps the phases of every layer
    long long* const stp = (a.stamps && w == 0 && blockIdx.y == 0) ? a.stamps + (size_t)g * RUN_MAX_LAYERS * 8 : nullptr;
#define RUN_STAMP(K) do { if (stp && lane == 0) stp[l * 8 + (K)] = wall_clock64(); } while (0)

    for (int l = 0; l < a.nlayers; ++l) {
        const RunLayer& L = a.L[l];
        const int cin = L.cin;
        const bool cok = c < cin;
        const bool two = L.pre >= RUN_HC;
        const bool cols = n0 < L.N;
        if (L.N == 0 && g != 0) break;           // the mel frame is written by column slice 0 alone

        // ---- 1. requests that do not depend on the hand-off: LayerNorm parameters, attention window start
        f32x4 g1v = zero4, b1v = zero4, g2v = zero4, b2v = zero4;
        if (L.pre != RUN_COPY && cok) {
            g1v = *(const f32x4*)(L.g1 + c); b1v = *(const f32x4*)(L.b1 + c);
            if (two) { g2v = *(const f32x4*)(L.g2 + c); b2v = *(const f32x4*)(L.b2 + c); }
        }
        const int p = L.pre == RUN_ATTN ? __builtin_amdgcn_readfirstlane(a.pcur[grow]) : 0;

        // ---- 2. this wave's raw row of the producing layer
        RUN_STAMP(0);
        int passes = 0;
        f32x4 av = zero4, uv = zero4;
        if (L.src) {
            const float* sp = L.src + (size_t)grow * L.ldsrc;
            if (cok) {
                av = *(const f32x4*)(sp + c);
                if (two) uv = *(const f32x4*)(sp + cin + c);
            }
        } else {
            const u64* row = a.gbuf + ((size_t)(l - 1) * a.Bpad + grow) * RUN_GCOLS;
            passes = sweep_row(row, c, cok, cin + c, two, a.epoch0 + (unsigned)l, lane, a.err, av, uv);
        }
        RUN_STAMP(1);
        if (stp && lane == 0) stp[l * 8 + 6] = passes;

        // ---- 3. prologue math (one row per wave).  [TF-sem] layer_norm: mean, biased variance, eps 1e-12 (modules.py:65)
        f32x4 x = av;
        if (L.pre != RUN_COPY) {
            const float invc = __builtin_amdgcn_rcpf((float)cin);
            float s1 = av[0] + av[1] + av[2] + av[3], s2 = uv[0] + uv[1] + uv[2] + uv[3];
            s1 = wave_sum(s1);
            if (two) s2 = wave_sum(s2);
            const float m1 = L.nonorm ? 0.f : s1 * invc, m2 = L.nonorm ? 0.f : s2 * invc;
            float q1 = 0.f, q2 = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d1 = cok ? av[e] - m1 : 0.f, d2 = cok ? uv[e] - m2 : 0.f;
                av[e] = d1; uv[e] = d2;
                q1 += d1 * d1; q2 += d2 * d2;
            }
            q1 = wave_sum(q1);
            if (two) q2 = wave_sum(q2);
            const float r1 = L.nonorm ? 1.0f : fast_rsqrt(q1 * invc + LN_EPS);
            const float r2 = L.nonorm ? 1.0f : fast_rsqrt(q2 * invc + LN_EPS);
            if (two) {          // highway: g = sigmoid(LN1(H1)), y = g*LN2(H2) + (1-g)*x   (modules.py:194-203)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float h1 = av[e] * r1 * g1v[e] + b1v[e], h2 = uv[e] * r2 * g2v[e] + b2v[e];
                    const float gte = fast_sigmoid(h1);
                    x[e] = cok ? gte * h2 + (1.0f - gte) * xprev[e] : 0.f;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = cok ? fast_act(av[e] * r1 * g1v[e] + b1v[e], L.act) : 0.f;
            }
        }
        xprev = x;

        if (L.N == 0) {
            // mel frame t -> Y[b][t] and the next step's decoder input S[t+1] (synthesize.py:204-209)
            if (live && grow < a.B && c < a.ldy) {
                *(f32x4*)(a.Yout + ((size_t)grow * a.max_T + a.t) * a.ldy + c) = x;
                if (c < a.ldtm) *(f32x4*)(a.Ytm + ((size_t)(a.t + 1) * a.Bpad + grow) * a.ldtm + c) = x;
            }
            break;
        }

        // ---- 4. stage the operand row: [tap x[t-2r] | tap x[t-r] | current]
        const int Ktot = L.ntaps * L.kc, ldxs = Ktot + 4, xcur = (L.ntaps - 1) * L.kc;
        float* xrow = xs + w * ldxs;
        if (L.pre == RUN_ATTN) {
            // R' = concat(softmax(Q K^T / sqrt(d)) V, Q) for row t under the current mask (networks.py:300-319)
            const int d = cin;
            const float* KVb = a.KV + (size_t)grow * a.N_keys * 2 * d;
            // Same arithmetic as attend_window (oph_device.h) -- only the window [p, p+win) is unmasked -- but with run-time
            // loops and the window's logits / probabilities kept one per LANE (lane i <-> key p+i) instead of in
            // unrolled arrays: this kernel sits at the 128-VGPR cap of a 1024-thread workgroup.
            const int nwin = min(a.win, a.N_keys - p);
            const float scale = fast_rsqrt((float)d);        // tf.rsqrt(tf.to_float(hp.d))  networks.py:300
            float scl = -INFINITY;
            for (int i = 0; i < nwin; ++i) {
                const f32x4 kv = cok ? *(const f32x4*)(KVb + (size_t)(p + i) * 2 * d + c) : zero4;
                const float sdot = wave_sum(x[0] * kv[0] + x[1] * kv[1] + x[2] * kv[2] + x[3] * kv[3]) * scale;
                if (lane == i) scl = sdot;
            }
            float mx = -INFINITY;
            for (int i = 0; i < nwin; ++i) mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, scl), i)));
            float prl = lane < nwin ? __builtin_amdgcn_exp2f(1.4426950408889634f * (scl - mx)) : 0.f;
            float den = 0.f;
            for (int i = 0; i < ATT_WMAX; ++i) den += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, prl), i));
            prl = prl * __builtin_amdgcn_rcpf(den);
            int arg = 0;
            float best = -1.f;
            f32x4 ctx = zero4;
            for (int i = 0; i < nwin; ++i) {
                const float pi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, prl), i));
                if (pi > best) { best = pi; arg = i; }       // first maximum, like tf.argmax
                const f32x4 vv = cok ? *(const f32x4*)(KVb + d + (size_t)(p + i) * 2 * d + c) : zero4;
#pragma unroll
                for (int e = 0; e < 4; ++e) ctx[e] += pi * vv[e];
            }
            if (cok) {
                *(f32x4*)(xrow + c) = ctx;
                *(f32x4*)(xrow + d + c) = x;
            }
            for (int c2 = 2 * d + lane; c2 < L.kc; c2 += 64) xrow[c2] = 0.f;
            if (g == 0 && live && grow < a.B) {
                if (cok) *(f32x4*)(a.Qhist + ((size_t)a.t * a.Bpad + grow) * d + c) = x;
                if (lane < nwin) a.align[(size_t)grow * a.N_keys * a.max_T + (size_t)(p + lane) * a.max_T + a.t] = prl;
                if (lane == 0) {
                    const int m = p + arg;
                    a.pnext[grow] = m;
                    if (a.t_ends[grow] == a.max_T && m >= a.ends[grow]) {      // synthesize.py:218-228
                        a.t_ends[grow] = a.t;
                        const int old = atomicAdd(a.n_ended, 1);
                        if (old + 1 == a.B && a.stop_mode == 0) *a.stop_flag = a.t;
                    }
                }
            }
        } else {
            if (c < L.kc) *(f32x4*)(xrow + xcur + c) = x;
            for (int c2 = 256 + c; c2 < L.kc; c2 += 256) *(f32x4*)(xrow + xcur + c2) = zero4;
            if (L.ccat > 0)     // speaker embedding appended to the input (row 0 of the table reads as zeros, modules.py:38-40)
                for (int j = lane; j < L.ccat; j += 64)
                    xrow[xcur + cin + j] = spk == 0 ? 0.f : L.cat_table[(size_t)spk * L.ccat + j];
            if (L.ntaps == 3 && c < L.kc) {
                *(f32x4*)(xrow + c) = tp0;
                *(f32x4*)(xrow + L.kc + c) = tp1;
            }
            if (g == 0 && L.xstore && live && c < L.kc) *(f32x4*)(L.xstore + (size_t)grow * L.ldstore + c) = x;
        }
        RUN_STAMP(2);
        __syncthreads();
        RUN_STAMP(3);

        // ---- 5. R x 16 slice: lane (j = lane&15, kq = lane>>4) accumulates column n0+j over k-quarter kq of this wave's chunks
        if (cols) {
            // 16 independent fmaf chains per lane (R rows x 4 k-lanes of a chunk), chunks in groups of 256 k (one tap);
            // a chunk index past the layer's K is clamped to a valid address: its weights are zero
            constexpr int GC = 16 / R;           // chunks per wave in a 256-k group
            float acc[R][4];
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[r][e] = 0.f;
            const float* xa = xs + kq * 4;
            const int nch = Ktot >> 4;
#pragma unroll
            for (int grp = 0; grp < PF / GC; ++grp) {
                if (grp * 16 < nch) {
#pragma unroll
                    for (int ii = 0; ii < GC; ++ii) {
                        const int i = grp * GC + ii;
                        const int ch = min(w + R * i, nch - 1);
                        f32x4 xf[R];
#pragma unroll
                        for (int r = 0; r < R; ++r) xf[r] = *(const f32x4*)(xa + r * ldxs + ch * 16);   // same address in 16 lanes: LDS broadcast
#pragma unroll
                        for (int r = 0; r < R; ++r)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[r][e] = fmaf(xf[r][e], bfrag[i][e], acc[r][e]);
                    }
                }
            }
            float* pw = part + ((w * 4 + kq) * 16 + r16) * R;
#pragma unroll
            for (int r = 0; r < R; ++r) pw[r] = (acc[r][0] + acc[r][1]) + (acc[r][2] + acc[r][3]);
        }
        const float bias_cur = bias_v;
        if (l + 1 < a.nlayers && a.L[l + 1].N > 0) fetch_layer(a.L[l + 1]);     // lands during the reduction and the hand-off
        RUN_STAMP(4);
        __syncthreads();
        if (cols && tid < 16 * R) {
            const int row = tid >> 4, col = tid & 15;
            float v = bias_cur;
#pragma unroll
            for (int wk = 0; wk < 4 * R; ++wk) v += part[(wk * 16 + col) * R + row];
            if (L.out) {
                if (live) L.out[(size_t)(row0 + row) * L.ldout + n0 + col] = v;
            } else {
                granule_store(a.gbuf + ((size_t)l * a.Bpad + row0 + row) * RUN_GCOLS + n0 + col, a.epoch0 + (unsigned)l + 1u, v);
            }
        }
        RUN_STAMP(5);
    }
#undef RUN_STAMP
}

// =====================================================================================
// dec_loop: the WHOLE decode loop of a batch in one launch (all steps x all layers), same layer anatomy as dec_run.
// =====================================================================================
// 16-byte row pieces that ANOTHER workgroup wrote earlier in this launch (tap history) or that a side-stream kernel
// wrote while this launch was running (cone rows): moved as two 8-byte agent-scope relaxed atomics (sc1: past the CU's
// L1, coherent across the XCDs' L2s, write-through), never as plain accesses -- a launch-long kernel gets no cache
// maintenance at step boundaries.
// Contraction on v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4 outer products per instruction, exact fp32): the 4 rows
// of the group are the M of every block; block (cg, kk) = lanes 16cg + 4kk .. +3 owns columns 4cg..4cg+3 and k-lane kk
// of each 16-wide chunk:  A[lane] = x[row lane&3][16 ch + 4 kk + e],  B[lane] = Wt[n0 + 4cg + (lane&3)][16 ch + 4 kk + e],
// D[lane][reg i] += A[4 blk + i] * B[lane]  (profiles/mfma4x4_probe.hip).  One ds_read_b128 + 4 MFMAs per chunk instead of
// 4 reads + 16 FMAs: the contraction of a highway layer drops from ~1.0 us to ~0.25 us per wave (profiles/r02 stamps).
// R rows per workgroup = R/4 row quads that share every weight fragment (R = 8: half the weight traffic of R = 4 per
// row, the dominant issue cost of a highway layer -- profiles/r02 ablations).
template <int R, bool DIAG>      // DIAG: phase stamps (OPH_RUN_STAMPS) and the ablation bits of LoopArgs::dbg; the production instance carries neither
__global__ __launch_bounds__(64 * R) void dec_loop(LoopArgs a) {
    const int dbg = DIAG ? a.dbg : 0;
    static_assert(R == 4 || R == 8, "rows per workgroup: one or two quads of the 4x4x1 MFMA");
    constexpr int RQ = R / 4;
    constexpr int PF = (RUN_KMAX / 16 + R - 1) / R;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* part = smem;                          // [R waves][16 columns][RQ quads][4 rows] K-split partial sums
    float* partq = smem + R * 16 * RQ * 4;       // the same for the Q half of the attention layer's contraction (QW)
    float* xs = smem + 2 * R * 16 * RQ * 4;      // [R][ldxs]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.x, n0 = g * 16, row0 = blockIdx.y * R, grow = row0 + w;
    const int c = lane * 4;
    const int mq = lane & 3, mkk = (lane >> 2) & 3, mcol = 4 * (lane >> 4) + mq;     // MFMA roles of this lane
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int NL = a.nlayers, Bpad = a.Bpad;
    LoopDescPtr Ls = (LoopDescPtr)a.L;
    const int spk = a.spk_ids ? a.spk_ids[grow < a.B ? grow : 0] : 0;
    const int my_end = __builtin_amdgcn_readfirstlane(a.ends[grow]);
    // a launch may continue a decode (t_begin > 0: a tile resumed to its batch's stop step, synthesize.py:225-228): the state a
    // step carries over is in memory -- prev_max, t_ends, the mel frame of step t_begin - 1 (Ytm), the histories
    int my_tend = a.t_begin > 0 ? a.t_ends[grow] : a.max_T;      // t_ends[grow]; only column slice 0 records it
    int p = a.t_begin > 0 ? a.p[(a.t_begin & 1) * Bpad + grow] : 0;   // prev_max of this wave's utterance: every workgroup attends for its own rows
    int* const stop_word = a.ctl + 1;
    int* const err = a.ctl + 2;
    if (a.clk && tid == 0) atomicMin((unsigned long long*)a.clk, (unsigned long long)wall_clock64());      // device-side witness: first workgroup in

    f32x4 xprev = zero4;
    f32x4 bfrag[PF], tp0 = zero4, tp1 = zero4;
    bool tpok0 = false, tpok1 = false;
    float bias_v = 0.f;
    // Early taps (kc = 256: every wave's chunk list is [tap0 x PT | tap1 x PT | current x PT]): the two older taps of a highway
    // layer are known before the hand-off, so their rows are staged and their 2/3 of the layer's contraction runs while the
    // sweep of the current row is still in flight -- one more barrier, overlapped with the sweep's round trip (round 3; before:
    // all 0.97 us of MFMAs after sweep, prologue and barrier).  (Measured first: the taps requested straight in the MFMA's
    // operand layout, no staging: 16 scattered requests per lane -- the CU's address unit made the layer 2 us longer.)
    constexpr int PT = 16 / R;
    f32x4 tk[8];                                 // the attention layer's K / V window rows
    bool dt_nxt = false;                         // the layer fetch_layer() was last called for runs its taps early
    auto fetch_layer = [&](const LoopDesc& D, int t) {       // weights, bias and the two older taps of layer D at step t
        const int ntaps = D.ntaps(), kc = D.kc();
        const int nch = (ntaps * kc) >> 4;
        if (n0 < D.N() && !(dbg & 1)) {       // chunks past the layer's K are never multiplied: load a valid address instead of branching
            // pre-swizzled on the host (build_loop_layers): [slice g][wave w][chunk i][lane] -- 1 KB contiguous per request
            const f32x4* wsw = (const f32x4*)D.Wt() + ((size_t)(g * R + w) * PF) * 64 + lane;
#pragma unroll
            for (int i = 0; i < PF; ++i) bfrag[i] = wsw[i * 64];
            bias_v = D.bias()[n0 + (tid & 15)];
        }
        // the two older taps: always requested from a valid row (clamped), masked where they are staged -- no branch, no
        // register copy between the request and its use
        const int kind = D.tapkind();
        dt_nxt = kind != 0 && ntaps == 3 && kc == 256 && !(dbg & 64);
        if (kind != 0 && c < kc && !(dbg & 4)) {
            const int o0 = D.off0(), o1 = D.off1();
            const float* tb = kind == 1 ? (const float*)D.hist() : D.cone(t & 1);
            const int r0 = kind == 1 ? max(t - o0, 0) : D.idx0(), r1 = kind == 1 ? max(t - o1, 0) : D.idx1();
            tp0 = ld_coherent(tb + ((size_t)r0 * Bpad + grow) * kc + c);
            tp1 = ld_coherent(tb + ((size_t)r1 * Bpad + grow) * kc + c);
            tpok0 = t - o0 >= 0; tpok1 = t - o1 >= 0;
        }
    };
    // The taps of an AudioDec highway layer read one LEVEL of this step's cone (side stream).  Each level has its own
    // word, raised by the launch that completes it; a layer waits only for the level it reads, so the cone's later
    // levels overlap the chain's first tap layers.  `seen` is a value requested earlier (one layer ahead).
    auto level_wait = [&](int lv1, int t, unsigned seen, bool have) {
        if (lv1 == 0 || t < 1 || (dbg & 32)) return;
        const unsigned* word = a.sig + LOOP_SIG_LEVEL0 + 16 * (lv1 - 1);
        const unsigned want = a.sig_base + (unsigned)t;
        if (!have) seen = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((int)(seen - want) >= 0) return;
        const bool dbgw = DIAG && a.sigdbg && g == 0 && blockIdx.y == 0 && w == 0 && lane == 0 && lv1 < 8;
        const long long tw0 = dbgw ? wall_clock64() : 0;
        long long t0 = 0;
        for (int it = 0; (int)(seen - want) < 0; ++it) {
            __builtin_amdgcn_s_sleep(2);
            seen = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((it & 63) == 63) {
                const long long now = wall_clock64();
                if (t0 == 0) t0 = now;
                if (now - t0 > RUN_TIMEOUT_TICKS || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                    if (lane == 0) __hip_atomic_store(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        if (dbgw) a.sigdbg[t * 8 + lv1] = wall_clock64() - tw0;
    };
    LoopDesc cur, nxt;
    desc_load(Ls, 0, cur);
    desc_pin(cur);
    fetch_layer(cur, a.t_begin);
    bool dt_cur = dt_nxt;                        // direct taps of the layer being run

    int t = a.t_begin;
    for (; t < a.t_end; ++t) {
        // Early stop (synthesize.py:225-228): the step that sets the flag is >= 1 full step (tens of us) in the past when
        // it is acted on here, so every workgroup takes the same decision; step stop+1 still runs, with its stores off.
        if (__hip_atomic_load(stop_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= t - 2) break;
        long long* const stp = (DIAG && a.stamps && t == a.stamp_t && w == 0 && blockIdx.y == 0) ? a.stamps + (size_t)g * LOOP_MAX_LAYERS * 8 : nullptr;
#define LOOP_STAMP(K) do { if (stp && lane == 0) stp[l * 8 + (K)] = wall_clock64(); } while (0)
        const long long step_w0 = stp ? wall_clock64() : 0, step_c0 = stp ? clock64() : 0;
        for (int l = 0; l < NL; ++l) {
            const bool first = l == 0, no_input = first && t == a.t_begin;      // S[0] = 0 (architectures.py:191); a resumed launch reads S[t_begin] from Ytm
            desc_load(Ls, l + 1 < NL ? l + 1 : 0, nxt);      // in flight across the hand-off wait below
            const int pre = cur.pre(), cin = cur.cin(), nonorm = cur.nonorm(), kc = cur.kc(), ntaps = cur.ntaps();
            const bool cok = c < cin, two = pre >= RUN_HC, cols = n0 < cur.N();
            const bool is_attn = l == a.attn_layer;

            // A column slice beyond this layer's width (k=1 layers are 256 or n_mels wide, highway layers 512) has
            // nothing to contract here.  It needs the layer's input only as the highway residual of the next prologue --
            // which the consumer of a k=1 layer never uses -- so it sits the layer out: fewer pollers on the hand-off.
            if (!cols && cur.next_pre() < RUN_HC && !(dbg & 16)) {
                level_wait(cur.next_level(), t, 0u, false);
                desc_pin(nxt);
                fetch_layer(nxt, l + 1 < NL ? t : t + 1);
                cur = nxt; dt_cur = dt_nxt;
                continue;
            }

            // ---- 1. requests that do not depend on the hand-off
            f32x4 g1v = zero4, b1v = zero4, g2v = zero4, b2v = zero4;
            if (pre != RUN_COPY && cok) {
                const float* lnp = cur.lnp() + c;
                const int ls = cur.ls();
                g1v = *(const f32x4*)lnp; b1v = *(const f32x4*)(lnp + ls);
                if (two) { g2v = *(const f32x4*)(lnp + 2 * ls); b2v = *(const f32x4*)(lnp + 3 * ls); }
            }
            int stop_v = 0x7fffffff;
            if (g == 0) stop_v = __hip_atomic_load(stop_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int nlv = cur.next_level();
            unsigned sigl = 0;
            if (nlv && t >= 1) sigl = __hip_atomic_load(a.sig + LOOP_SIG_LEVEL0 + 16 * (nlv - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            // attention window [p, p+win): its K and V rows depend on p alone -- requested before the hand-off, not inside
            // the softmax loops (each was an exposed ~1.5 us round trip: profiles/r02 stamps, 10 us per step)
            constexpr int AW = 4;
#define kwin(i) tk[(i)]
#define vwin(i) tk[AW + (i)]
            if (is_attn) {
                const float* KVb = a.KV + (size_t)grow * a.N_keys * 2 * cin;
#pragma unroll
                for (int i = 0; i < AW; ++i) {
                    const bool in = cok && i < a.win && p + i < a.N_keys;
                    kwin(i) = in ? *(const f32x4*)(KVb + (size_t)(p + i) * 2 * cin + c) : zero4;
                    vwin(i) = in ? *(const f32x4*)(KVb + cin + (size_t)(p + i) * 2 * cin + c) : zero4;
                }
            }

            // ---- 2. this wave's raw row of the producing layer (layer 0: the last layer of the previous step)
            LOOP_STAMP(0);
            int passes = 0;
            f32x4 av = zero4, uv = zero4;
            f32x4 acc[RQ][2], accq[RQ];
#pragma unroll
            for (int rq = 0; rq < RQ; ++rq) { acc[rq][0] = zero4; acc[rq][1] = zero4; accq[rq] = zero4; }
            {
                const int slot = first ? NL - 1 : l - 1;
                const unsigned ep = a.epoch0 + (unsigned)((first ? t - 1 : t) * LOOP_MAX_LAYERS + slot + 1);
                const u64* grow_p = a.gbuf + ((size_t)slot * Bpad + grow) * RUN_GCOLS;
                // first pass of the hand-off sweep: requested now, looked at after the older taps' share of the contraction
                const u64 want = (u64)ep << 32;
                u64 ga[4], gu[4];
                if (!no_input) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) ga[e] = cok ? granule_load(grow_p + c + e) : want;
#pragma unroll
                    for (int e = 0; e < 4; ++e) gu[e] = (cok && two) ? granule_load(grow_p + cin + c + e) : want;
                }
                __builtin_amdgcn_sched_barrier(0);
                if (dt_cur) {
                    // taps x[t-2r], x[t-r]: staged now (their rows arrived during the previous layer), chunks i < 2 PT of every
                    // wave's list contracted before the sweep's answer is looked at
                    const int ldxs_ = 3 * 256 + 16;
                    float* xrow_ = xs + w * ldxs_;
                    *(f32x4*)(xrow_ + c) = tpok0 ? tp0 : zero4;
                    *(f32x4*)(xrow_ + 256 + c) = tpok1 ? tp1 : zero4;
                    __syncthreads();
                    if (cols) {
                        const float* xa_ = xs + mq * ldxs_ + mkk * 4;
                        f32x4 xt[2 * PT][RQ];
#pragma unroll
                        for (int i = 0; i < 2 * PT; ++i)
#pragma unroll
                            for (int rq = 0; rq < RQ; ++rq) xt[i][rq] = *(const f32x4*)(xa_ + rq * 4 * ldxs_ + (w + R * i) * 16);
#pragma unroll
                        for (int i = 0; i < 2 * PT; ++i)
#pragma unroll
                            for (int rq = 0; rq < RQ; ++rq) {
                                acc[rq][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(xt[i][rq][0], bfrag[i][0], acc[rq][0], 0, 0, 0);
                                acc[rq][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(xt[i][rq][1], bfrag[i][1], acc[rq][1], 0, 0, 0);
                                acc[rq][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(xt[i][rq][2], bfrag[i][2], acc[rq][0], 0, 0, 0);
                                acc[rq][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(xt[i][rq][3], bfrag[i][3], acc[rq][1], 0, 0, 0);
                            }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!no_input) {
                    bool ok = true;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ok = ok && (unsigned)(ga[e] >> 32) == ep && (unsigned)(gu[e] >> 32) == ep;
                    passes = 1;
                    if (__all(ok) || (dbg & 2)) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { av[e] = __uint_as_float((unsigned)ga[e]); uv[e] = __uint_as_float((unsigned)gu[e]); }
                    } else {
                        passes += sweep_row(grow_p, c, cok, cin + c, two, ep, lane, err, av, uv, false);
                    }
                }
            }
            LOOP_STAMP(1);
            if (stp && lane == 0) stp[l * 8 + 6] = passes;
            desc_pin(nxt);

            // ---- 3. prologue math (one row per wave)
            f32x4 x = av;
            if (pre != RUN_COPY && !(dbg & 8)) {
                const float invc = __builtin_amdgcn_rcpf((float)cin);
                float s1 = av[0] + av[1] + av[2] + av[3], s2 = uv[0] + uv[1] + uv[2] + uv[3];
                s1 = wave_sum(s1);
                if (two) s2 = wave_sum(s2);
                const float m1 = nonorm ? 0.f : s1 * invc, m2 = nonorm ? 0.f : s2 * invc;
                float q1 = 0.f, q2 = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d1 = cok ? av[e] - m1 : 0.f, d2 = cok ? uv[e] - m2 : 0.f;
                    av[e] = d1; uv[e] = d2;
                    q1 += d1 * d1; q2 += d2 * d2;
                }
                q1 = wave_sum(q1);
                if (two) q2 = wave_sum(q2);
                const float r1 = nonorm ? 1.0f : fast_rsqrt(q1 * invc + LN_EPS);
                const float r2 = nonorm ? 1.0f : fast_rsqrt(q2 * invc + LN_EPS);
                if (two) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float h1 = av[e] * r1 * g1v[e] + b1v[e], h2 = uv[e] * r2 * g2v[e] + b2v[e];
                        const float gte = fast_sigmoid(h1);
                        const float y = gte * h2 + (1.0f - gte) * xprev[e];
                        x[e] = cok ? y : 0.f;
                    }
                } else {
                    const int act = cur.act();
                    f32x4 y;
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = av[e] * r1 * g1v[e] + b1v[e];
                    if (act == ACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], 0.f);
                    } else if (act == ACT_SIGMOID) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] = fast_sigmoid(y[e]);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = cok ? y[e] : 0.f;
                }
            }
            if (no_input) x = (t > 0 && c < a.ldy) ? *(const f32x4*)(a.Ytm + ((size_t)t * Bpad + grow) * a.ldy + c) : zero4;
            xprev = x;
            if (first && t >= 1) {
                // x is mel frame t-1 -> Y[b][t-1] and the decoder input S[t] (synthesize.py:204-209)
                if (g == 0 && grow < a.B && t - 1 <= stop_v && c < a.ldy) {
                    // written through: SSRN chunks on another stream read the frames while this launch is still running
                    st_coherent(a.Yout + ((size_t)grow * a.max_T + (t - 1)) * a.ldy + c, x);
                    *(f32x4*)(a.Ytm + ((size_t)t * Bpad + grow) * a.ldy + c) = x;
                }
            }
            // ---- 4. stage the operand row: [tap x[t-2r] | tap x[t-r] | current]
            const int Ktot = ntaps * kc, ldxs = Ktot + 16, xcur = (ntaps - 1) * kc;     // +16: the 4 rows' b128 reads hit disjoint banks
            float* xrow = xs + w * ldxs;
            const bool live = t <= stop_v;
            if (pre == RUN_ATTN) {
                // R' = concat(softmax(Q K^T / sqrt(d)) V, Q) for row t under the current mask (networks.py:300-319); see dec_run
                const int d = cin;
                const float* KVb = a.KV + (size_t)grow * a.N_keys * 2 * d;
                const int nwin = min(a.win, a.N_keys - p);
                const float scale = fast_rsqrt((float)d);
                float scl = -INFINITY;
#pragma unroll
                for (int i = 0; i < AW; ++i) {
                    if (i < nwin) {
                        const float sdot = wave_sum(x[0] * kwin(i)[0] + x[1] * kwin(i)[1] + x[2] * kwin(i)[2] + x[3] * kwin(i)[3]) * scale;
                        if (lane == i) scl = sdot;
                    }
                }
                for (int i = AW; i < nwin; ++i) {           // windows wider than the prefetched ones
                    const f32x4 kv = cok ? *(const f32x4*)(KVb + (size_t)(p + i) * 2 * d + c) : zero4;
                    const float sdot = wave_sum(x[0] * kv[0] + x[1] * kv[1] + x[2] * kv[2] + x[3] * kv[3]) * scale;
                    if (lane == i) scl = sdot;
                }
                float mx = -INFINITY;
                for (int i = 0; i < nwin; ++i) mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, scl), i)));
                float prl = lane < nwin ? __builtin_amdgcn_exp2f(1.4426950408889634f * (scl - mx)) : 0.f;
                float den = 0.f;
                for (int i = 0; i < ATT_WMAX; ++i) den += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, prl), i));
                prl = prl * __builtin_amdgcn_rcpf(den);
                int arg = 0;
                float best = -1.f;
                f32x4 ctx = zero4;
#pragma unroll
                for (int i = 0; i < AW; ++i) {
                    if (i < nwin) {
                        const float pi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, prl), i));
                        if (pi > best) { best = pi; arg = i; }       // first maximum, like tf.argmax
#pragma unroll
                        for (int e = 0; e < 4; ++e) ctx[e] += pi * vwin(i)[e];
                    }
                }
                for (int i = AW; i < nwin; ++i) {
                    const float pi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, prl), i));
                    if (pi > best) { best = pi; arg = i; }
                    const f32x4 vv = cok ? *(const f32x4*)(KVb + d + (size_t)(p + i) * 2 * d + c) : zero4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ctx[e] += pi * vv[e];
                }
                if (cok) {
                    *(f32x4*)(xrow + c) = ctx;
                    *(f32x4*)(xrow + d + c) = x;
                }
                for (int c2 = 2 * d + lane; c2 < kc; c2 += 64) xrow[c2] = 0.f;
                const int m = p + arg;
                if (g == 0 && live && grow < a.B) {
                    if (cok) st_coherent(a.Qhist + ((size_t)t * Bpad + grow) * d + c, x);          // read by the cone kernels
                    if (lane < nwin) a.align[(size_t)grow * a.N_keys * a.max_T + (size_t)(p + lane) * a.max_T + t] = prl;
                    if (lane == 0) {
                        __hip_atomic_store(a.p + ((t + 1) & 1) * Bpad + grow, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (my_tend == a.max_T && m >= my_end) {       // synthesize.py:218-228
                            my_tend = t;
                            a.t_ends[grow] = t;
                            const int old = atomicAdd(a.ctl, 1);
                            if (old + 1 == a.B && a.stop_mode == 0) {
                                __hip_atomic_store(stop_word, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                __hip_atomic_store((int*)a.host_progress + 1, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            }
                        }
                    }
                } else if (my_tend == a.max_T && m >= my_end) my_tend = t;
                p = m;
            } else {
                if (c < kc) *(f32x4*)(xrow + xcur + c) = x;
                for (int c2 = 256 + c; c2 < kc; c2 += 256) *(f32x4*)(xrow + xcur + c2) = zero4;
                const int ccat = cur.ccat();
                if (ccat > 0) {
                    const float* tab = cur.cat_table();
                    for (int j = lane; j < ccat; j += 64) xrow[xcur + cin + j] = spk == 0 ? 0.f : tab[(size_t)spk * ccat + j];
                }
                if (ntaps == 3 && c < kc && !dt_cur) {
                    *(f32x4*)(xrow + c) = tpok0 ? tp0 : zero4;
                    *(f32x4*)(xrow + kc + c) = tpok1 ? tp1 : zero4;
                }
                if (cur.tapkind() == 1 && g == 0 && live && c < kc) st_coherent(cur.hist() + ((size_t)t * Bpad + grow) * kc + c, x);
            }
            LOOP_STAMP(2);
            __syncthreads();
            LOOP_STAMP(3);

            // ---- 5. R x 16 slice on the 4x4x1 MFMA, K split round-robin over the R waves; every LDS read of the layer is
            //         issued before the first MFMA, two accumulators per row quad
            if (cols) {
                const float* xa = xs + mq * ldxs + mkk * 4;
                const int nch = Ktot >> 4;
                // attention layer: its operand is [context | Q]; the chunks of the Q half also go to their own accumulator
                const bool want_qw = is_attn && a.QW != nullptr;
                const int qch0 = want_qw ? (cin >> 4) : 0x7fffffff;
                f32x4 xf[PF][RQ];
                const int i_first = dt_cur ? 2 * PT : 0;          // direct taps: chunks [0, 2 PT) are done
#pragma unroll
                for (int i = 0; i < PF; ++i) {
                    const int ch = min(w + R * i, nch - 1);
                    if (i >= i_first) {
#pragma unroll
                        for (int rq = 0; rq < RQ; ++rq) xf[i][rq] = *(const f32x4*)(xa + rq * 4 * ldxs + ch * 16);
                    }
                }
#pragma unroll
                for (int i = 0; i < PF; ++i) {
                    if (i >= i_first && w + R * i < nch) {
                        if (w + R * i >= qch0) {
#pragma unroll
                            for (int rq = 0; rq < RQ; ++rq)
#pragma unroll
                                for (int e = 0; e < 4; ++e) accq[rq] = __builtin_amdgcn_mfma_f32_4x4x1f32(xf[i][rq][e], bfrag[i][e], accq[rq], 0, 0, 0);
                        } else {
#pragma unroll
                            for (int rq = 0; rq < RQ; ++rq) {
                                acc[rq][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(xf[i][rq][0], bfrag[i][0], acc[rq][0], 0, 0, 0);
                                acc[rq][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(xf[i][rq][1], bfrag[i][1], acc[rq][1], 0, 0, 0);
                                acc[rq][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(xf[i][rq][2], bfrag[i][2], acc[rq][0], 0, 0, 0);
                                acc[rq][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(xf[i][rq][3], bfrag[i][3], acc[rq][1], 0, 0, 0);
                            }
                        }
                    }
                }
                // the 4 k-lanes of a column sit 4 lanes apart in one 16-lane row: two DPP row rotations sum them in
                // registers, and k-lane 0 alone writes (the reducer then reads R values per output, not 4 R)
#pragma unroll
                for (int rq = 0; rq < RQ; ++rq) {
                    f32x4 v = (acc[rq][0] + acc[rq][1]) + accq[rq];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float y = v[e];
                        y += dpp_mov<0x124>(y);      // row_ror:4
                        y += dpp_mov<0x128>(y);      // row_ror:8
                        v[e] = y;
                    }
                    if (mkk == 0) *(f32x4*)(part + ((w * 16 + mcol) * RQ + rq) * 4) = v;      // [row i] of (quad rq, column mcol)
                    if (want_qw) {
                        f32x4 q = accq[rq];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float y = q[e];
                            y += dpp_mov<0x124>(y);
                            y += dpp_mov<0x128>(y);
                            q[e] = y;
                        }
                        if (mkk == 0) *(f32x4*)(partq + ((w * 16 + mcol) * RQ + rq) * 4) = q;
                    }
                }
            }
            LOOP_STAMP(7);
            const float bias_cur = bias_v;
            level_wait(nlv, t, sigl, true);
            fetch_layer(nxt, l + 1 < NL ? t : t + 1);       // (after the last step: layer 0 of a step that never runs -- valid rows, unused)
            LOOP_STAMP(4);
            __syncthreads();
            if (cols && tid < 16 * R) {
                const int row = tid >> 4, col = tid & 15;
                const float* pr = part + (col * RQ + (row >> 2)) * 4 + (row & 3);
                float pv[R];
#pragma unroll
                for (int ww = 0; ww < R; ++ww) pv[ww] = pr[ww * 16 * RQ * 4];
                float v = bias_cur;
#pragma unroll
                for (int ww = 0; ww < R; ww += 2) v += pv[ww] + pv[ww + 1];
                granule_store(a.gbuf + ((size_t)l * Bpad + row0 + row) * RUN_GCOLS + n0 + col,
                              a.epoch0 + (unsigned)(t * LOOP_MAX_LAYERS + l + 1), v);
                if (is_attn && a.QW != nullptr) {
                    // QW[t] = Q[t] . Wq + bias for the cone head's cache (written through: the cone kernels read it after their acquire)
                    const float* pq = partq + (col * RQ + (row >> 2)) * 4 + (row & 3);
                    float vq = bias_cur;
#pragma unroll
                    for (int ww = 0; ww < R; ww += 2) vq += pq[ww * 16 * RQ * 4] + pq[(ww + 1) * 16 * RQ * 4];
                    if (row0 + row < a.B && n0 + col < cin)
                        __hip_atomic_store(a.QW + ((size_t)t * Bpad + row0 + row) * cin + n0 + col, vq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            LOOP_STAMP(5);
            if (is_attn && (a.QW != nullptr ? cols : g == 0)) {
                // release the cone of step t+1 on the side stream: Q[t] and prev_max are written through; once every
                // row group has arrived, one lane raises the word the stream's wait-value operation polls
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) {
                    const int old = atomicAdd(a.ctl + 3, 1);
                    if (old + 1 == (Bpad / R) * (a.QW != nullptr ? a.attn_slices : 1) * (t + 1 - a.t_begin)) {
                        __hip_atomic_fetch_max(a.sig, a.sig_base + (unsigned)t + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        __hip_atomic_store((int*)a.host_progress, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        if (DIAG && a.sigdbg && t + 1 < a.max_T) a.sigdbg[(t + 1) * 8 + 0] = wall_clock64();
                    }
                }
            }
            cur = nxt; dt_cur = dt_nxt;
        }
        if (stp && lane == 0) {     // shader clock over this step: (c1 - c0) cycles in (w1 - w0) * 10 ns
            long long* q = stp + (LOOP_MAX_LAYERS - 1) * 8;
            q[0] = step_w0; q[1] = wall_clock64(); q[2] = step_c0; q[3] = clock64();
        }
#undef LOOP_STAMP
    }
    // ---- the last executed step's mel frame (its consumer, layer 0 of the next step, does not run)
    const int t_last = t - 1;
    if (g == 0 && t_last >= 0) {
        const int stop_v = __hip_atomic_load(stop_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t_last <= stop_v) {
            desc_load(Ls, 0, cur);
            const int cin = cur.cin();
            const bool cok = c < cin;
            f32x4 g1v = zero4, b1v = zero4, av = zero4, uv = zero4;
            if (cok) { g1v = *(const f32x4*)(cur.lnp() + c); b1v = *(const f32x4*)(cur.lnp() + cur.ls() + c); }
            sweep_row(a.gbuf + ((size_t)(NL - 1) * Bpad + grow) * RUN_GCOLS, c, cok, cin + c, false,
                      a.epoch0 + (unsigned)(t_last * LOOP_MAX_LAYERS + NL), lane, err, av, uv);
            const float invc = __builtin_amdgcn_rcpf((float)cin);
            const float m1 = cur.nonorm() ? 0.f : wave_sum(av[0] + av[1] + av[2] + av[3]) * invc;
            float q1 = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d1 = cok ? av[e] - m1 : 0.f; av[e] = d1; q1 += d1 * d1; }
            const float r1 = cur.nonorm() ? 1.0f : fast_rsqrt(wave_sum(q1) * invc + LN_EPS);
            f32x4 x;
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = cok ? fast_sigmoid(av[e] * r1 * g1v[e] + b1v[e]) : 0.f;
            if (grow < a.B && c < a.ldy) {
                *(f32x4*)(a.Yout + ((size_t)grow * a.max_T + t_last) * a.ldy + c) = x;
                *(f32x4*)(a.Ytm + ((size_t)(t_last + 1) * Bpad + grow) * a.ldy + c) = x;
            }
        }
    }
    // whatever happened, the side stream's remaining wait-value operations must not wait for steps that never ran
    if (g == 0 && blockIdx.y == 0 && tid == 0)
        __hip_atomic_fetch_max(a.sig, a.sig_base + (unsigned)a.max_T + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (a.clk && tid == 0) atomicMax((unsigned long long*)a.clk + 1, (unsigned long long)wall_clock64());      // device-side witness: last workgroup out
}

template <int R, bool DIAG>
static void launch_dec_loop_t(const LoopArgs& a, int col_slices, int kmax, hipStream_t s) {
    static thread_local std::map<int, size_t> done;
    const size_t lds_bytes = (size_t)(2 * R * 16 * (R / 4) * 4 + R * (kmax + 16)) * 4;
    int dev = 0;
    (void)hipGetDevice(&dev);
    size_t& d = done[dev];
    if (d < lds_bytes) { (void)hipFuncSetAttribute((const void*)dec_loop<R, DIAG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); d = lds_bytes; }
    hipLaunchKernelGGL((dec_loop<R, DIAG>), dim3(col_slices, a.Bpad / R), dim3(64 * R), lds_bytes, s, a);
}
void launch_dec_loop(const LoopArgs& a, int col_slices, int rows_per_group, int kmax, hipStream_t s) {
    const bool diag = a.dbg != 0 || a.stamps != nullptr || a.sigdbg != nullptr;
    if (rows_per_group == 4) { if (diag) launch_dec_loop_t<4, true>(a, col_slices, kmax, s); else launch_dec_loop_t<4, false>(a, col_slices, kmax, s); }
    else { if (diag) launch_dec_loop_t<8, true>(a, col_slices, kmax, s); else launch_dec_loop_t<8, false>(a, col_slices, kmax, s); }
}
// workgroups of dec_loop that fit on one CU at once (the loop kernel needs ALL of its workgroups resident)
template <int R>
static int blocks_per_cu_t(int kmax) {
    const size_t lds_bytes = (size_t)(2 * R * 16 * (R / 4) * 4 + R * (kmax + 16)) * 4;
    (void)hipFuncSetAttribute((const void*)dec_loop<R, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)dec_loop<R, true>, 64 * R, lds_bytes) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}
int dec_loop_blocks_per_cu(int rows_per_group, int kmax) { return rows_per_group == 4 ? blocks_per_cu_t<4>(kmax) : blocks_per_cu_t<8>(kmax); }

template <int R>
static void launch_dec_run_t(const RunArgs& a, int col_slices, int kmax, hipStream_t s) {
    // per (device); thread_local: a handle is confined to one host thread, different threads may drive different devices
    static thread_local std::map<int, size_t> done;
    const size_t lds_bytes = (size_t)(R * 4 * 16 * R + R * (kmax + 4)) * 4;
    int dev = 0;
    (void)hipGetDevice(&dev);
    size_t& d = done[dev];
    if (d < lds_bytes) { (void)hipFuncSetAttribute((const void*)dec_run<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); d = lds_bytes; }
    hipLaunchKernelGGL(dec_run<R>, dim3(col_slices, a.Bpad / R), dim3(64 * R), lds_bytes, s, a);
}
void launch_dec_run(const RunArgs& a, int col_slices, int rows_per_group, int kmax, hipStream_t s) {
    if (rows_per_group == 8) launch_dec_run_t<8>(a, col_slices, kmax, s);
    else launch_dec_run_t<4>(a, col_slices, kmax, s);
}

}  // namespace oph
