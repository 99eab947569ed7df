// dec_chain: the whole decode loop of a 16-utterance tile in ONE launch (all steps x all AudioEnc / Attention / AudioDec layers,
// synthesize.py:181-209 over networks.py:214-435) -- the lean successor of dec_loop (oph_decrun.hip) for the standard geometry.
//
// Same protocol as dec_loop (oph_internal.h: LoopArgs, packed layer descriptors, cone level words, stop word, progress words;
// the layer hand-offs travel tag-free since round 5, see CH_SENT below), same arithmetic in the same order (bitwise the same results), different code:
// round 4's stamps showed that a layer of dec_loop is bound by its own instruction stream, not by the hand-off -- every first
// sweep pass found its granules already there, and a wave executed ~1500 instructions per layer (one wave per SIMD, in order),
// a third of them scalar-register spills (v_readlane / v_writelane), exec-masked branches around single loads and descriptor
// bit-field decoding for a geometry that is the same in 22 of 24 layers.  Here
//   * the geometry is fixed at compile time: d = 256 channels per row (lane l <-> channels 4l..4l+3, no channel predicate), operand
//     rows [tap x[t-2r] | tap x[t-r] | current] at a constant LDS stride, R = 8 utterance rows x 16 columns per workgroup, K split
//     over the 8 waves; only the mel layer (n_mels <= 256 channels) is masked;
//   * every layer is one of five instantiations of one body (prologue CONV / HC / ATTENTION / MEL x contraction K1 / HC3), picked by
//     a scalar switch, so no instantiation carries another's branches;
//   * all loads are unconditional (clamped addresses, masked where they are used), the next layer's weights, taps and parameters
//     are requested right AFTER the publish, in the shadow of the hand-off's own round trip, instead of in front of it.
// Configurations outside this geometry (hp.norm None, LCC, non-monotonic / fixed attention, d != 256) keep dec_loop.
#include "oph_internal.h"
#include "oph_device.h"
#include "oph_loopdev.h"

#include <map>
#include <type_traits>

namespace oph {

namespace {
constexpr int CH_R = 8, CH_RQ = 2, CH_PF = 6, CH_PT = 2;      // rows per workgroup, row quads, 16-wide k chunks per wave, chunks per tap
constexpr int CH_D = 256;                                   // channels per row
constexpr int CH_LDXS = 3 * CH_D + 16;                      // operand row stride in LDS (+16: the 4 rows' b128 reads hit disjoint banks)
constexpr int CH_AW = 4;                                    // attention window rows held in registers
// Hand-off transport (round 5): the raw outputs of a layer travel as plain 4-byte values -- no {epoch, value} granules.  A slot
// [step parity][layer][row][512] holds either the values of that (step, layer) or CH_SENT in every word: the producer of (t, l)
// resets its columns of the OTHER parity's slot (the values of step t-1, consumed a whole step ago by every reader: each of them
// has since published something this workgroup gathered) at the top of the layer, waits for its own sweep (s_waitcnt vmcnt(0):
// the reset is acknowledged) and only then publishes (t, l); the slot is rewritten a full step later.  A reader polls until none
// of its words is the sentinel.  A word is never torn, so 16-byte sc1 loads and stores carry four values each: a wave's row is
// 2 requests instead of 8, a workgroup's publish 32 stores instead of 128, half the bytes -- profiles/handoff2_probe.hip: 1.55 us
// per all-to-all against 2.06 us for the granules.  CH_SENT is a NaN pattern no arithmetic produces (a result with exactly
// these bits is published as the canonical NaN).
constexpr unsigned CH_SENT = 0xFFFFFFFFu;
enum { P_CONV = 0, P_HC = 1, P_ATTN = 2, P_MEL = 3 };       // prologue kinds
enum { C_K1 = 0, C_HC3 = 1 };                               // contraction kinds
template <int V> using ic = std::integral_constant<int, V>;
}  // namespace

template <bool STAMPS>
__global__ __launch_bounds__(64 * CH_R) void dec_chain(LoopArgs a) {
    constexpr int R = CH_R, RQ = CH_RQ, PF = CH_PF, PT = CH_PT, LDXS = CH_LDXS, AW = CH_AW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const part = smem;                          // [R waves][16 columns][RQ quads][4 rows] K-split partial sums
    float* const partq = smem + R * 16 * RQ * 4;       // the same for the Q half of the attention layer's contraction (QW)
    float* const xs = smem + 2 * R * 16 * RQ * 4;      // [R][LDXS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.x, n0 = g * 16, row0 = blockIdx.y * R, grow = row0 + w;
    const int c = lane * 4;
    const int mq = lane & 3, mkk = (lane >> 2) & 3, mcol = 4 * (lane >> 4) + mq;     // MFMA roles of this lane (profiles/mfma4x4_probe.hip)
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int NL = a.nlayers, Bpad = a.Bpad;
    LoopDescPtr Ls = (LoopDescPtr)a.L;
    const int spk = a.spk_ids ? a.spk_ids[grow < a.B ? grow : 0] : 0;
    const int my_end = __builtin_amdgcn_readfirstlane(a.ends[grow]);
    int my_tend = a.t_begin > 0 ? a.t_ends[grow] : a.max_T;
    int p = a.t_begin > 0 ? a.p[(a.t_begin & 1) * Bpad + grow] : 0;       // prev_max of this wave's utterance
    int* const stop_word = a.ctl + 1;
    int* const err = a.ctl + 2;
    // the hand-off buffer [2 step parities][LOOP_MAX_LAYERS][Bpad][RUN_GCOLS] floats through ONE buffer resource: scalar offset =
    // slot and row, lane offset = column (16-byte sc1 requests)
    const __amdgpu_buffer_rsrc_t vres = __builtin_amdgcn_make_buffer_rsrc((void*)a.vbuf, 0, 0x7fffffff, 0x00020000);
    auto slot_off = [&](int step, int layer, int row) -> int {      // byte offset of a row of slot (step & 1, layer)
        return ((((step & 1) * LOOP_MAX_LAYERS + layer) * Bpad + row) * RUN_GCOLS) * 4;
    };
    auto vld = [&](int soff, int voff) -> f32x4 { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(vres, voff, soff, 16 /* sc1 */)); };
    auto vst = [&](int soff, int voff, const f32x4& v) { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_, v), vres, voff, soff, 16 /* sc1 */); };
    // one wave re-reads its row until no word is the sentinel; bounded like every spin of the launch (error word, results undefined then)
    auto sweep_vals = [&](int soff, int v1, bool cok, int v2, bool two, f32x4& av, f32x4& uv) -> int {
        long long t0 = 0;
        for (int it = 0;; ++it) {
            const f32x4 ga = vld(soff, v1);
            f32x4 gu = ga;
            if (two) gu = vld(soff, v2);
            bool ok = true;
#pragma unroll
            for (int e = 0; e < 4; ++e) ok = ok && (__float_as_uint(ga[e]) != CH_SENT || !cok) && (__float_as_uint(gu[e]) != CH_SENT || !two);
            bool give_up = false;
            if (!__all(ok) && it >= 64 && (it & 63) == 0) {
                const long long now = wall_clock64();
                if (t0 == 0) t0 = now;
                give_up = now - t0 > RUN_TIMEOUT_TICKS || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                if (give_up && lane == 0) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (__all(ok) || give_up) { av = ga; uv = gu; return it + 1; }
            __builtin_amdgcn_s_sleep(1);
        }
    };
    if (a.clk && tid == 0) atomicMin((unsigned long long*)a.clk, (unsigned long long)wall_clock64());      // device-side witness: first workgroup in

    f32x4 xprev = zero4;
    f32x4 bfrag[PF], tp0 = zero4, tp1 = zero4;
    bool tpok0 = false, tpok1 = false;
    float bias_v = 0.f;
    float* const xrow = xs + w * LDXS;
    const float* const xa = xs + mq * LDXS + mkk * 4;

    // weights, bias and the two older taps of layer D at step t (immutable data: plain loads; taps: rows another workgroup or a
    // side-stream kernel wrote during this launch: coherent loads).  Chunks past the layer's K hold zeros in the swizzled copy.
    auto fetch_layer = [&](const LoopDesc& D, int t) {
        if (n0 < D.N()) {
            const f32x4* wsw = (const f32x4*)D.Wt() + ((size_t)(g * R + w) * PF) * 64 + lane;
            const int nch = (D.ntaps() * D.kc()) >> 4;
            bfrag[0] = wsw[0];
            bfrag[1] = wsw[64];
            if (nch > 2 * R) { bfrag[2] = wsw[128]; bfrag[3] = wsw[192]; }
            if (nch > 4 * R) { bfrag[4] = wsw[256]; bfrag[5] = wsw[320]; }
            bias_v = D.bias()[n0 + (tid & 15)];
        }
        const int kind = D.tapkind();
        if (kind != 0) {
            const int o0 = D.off0(), o1 = D.off1();
            const float* tb = kind == 1 ? (const float*)D.hist() : D.cone(t & 1);
            const int r0 = kind == 1 ? max(t - o0, 0) : D.idx0(), r1 = kind == 1 ? max(t - o1, 0) : D.idx1();
            tp0 = ld_coherent(tb + ((size_t)r0 * Bpad + grow) * CH_D + c);
            tp1 = ld_coherent(tb + ((size_t)r1 * Bpad + grow) * CH_D + c);
            tpok0 = t - o0 >= 0; tpok1 = t - o1 >= 0;
        }
    };
    // The taps of an AudioDec highway layer read one LEVEL of this step's cone (side stream); each level has its own word.
    auto level_wait = [&](int lv1, int t, unsigned seen) {
        if (lv1 == 0 || t < 1 || (a.dbg & 32)) return;
        const unsigned* word = a.sig + LOOP_SIG_LEVEL0 + 16 * (lv1 - 1);
        const unsigned want = a.sig_base + (unsigned)t;
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
    };
    auto level_word = [&](int lv1, int t) -> unsigned {
        if (lv1 == 0 || t < 1) return 0u;
        return __hip_atomic_load(a.sig + LOOP_SIG_LEVEL0 + 16 * (lv1 - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };

    LoopDesc cur, nxt;
    desc_load(Ls, 0, cur);
    desc_pin(cur);
    fetch_layer(cur, a.t_begin);

    int t = a.t_begin;
    for (; t < a.t_end; ++t) {
        // Early stop (synthesize.py:225-228): the step that sets the flag is >= 1 full step in the past when it is acted on here,
        // so every workgroup takes the same decision; step stop+1 still runs, with its stores off.
        if (__hip_atomic_load(stop_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= t - 2) break;
        long long* const stp = (STAMPS && a.stamps && t == a.stamp_t && w == 0 && blockIdx.y == 0) ? a.stamps + (size_t)g * LOOP_MAX_LAYERS * 8 : nullptr;
        const long long step_w0 = stp ? wall_clock64() : 0, step_c0 = stp ? clock64() : 0;
        for (int l = 0; l < NL; ++l) {
#define CH_STAMP(K) do { if (STAMPS && stp && lane == 0) stp[l * 8 + (K)] = wall_clock64(); } while (0)
            const bool cols = n0 < cur.N();
            const int pre = cur.pre();
            // A column slice beyond this layer's width has nothing to contract here and nobody needs its copy of the layer's input
            // (the consumer of a k = 1 layer never uses the highway residual): it sits the layer out.
            if (!cols && cur.next_pre() < RUN_HC) {
                const int nlv = cur.next_level();
                desc_load(Ls, l + 1 < NL ? l + 1 : 0, nxt);
                level_wait(nlv, t, level_word(nlv, t));
                desc_pin(nxt);
                fetch_layer(nxt, l + 1 < NL ? t : t + 1);
                cur = nxt;
                continue;
            }
            auto layer = [&](auto pro_t, auto con_t) {
                constexpr int PRO = decltype(pro_t)::value, CON = decltype(con_t)::value;
                constexpr bool two = PRO == P_HC || PRO == P_ATTN;
                const bool no_input = PRO == P_MEL && t == a.t_begin;      // S[0] = 0 (architectures.py:191); a resumed launch reads S[t_begin] from Ytm
                desc_load(Ls, l + 1 < NL ? l + 1 : 0, nxt);                // in flight across the hand-off
                // ---- 1. requests that do not depend on the hand-off
                const int cin = PRO == P_MEL ? cur.cin() : CH_D;
                const bool cok = PRO == P_MEL ? c < cin : true;
                const int ci = cok ? c : 0;
                const float* lnp = cur.lnp() + ci;
                const int ls = cur.ls();
                const f32x4 g1v = *(const f32x4*)lnp, b1v = *(const f32x4*)(lnp + ls);
                f32x4 g2v = zero4, b2v = zero4;
                if (two) { g2v = *(const f32x4*)(lnp + 2 * ls); b2v = *(const f32x4*)(lnp + 3 * ls); }
                int stop_v = 0x7fffffff;
                if ((PRO == P_MEL || PRO == P_ATTN || CON == C_HC3) && g == 0) stop_v = __hip_atomic_load(stop_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int nlv = cur.next_level();
                const unsigned sigl = level_word(nlv, t);
                f32x4 tk[2 * AW];                              // the attention layer's K / V window rows: they depend on p alone
                if (PRO == P_ATTN) {
                    const float* KVb = a.KV + (size_t)grow * a.N_keys * 2 * CH_D + c;
#pragma unroll
                    for (int i = 0; i < AW; ++i) {
                        const int key = min(p + i, a.N_keys - 1);
                        tk[i] = *(const f32x4*)(KVb + (size_t)key * 2 * CH_D);
                        tk[AW + i] = *(const f32x4*)(KVb + (size_t)key * 2 * CH_D + CH_D);
                    }
                }
                // ---- 2. this wave's raw row of the producing layer (layer 0: the last layer of the previous step): first pass of the
                //         sweep requested now, looked at after the older taps' share of the contraction
                CH_STAMP(0);
                // (before it: this workgroup's columns of the other parity's slot of this layer go back to the sentinel -- see CH_SENT)
                if (cols && tid < 16 * R && (tid & 3) == 1) {
                    const f32x4 sv = {__uint_as_float(CH_SENT), __uint_as_float(CH_SENT), __uint_as_float(CH_SENT), __uint_as_float(CH_SENT)};
                    // (the row goes into the LANE offset: a scalar offset that differs between lanes makes the compiler loop over its values)
                    vst(slot_off(t + 1, l, row0), ((tid >> 4) * RUN_GCOLS + n0 + (tid & 12)) * 4, sv);
                }
                const int in_off = slot_off(PRO == P_MEL ? t - 1 : t, PRO == P_MEL ? NL - 1 : l - 1, grow);
                f32x4 ga = zero4, gu = zero4;
                if (!no_input) {
                    ga = vld(in_off, ci * 4);
                    if (two) gu = vld(in_off, (CH_D + c) * 4);
                }
                f32x4 acc[RQ][2], accq[RQ];
#pragma unroll
                for (int rq = 0; rq < RQ; ++rq) { acc[rq][0] = zero4; acc[rq][1] = zero4; accq[rq] = zero4; }
                if (CON == C_HC3) {
                    // taps x[t-2r], x[t-r]: their rows arrived during the previous layer; chunks i < 2 PT of every wave's list are
                    // contracted while the sweep's round trip is in flight
                    *(f32x4*)(xrow + c) = tpok0 ? tp0 : zero4;
                    *(f32x4*)(xrow + CH_D + c) = tpok1 ? tp1 : zero4;
                    __syncthreads();
                    f32x4 xt[2 * PT][RQ];
#pragma unroll
                    for (int i = 0; i < 2 * PT; ++i)
#pragma unroll
                        for (int rq = 0; rq < RQ; ++rq) xt[i][rq] = *(const f32x4*)(xa + rq * 4 * LDXS + (w + R * i) * 16);
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
                f32x4 av = zero4, uv = zero4;
                int passes = 0;
                if (!no_input) {
                    bool ok = true;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ok = ok && (__float_as_uint(ga[e]) != CH_SENT || !cok);
                    if (two) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) ok = ok && __float_as_uint(gu[e]) != CH_SENT;
                    }
                    passes = 1;
                    if (!__all(ok)) passes += sweep_vals(in_off, ci * 4, cok, (CH_D + c) * 4, two, ga, gu);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { av[e] = cok ? ga[e] : 0.f; if (two) uv[e] = gu[e]; }
                }
                // every request of this wave has returned -- the sentinel store above included: it is acknowledged before this layer's
                // publish is issued (the protocol's one ordering requirement)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                CH_STAMP(1);
                if (STAMPS && stp && lane == 0) stp[l * 8 + 6] = passes;
                desc_pin(nxt);

                // ---- 3. prologue (one row per wave).  [TF-sem] layer_norm: mean, biased variance, eps 1e-12 (modules.py:65)
                f32x4 x;
                {
                    const float invc = __builtin_amdgcn_rcpf((float)cin);
                    float s1 = av[0] + av[1] + av[2] + av[3], s2 = uv[0] + uv[1] + uv[2] + uv[3];
                    s1 = wave_sum(s1);
                    if (two) s2 = wave_sum(s2);
                    const float m1 = s1 * invc, m2 = s2 * invc;
                    float q1 = 0.f, q2 = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float d1 = cok ? av[e] - m1 : 0.f, d2 = uv[e] - m2;
                        av[e] = d1; uv[e] = d2;
                        q1 += d1 * d1; q2 += d2 * d2;
                    }
                    q1 = wave_sum(q1);
                    if (two) q2 = wave_sum(q2);
                    const float r1 = fast_rsqrt(q1 * invc + LN_EPS);
                    if (two) {          // highway: g = sigmoid(LN1(H1)), y = g * LN2(H2) + (1 - g) * x   (modules.py:194-203)
                        const float r2 = fast_rsqrt(q2 * invc + LN_EPS);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float h1 = av[e] * r1 * g1v[e] + b1v[e], h2 = uv[e] * r2 * g2v[e] + b2v[e];
                            const float gte = fast_sigmoid(h1);
                            x[e] = gte * h2 + (1.0f - gte) * xprev[e];
                        }
                    } else {
                        const int act = PRO == P_MEL ? ACT_SIGMOID : cur.act();
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
                if (PRO == P_MEL) {
                    if (no_input) x = (t > 0 && c < a.ldy) ? *(const f32x4*)(a.Ytm + ((size_t)t * Bpad + grow) * a.ldy + c) : zero4;
                    // x is mel frame t-1 -> Y[b][t-1] and the decoder input S[t] (synthesize.py:204-209); written through: SSRN chunks
                    // on another stream read the frames while this launch is still running
                    if (t >= 1 && g == 0 && grow < a.B && t - 1 <= stop_v && c < a.ldy) {
                        st_coherent(a.Yout + ((size_t)grow * a.max_T + (t - 1)) * a.ldy + c, x);
                        *(f32x4*)(a.Ytm + ((size_t)t * Bpad + grow) * a.ldy + c) = x;
                    }
                }
                xprev = x;
                // ---- 4. stage the operand row
                const bool live = t <= stop_v;
                const int kc = cur.kc();
                if (PRO == P_ATTN) {
                    // R' = concat(softmax(Q K^T / sqrt(d)) V, Q) for row t under the current mask (networks.py:300-319): only the window
                    // [p, p + win) is unmasked; logits / probabilities one per LANE (lane i <-> key p + i)
                    const int nwin = min(a.win, a.N_keys - p);
                    const float scale = fast_rsqrt((float)CH_D);            // tf.rsqrt(tf.to_float(hp.d))  networks.py:300
                    float scl = -INFINITY;
#pragma unroll
                    for (int i = 0; i < AW; ++i) {
                        if (i < nwin) {
                            const float sdot = wave_sum(x[0] * tk[i][0] + x[1] * tk[i][1] + x[2] * tk[i][2] + x[3] * tk[i][3]) * scale;
                            if (lane == i) scl = sdot;
                        }
                    }
                    float mx = -INFINITY;
#pragma unroll
                    for (int i = 0; i < AW; ++i) if (i < nwin) mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, scl), i)));
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
                            for (int e = 0; e < 4; ++e) ctx[e] += pi * tk[AW + i][e];
                        }
                    }
                    *(f32x4*)(xrow + c) = ctx;
                    *(f32x4*)(xrow + CH_D + c) = x;
                    const int m = p + arg;
                    if (g == 0 && live && grow < a.B) {
                        st_coherent(a.Qhist + ((size_t)t * Bpad + grow) * CH_D + c, x);          // read by the cone kernels
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
                    constexpr int xcur = CON == C_HC3 ? 2 * CH_D : 0;
                    *(f32x4*)(xrow + xcur + c) = x;
                    if (CON == C_K1) {
                        const int ccat = cur.ccat();
                        if (ccat > 0) {       // speaker embedding appended to the input (row 0 of the table reads as zeros, modules.py:38-40)
                            const float* tab = cur.cat_table();
                            for (int j = lane; j < kc - CH_D; j += 64) xrow[CH_D + j] = (spk == 0 || j >= ccat) ? 0.f : tab[(size_t)spk * ccat + j];
                        }
                    }
                    if (CON == C_HC3 && cur.tapkind() == 1 && g == 0 && live) st_coherent(cur.hist() + ((size_t)t * Bpad + grow) * CH_D + c, x);
                }
                CH_STAMP(2);
                __syncthreads();
                CH_STAMP(3);
                // ---- 5. R x 16 slice on the 4x4x1 MFMA, K split round-robin over the R waves
                if (cols) {
                    if (CON == C_HC3) {
                        f32x4 xf[PT][RQ];
#pragma unroll
                        for (int i = 0; i < PT; ++i)
#pragma unroll
                            for (int rq = 0; rq < RQ; ++rq) xf[i][rq] = *(const f32x4*)(xa + rq * 4 * LDXS + (w + R * (2 * PT + i)) * 16);
#pragma unroll
                        for (int i = 0; i < PT; ++i)
#pragma unroll
                            for (int rq = 0; rq < RQ; ++rq) {
                                acc[rq][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(xf[i][rq][0], bfrag[2 * PT + i][0], acc[rq][0], 0, 0, 0);
                                acc[rq][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(xf[i][rq][1], bfrag[2 * PT + i][1], acc[rq][1], 0, 0, 0);
                                acc[rq][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(xf[i][rq][2], bfrag[2 * PT + i][2], acc[rq][0], 0, 0, 0);
                                acc[rq][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(xf[i][rq][3], bfrag[2 * PT + i][3], acc[rq][1], 0, 0, 0);
                            }
                    } else {
                        // k = 1 layers: K = kc (96 ... 512) = nch chunks; the attention layer's operand is [context | Q] and the chunks of its
                        // Q half also go to their own accumulator (QW, the cone head's cache)
                        const int nch = kc >> 4;
                        constexpr int NI = 4;      // chunks per wave: K <= 512
                        f32x4 xf[NI][RQ];
#pragma unroll
                        for (int i = 0; i < NI; ++i) {
                            if (w + R * i < nch) {
#pragma unroll
                                for (int rq = 0; rq < RQ; ++rq) xf[i][rq] = *(const f32x4*)(xa + rq * 4 * LDXS + (w + R * i) * 16);
                            }
                        }
#pragma unroll
                        for (int i = 0; i < NI; ++i) {
                            if (w + R * i < nch) {
                                if (PRO == P_ATTN && i >= 2) {
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
                    }
                    // the 4 k-lanes of a column sit 4 lanes apart in one 16-lane row: two DPP row rotations sum them, k-lane 0 writes
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
                        if (mkk == 0) *(f32x4*)(part + ((w * 16 + mcol) * RQ + rq) * 4) = v;
                        if (PRO == P_ATTN) {
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
                CH_STAMP(7);
                CH_STAMP(4);
                __syncthreads();
                if (cols && tid < 16 * R) {
                    const int row = tid >> 4, col = tid & 15;
                    const float* pr = part + (col * RQ + (row >> 2)) * 4 + (row & 3);
                    float pv[R];
#pragma unroll
                    for (int ww = 0; ww < R; ++ww) pv[ww] = pr[ww * 16 * RQ * 4];
                    float v = bias_v;
#pragma unroll
                    for (int ww = 0; ww < R; ww += 2) v += pv[ww] + pv[ww + 1];
                    {   // four adjacent columns of a row sit in one quad: lane (col & 3) == 0 stores the 16 bytes
                        if (__float_as_uint(v) == CH_SENT) v = __uint_as_float(0x7FC00000u);
                        f32x4 q;
                        q[0] = dpp_mov<0x00>(v); q[1] = dpp_mov<0x55>(v); q[2] = dpp_mov<0xAA>(v); q[3] = dpp_mov<0xFF>(v);
                        if ((col & 3) == 0) vst(slot_off(t, l, row0), (row * RUN_GCOLS + n0 + col) * 4, q);
                    }
                    if (PRO == P_ATTN) {
                        // QW[t] = Q[t] . Wq + bias for the cone head's cache (written through: the cone kernels read it after their acquire)
                        const float* pq = partq + (col * RQ + (row >> 2)) * 4 + (row & 3);
                        float vq = bias_v;
#pragma unroll
                        for (int ww = 0; ww < R; ww += 2) vq += pq[ww * 16 * RQ * 4] + pq[(ww + 1) * 16 * RQ * 4];
                        if (row0 + row < a.B)
                            __hip_atomic_store(a.QW + ((size_t)t * Bpad + row0 + row) * CH_D + n0 + col, vq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                CH_STAMP(5);
                if (PRO == P_ATTN) {
                    // release the cone of step t+1 on the side stream: Q[t], QW[t] and prev_max are written through; once every
                    // workgroup of the layer has arrived, one lane raises the word the cone's first kernel polls
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (tid == 0) {
                        const int old = atomicAdd(a.ctl + 3, 1);
                        if (old + 1 == (Bpad / R) * a.attn_slices * (t + 1 - a.t_begin)) {
                            __hip_atomic_fetch_max(a.sig, a.sig_base + (unsigned)t + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            __hip_atomic_store((int*)a.host_progress, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        }
                    }
                }
                // ---- 6. behind the publish: the next layer's cone level, weights, bias and taps
                level_wait(nlv, t, sigl);
                fetch_layer(nxt, l + 1 < NL ? t : t + 1);
            };
            const bool hc3 = cur.ntaps() == 3;
            if (l == 0) layer(ic<P_MEL>(), ic<C_K1>());
            else if (pre == RUN_ATTN) layer(ic<P_ATTN>(), ic<C_K1>());
            else if (pre == RUN_HC) { if (hc3) layer(ic<P_HC>(), ic<C_HC3>()); else layer(ic<P_HC>(), ic<C_K1>()); }
            else { if (hc3) layer(ic<P_CONV>(), ic<C_HC3>()); else layer(ic<P_CONV>(), ic<C_K1>()); }
            cur = nxt;
#undef CH_STAMP
        }
        if (STAMPS && stp && lane == 0) {     // shader clock over this step: (c1 - c0) cycles in (w1 - w0) * 10 ns
            long long* q = stp + (LOOP_MAX_LAYERS - 1) * 8;
            q[0] = step_w0; q[1] = wall_clock64(); q[2] = step_c0; q[3] = clock64();
        }
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
            sweep_vals(slot_off(t_last, NL - 1, grow), (cok ? c : 0) * 4, cok, 0, false, av, uv);
#pragma unroll
            for (int e = 0; e < 4; ++e) av[e] = cok ? av[e] : 0.f;
            const float invc = __builtin_amdgcn_rcpf((float)cin);
            const float m1 = wave_sum(av[0] + av[1] + av[2] + av[3]) * invc;
            float q1 = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d1 = cok ? av[e] - m1 : 0.f; av[e] = d1; q1 += d1 * d1; }
            const float r1 = fast_rsqrt(wave_sum(q1) * invc + LN_EPS);
            f32x4 x;
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = cok ? fast_sigmoid(av[e] * r1 * g1v[e] + b1v[e]) : 0.f;
            if (grow < a.B && c < a.ldy) {
                *(f32x4*)(a.Yout + ((size_t)grow * a.max_T + t_last) * a.ldy + c) = x;
                *(f32x4*)(a.Ytm + ((size_t)(t_last + 1) * Bpad + grow) * a.ldy + c) = x;
            }
        }
    }
    // whatever happened, the side stream's remaining waits must not wait for steps that never ran
    if (g == 0 && blockIdx.y == 0 && tid == 0)
        __hip_atomic_fetch_max(a.sig, a.sig_base + (unsigned)a.max_T + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (a.clk && tid == 0) atomicMax((unsigned long long*)a.clk + 1, (unsigned long long)wall_clock64());      // device-side witness: last workgroup out
}

static size_t dec_chain_lds_bytes() { return (size_t)(2 * CH_R * 16 * CH_RQ * 4 + CH_R * CH_LDXS) * 4; }

void launch_dec_chain(const LoopArgs& a, int col_slices, hipStream_t s) {
    static thread_local std::map<int, bool> done;
    const size_t lds_bytes = dec_chain_lds_bytes();
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!done[dev]) {
        (void)hipFuncSetAttribute((const void*)dec_chain<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        (void)hipFuncSetAttribute((const void*)dec_chain<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        done[dev] = true;
    }
    if (a.stamps) hipLaunchKernelGGL(dec_chain<true>, dim3(col_slices, a.Bpad / CH_R), dim3(64 * CH_R), lds_bytes, s, a);
    else hipLaunchKernelGGL(dec_chain<false>, dim3(col_slices, a.Bpad / CH_R), dim3(64 * CH_R), lds_bytes, s, a);
}
// workgroups of dec_chain that fit on one CU at once (the kernel needs ALL of its workgroups resident)
int dec_chain_blocks_per_cu() {
    const size_t lds_bytes = dec_chain_lds_bytes();
    (void)hipFuncSetAttribute((const void*)dec_chain<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)dec_chain<true>, 64 * CH_R, lds_bytes) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

}  // namespace oph
