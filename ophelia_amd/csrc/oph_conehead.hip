// cone_head: level 0 of the AudioDec history cone of one decode step in one launch (ConeHeadArgs, oph_internal.h) -- moved out of
// oph_kernels.hip in round 5.
#include "oph_internal.h"
#include "oph_device.h"

namespace oph {

// cone_head: see ConeHeadArgs (oph_internal.h).  One launch, 16 waves per workgroup, lane l owns channels 4l..4l+3
// (d <= 256).  Workgroup b < B belongs to utterance b's NEWEST history position (index i_new, time j - off[i_new]): its
// 16 waves first compute that position's Q . Wq + bias row (K split over the waves, every weight request issued before
// the first use), cache it in QW, and wave 0 finishes the row.  The other workgroups take 16 rows each of the remaining
// positions, reading the cached QW.  First launch of a cone in dec_loop mode: waits for the loop kernel's signal.
// WF: window positions handled (4: every request of the row -- Q, the window's K rows, its V . Wc rows, gamma, beta -- is issued
// before the first use, rows past the window clamped and masked out of the arithmetic; as loads under `if (w < nwin)` they were
// 2 win + 3 dependent round trips per row, most of this launch's 8-11 us.  8: the general form, windows of 5..8 keys)
template <int WF>
static __device__ __forceinline__ void cone_head_row(const ConeHeadArgs& a, int i, int b, int tq, const f32x4& qw, int lane, int p) {
    const int d = a.d, c = lane * 4;
    const bool cok = c < d;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const size_t qrow = ((size_t)tq * a.Bpad + b) * d;
    // attention window [p, p+win) under the CURRENT mask (networks.py:300-315); p = a.p[b], requested by the caller beside the row's time index
    const f32x4 q = cok ? *(const f32x4*)(a.Q + qrow + c) : zero4;
    const int nwin = min(a.win, a.N_keys - p);
    const float* Kb = a.KV + (size_t)b * a.N_keys * 2 * d;
    const float* VWb = a.VW + (size_t)b * a.N_keys * a.ldvw;
    const float scale = 1.0f / sqrtf((float)d);
    f32x4 kvs[WF], vws[WF], gpre = zero4, bpre = zero4;
    if constexpr (WF <= 4) {
#pragma unroll
        for (int w = 0; w < WF; ++w) {
            const int row = min(p + w, a.N_keys - 1);
            kvs[w] = cok ? *(const f32x4*)(Kb + (size_t)row * 2 * d + c) : zero4;
            vws[w] = cok ? *(const f32x4*)(VWb + (size_t)row * a.ldvw + c) : zero4;
        }
        if (cok) { gpre = *(const f32x4*)(a.gamma + c); bpre = *(const f32x4*)(a.beta + c); }
    }
    float sc[WF], mx = -INFINITY;
#pragma unroll
    for (int w = 0; w < WF; ++w) {
        sc[w] = -INFINITY;
        if constexpr (WF <= 4) {
            const f32x4 kv = kvs[w];
            const float v = wave_sum(q[0] * kv[0] + q[1] * kv[1] + q[2] * kv[2] + q[3] * kv[3]) * scale;
            if (w < nwin) { sc[w] = v; mx = fmaxf(mx, v); }
        } else if (w < nwin) {
            const f32x4 kv = cok ? *(const f32x4*)(Kb + (size_t)(p + w) * 2 * d + c) : zero4;
            sc[w] = wave_sum(q[0] * kv[0] + q[1] * kv[1] + q[2] * kv[2] + q[3] * kv[3]) * scale;
            mx = fmaxf(mx, sc[w]);
        }
    }
    float den = 0.f, pr[WF];
#pragma unroll
    for (int w = 0; w < WF; ++w) { pr[w] = w < nwin ? expf(sc[w] - mx) : 0.f; den += pr[w]; }
    f32x4 h = qw;
#pragma unroll
    for (int w = 0; w < WF; ++w) {
        if (w < nwin) {
            const float pw = pr[w] / den;
            f32x4 vw;
            if constexpr (WF <= 4) vw = vws[w];
            else vw = cok ? *(const f32x4*)(VWb + (size_t)(p + w) * a.ldvw + c) : zero4;
#pragma unroll
            for (int n = 0; n < 4; ++n) h[n] = fmaf(pw, vw[n], h[n]);
        }
    }
    // LayerNorm (modules.py:137-139; C_1 has no activation)
    const float invd = 1.0f / (float)d;
    const float mean = a.nonorm ? 0.f : wave_sum(h[0] + h[1] + h[2] + h[3]) * invd;
    float qq = 0.f;
#pragma unroll
    for (int n = 0; n < 4; ++n) { const float dl = cok ? h[n] - mean : 0.f; h[n] = dl; qq += dl * dl; }
    const float rstd = a.nonorm ? 1.0f : 1.0f / sqrtf(wave_sum(qq) * invd + LN_EPS);
    float* y = a.Y + ((size_t)i * a.Bpad + b) * a.ldy;
    if (cok) {
        f32x4 g, bt;
        if constexpr (WF <= 4) { g = gpre; bt = bpre; }
        else { g = *(const f32x4*)(a.gamma + c); bt = *(const f32x4*)(a.beta + c); }
        f32x4 o;
#pragma unroll
        for (int n = 0; n < 4; ++n) o[n] = h[n] * rstd * g[n] + bt[n];
        if ((a.done_sig && (i == a.coh0 || i == a.coh1))) st_coherent(y + c, o);       // a row the RUNNING chain reads
        else *(f32x4*)(y + c) = o;
        if (a.Yh) {
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            h4 hi, lo;
#pragma unroll
            for (int n = 0; n < 4; ++n) { hi[n] = (_Float16)o[n]; lo[n] = (_Float16)(o[n] - (float)hi[n]); }
            const size_t po = ((size_t)(c >> 6) * a.nrows + (size_t)i * a.Bpad + b) * 64 + (c & 63);
            *(h4*)((_Float16*)a.Yh + po) = hi;
            *(h4*)((_Float16*)a.Yl + po) = lo;
        }
    }
    int ctot = d;
    if (a.spk_table) {
        const int id = a.spk_ids[b];
        for (int c2 = lane; c2 < a.spk_dim; c2 += 64) y[d + c2] = id == 0 ? 0.f : a.spk_table[(size_t)id * a.spk_dim + c2];
        ctot += a.spk_dim;
    }
    for (int c2 = ctot + lane; c2 < a.ldy; c2 += 64) y[c2] = 0.f;
}
static __device__ __forceinline__ void cone_head_body(const ConeHeadArgs& a) {
    __shared__ float ps[16][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool newest = a.i_new >= 0 && (int)blockIdx.x < a.B;
    int pos_b;                              // position index this workgroup's rows belong to (one position: Bpad % 16 == 0)
    // i_new < 0: every position's Q . Wq row is already cached (dec_loop's attention layer emits it): no special workgroups
    const int nb_new = a.i_new >= 0 ? a.B : 0;
    const int RB = a.rb;                     // rows per workgroup: 16, or 4 when no workgroup has a Q . Wq row to compute (quicker to start)
    if (newest) pos_b = a.i_new;
    else { pos_b = ((int)blockIdx.x - nb_new) * RB / a.Bpad; if (a.i_new >= 0 && pos_b >= a.i_new) ++pos_b; }
    if (a.wait_sig) {
        if (threadIdx.x == 0) {
            long long t0 = 0;
            for (int it = 0; (int)(__hip_atomic_load(a.wait_sig, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - a.wait_val) < 0; ++it) {
                __builtin_amdgcn_s_sleep(8);
                if ((it & 255) == 255) {
                    const long long now = wall_clock64();
                    if (t0 == 0) t0 = now;
                    if (now - t0 > 200000000LL || __hip_atomic_load(a.ctl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                        __hip_atomic_store(a.ctl + 2, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    // (the stop word is written by the running decode kernel: read past the L1; a pipelined launch was dispatched long before its step)
    // (round 5: the row's two scalars -- prev_max of its utterance, its position's time offset -- are requested BESIDE the stop word,
    //  not behind the branch on it: one round trip instead of two in front of the row's own requests)
    const int rl_o = ((int)blockIdx.x - nb_new) * RB + w;       // row among the positions other than the newest
    int i_o = rl_o / a.Bpad; const int b_o = rl_o - i_o * a.Bpad;
    if (a.i_new >= 0 && i_o >= a.i_new) ++i_o;
    const bool row_o = !newest && i_o < a.npos && b_o < a.B;
    const int pb_o = row_o ? a.p[b_o] : 0, off_o = row_o ? a.off[i_o] : 0;
    // (the stop word LAST: its value is wanted at once -- the comparison is wave-uniform -- and the wait for it covers the two above)
    const int stop_v = a.ctl ? __hip_atomic_load(a.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x7fffffff;
    const bool live = !(a.ctl && a.t > stop_v);
    const int d = a.d;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    if (live && newest) {
        const int b = blockIdx.x, tq = a.j - a.off[a.i_new];
        if (tq >= 0) {                      // block-uniform
            const size_t qrow = ((size_t)tq * a.Bpad + b) * d;
            const int kper = (d + 15) / 16, k0 = w * kper, c = lane * 4;
            f32x4 wv[16]; float qv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int k = k0 + i;
                const bool in = i < kper && k < d;
                wv[i] = (in && c < d) ? *(const f32x4*)(a.Wq + (size_t)k * a.ldn + c) : zero4;
                qv[i] = in ? a.Q[qrow + k] : 0.f;
            }
            f32x4 acc = zero4;
#pragma unroll
            for (int i = 0; i < 16; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = fmaf(qv[i], wv[i][e], acc[e]);
            *(f32x4*)&ps[w][lane * 4] = acc;
            __syncthreads();
            if (w == 0) {
                f32x4 qw = zero4;
                if (c < d) {
                    qw = *(const f32x4*)(a.bias + c);
#pragma unroll
                    for (int ww = 0; ww < 16; ++ww) qw += *(const f32x4*)&ps[ww][c];
                    *(f32x4*)(a.QW + qrow + c) = qw;          // cached: every later step reads this position's term
                }
                const int pb = a.p[b];
                if (a.win <= 4) cone_head_row<4>(a, a.i_new, b, tq, qw, lane, pb); else cone_head_row<ATT_WMAX>(a, a.i_new, b, tq, qw, lane, pb);
            }
        }
    } else if (live) {
        const int i = i_o, b = b_o;
        if (row_o) {
            const int pb = pb_o;
            const int tq = a.j - off_o;
            if (tq >= 0) {
                const int c = lane * 4;
                const f32x4 qw = c < d ? *(const f32x4*)(a.QW + ((size_t)tq * a.Bpad + b) * d + c) : zero4;
                if (a.win <= 4) cone_head_row<4>(a, i, b, tq, qw, lane, pb); else cone_head_row<ATT_WMAX>(a, i, b, tq, qw, lane, pb);
            }
        }
    }
    if (a.done_sig && (pos_b == a.coh0 || pos_b == a.coh1)) {       // cone level 0: its tap rows are written, raise its word (see ln_rows)
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
__global__ __launch_bounds__(1024) void cone_head(ConeHeadArgs a) { cone_head_body(a); }

void launch_cone_head(const ConeHeadArgs& a, hipStream_t s) {
    ConeHeadArgs c = a;
    if (a.i_new < 0) { c.rb = 4; hipLaunchKernelGGL(cone_head, dim3((a.npos * a.Bpad + 3) / 4), dim3(256), 0, s, c); return; }
    const int others = (a.npos - 1) * a.Bpad;
    c.rb = 16;
    hipLaunchKernelGGL(cone_head, dim3(a.B + (others + 15) / 16), dim3(1024), 0, s, c);
}

}  // namespace oph
