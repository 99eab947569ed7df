// libophelia_hip.so -- the AudioDec history cone of one decode step (side stream): networks.py:360-435 re-evaluated under the current
// attention mask for the positions the newest frame depends on.
#include "oph_host.h"

// AudioDec history cone for step t under the mask p_t (= max_attentions of step t-1).
// The reference re-evaluates R[t'] for ALL t' <= t with the current prev_max (networks.py:311
// tiles one mask over max_T), so AudioDec's hidden history cannot be cached across steps; what is
// recomputed here is the sparse receptive cone of row t: the highway-layer inputs at the history
// offsets Hset[k] (84, 82, 44, 14, 4, 2 positions for rates 1,3,9,27,1,1).  It depends only on
// p_t and Q[<t], both known right after attn_step(t-1): it runs on the SIDE stream, concurrently
// with the AudioEnc chain of step t, into the ping-pong buffer cone[t&1].
// Do the workgroups of every hc_fused launch of a cone fit the cone partition at once (the column tiles of a row block wait for each
// other's statistics)?  Evaluated once per handle (hcf_capacity: -1 unknown, 0 no -- also set after a time-out --, 1 yes).
bool hcf_fits(oph_handle* h) {
    if (h->hcf_capacity < 0) {
        const int nh = h->n_hc_dec;
        int ncu = 0;
        for (int i = 0; i < h->mask_words; ++i) ncu += __builtin_popcount(h->m_conep[i]);
        if (h->mask_words == 0) { hipDeviceProp_t prop; ncu = hipGetDeviceProperties(&prop, h->device) == hipSuccess ? prop.multiProcessorCount : 0; }
        bool fits = true;
        for (int k = 0; k + 1 < nh; ++k) {
            const int Mk = (int)h->Hset[k + 1].size() * h->Bpad;
            fits = fits && hc_fused_grid(Mk) <= hc_fused_blocks_per_cu(Mk) * ncu;
        }
        h->hcf_capacity = fits ? 1 : 0;
    }
    return h->hcf_capacity != 0;
}

void launch_cone(oph_handle* h, int t) {
    const oph_dims& m = h->dm;
    const int d = m.d, Bpad = h->Bpad, B = h->B;
    int* stop_after = h->d_ctl + 1;
    const int* pcur = h->d_p + (t & 1) * Bpad;
    const int pre = h->dec_pre, nh = h->n_hc_dec;
    std::vector<float*>& cone = h->cone[t & 1];
    hipStream_t saved = g_cur;
    g_cur = h->scone;
    const int n0 = (int)h->Hset[0].size();
    AttnRowsArgs ar{};
    ar.mode = 0; ar.Q = h->Qhist; ar.ldq = d; ar.K = h->KV; ar.V = h->KV + d; ar.ldkv = 2 * d; ar.N = m.max_N; ar.d = d;
    ar.win = m.attention_win_size; ar.p = pcur; ar.B = B; ar.Bpad = Bpad; ar.nrows = n0 * Bpad; ar.off = h->d_off0; ar.j = t;
    ar.R = h->coneR; ar.ldr = 2 * d; ar.stop_after = stop_after; ar.t = t;
    if (m.flags & OPH_FLAG_NO_MONOTONIC) ar.ends = h->d_ends;
    if (h->fixed_att) ar.ptab = h->d_ptab;
    if (h->cone_inline_sig) { ar.wait_sig = h->d_sig; ar.wait_val = h->cone_wait_val; ar.wait_err = h->d_ctl + 2; }
    int pre_first = 0;               // first k=1 layer still to run as GEMM + LayerNorm
    const bool head = h->cone_head_ok && !h->fixed_att;
    // dec_loop mode: the launch that completes cone level `lvl` (nblocks workgroups) raises that level's word
    auto stamp_of = [&](int lvl) -> long long* { return (h->d_lvldbg && lvl >= 0 && lvl < 8 && t < m.max_T) ? h->d_lvldbg + (size_t)t * 8 + lvl : nullptr; };
    auto level_done = [&](int lvl, unsigned*& sig, unsigned& val, unsigned*& count, unsigned& target, int& coh0, int& coh1) {
        if (!h->cone_inline_sig || lvl < 0 || lvl >= LOOP_MAX_LEVELS || lvl >= nh) return;
        const Layer& tl = h->audiodec[pre + lvl];         // the chain layer whose taps read this level (build_loop_layers)
        coh0 = idx_of(h->Hset[lvl], -tl.off[0]); coh1 = idx_of(h->Hset[lvl], -tl.off[1]);
        h->cone_done_total[lvl] += (unsigned)((coh0 >= 0) + (coh1 >= 0 && coh1 != coh0)) * (unsigned)(Bpad / 4);
        sig = h->d_sig + LOOP_SIG_LEVEL0 + 16 * lvl; val = h->cone_done_val; count = h->d_cone_count + lvl; target = h->cone_done_total[lvl];
    };
    if (head) {
        const Layer& c1 = h->audiodec[0];
        ConeHeadArgs ch{};
        ch.Q = h->Qhist; ch.d = d; ch.KV = h->KV; ch.N_keys = m.max_N; ch.win = m.attention_win_size; ch.VW = h->VW; ch.ldvw = h->ldvw;
        ch.QW = h->QWhist; ch.Wq = c1.Wkn + (size_t)d * c1.ldn; ch.ldn = c1.ldn; ch.bias = c1.bias; ch.gamma = c1.g1; ch.beta = c1.b1; ch.nonorm = !c1.ln;
        ch.p = pcur; ch.B = B; ch.Bpad = Bpad; ch.nrows = n0 * Bpad; ch.off = h->d_off0; ch.j = t;
        const bool spk_next = pre > 1 && h->audiodec[1].ccat > 0;
        if (spk_next) { ch.Y = h->coneTmp; ch.ldy = h->audiodec[1].kc; ch.spk_table = h->emb_spk; ch.spk_ids = h->d_spk; ch.spk_dim = h->audiodec[1].ccat; }
        else { ch.Y = cone[0]; ch.ldy = h->audiodec[pre].kc; }
        ch.ctl = h->d_ctl; ch.t = t;
        const bool fused = h->cone_fused_ok && !spk_next && h->cone_prec == 2 && h->hcf_capacity != 0;
        if (fused) { ch.Yh = h->coneH[t & 1][0]; ch.Yl = h->coneL[t & 1][0]; }
        ch.wait_sig = ar.wait_sig; ch.wait_val = ar.wait_val;
        ch.npos = n0; ch.i_new = 0;
        for (int i = 1; i < n0; ++i) if (h->Hset[0][i] < h->Hset[0][ch.i_new]) ch.i_new = i;
        if (h->qw_from_loop) ch.i_new = -1;          // dec_loop's attention layer wrote QW[t-1] before it released this cone
        if (!spk_next && h->cone_inline_sig) {
            // level 0's tap rows: the newest position is spread over B workgroups (one per utterance), any other over Bpad/16
            const Layer& tl = h->audiodec[pre];
            ch.coh0 = idx_of(h->Hset[0], -tl.off[0]); ch.coh1 = idx_of(h->Hset[0], -tl.off[1]);
            auto blocks_of = [&](int pos) { return pos < 0 ? 0u : (ch.i_new < 0 ? (unsigned)(Bpad / 4) : (pos == ch.i_new ? (unsigned)B : (unsigned)(Bpad / 16))); };
            h->cone_done_total[0] += blocks_of(ch.coh0) + (ch.coh1 != ch.coh0 ? blocks_of(ch.coh1) : 0u);
            ch.done_sig = h->d_sig + LOOP_SIG_LEVEL0; ch.done_val = h->cone_done_val; ch.done_count = h->d_cone_count; ch.done_target = h->cone_done_total[0];
            ch.done_stamp = stamp_of(0);
        }
        h->pbegin(PC_CONEHEAD);
        launch_cone_head(ch, g_cur);
        h->pend(PC_CONEHEAD, ((double)n0 * B * (3.0 * d + 2.0 * m.attention_win_size * d) + (double)d * d) * 4.0, (double)n0 * B * 4.0 * m.attention_win_size * d + 2.0 * B * d * d);
        pre_first = 1;
        if (fused) {
            // levels 1 .. nh-1: one hc_fused launch each (contraction on the planes + LayerNorm x 2 + gate + mix)
            // the 8 workgroups of a row block exchange statistics: every workgroup of a launch must be resident (the cone's launches run
            // one after the other on their own CU partition)
            const bool fits = hcf_fits(h);
            if (!fits) h->hcf_capacity = 0;        // (this launch already wrote the planes; harmless) -> the unfused path from here on
            else {
                if (h->hcf_epoch > 0xF0000000u) {
                    hipStreamSynchronize(h->scone);
                    hipMemsetAsync(h->d_hcf_stats, 0, (size_t)2 * nh * h->hcf_stats_stride * sizeof(unsigned long long), g_cur);
                    h->hcf_epoch = 0;
                }
                for (int k = 0; k + 1 < nh; ++k) {
                    const Layer& l = h->audiodec[pre + k];
                    const int n_out = (int)h->Hset[k + 1].size();
                    HcFusedArgs f{};
                    f.Xh = h->coneH[t & 1][k]; f.Xl = h->coneL[t & 1][k]; f.in_rows = (int)h->Hset[k].size() * Bpad; f.Xres = cone[k]; f.restab = h->d_res[k];
                    f.tab = h->d_tab[k]; f.need = h->d_need[k]; f.n_out = n_out; f.j = t; f.Bpad = Bpad; f.M = n_out * Bpad;
                    f.Wh = l.Wph; f.Wl = l.Wpl; f.bias = l.bias_p; f.g1 = l.g1; f.b1 = l.b1; f.g2 = l.g2; f.b2 = l.b2;
                    f.Y = cone[k + 1]; f.Yh = h->coneH[t & 1][k + 1]; f.Yl = h->coneL[t & 1][k + 1];
                    // (the statistics regions alternate with the step parity)
                    f.stats = h->d_hcf_stats + ((size_t)k * 2 + (t & 1)) * h->hcf_stats_stride; f.epoch = ++h->hcf_epoch; f.ctl = h->d_ctl; f.zeros = h->d_zeros;
                    f.t = t;
                    if (h->cone_inline_sig && k + 1 < LOOP_MAX_LEVELS) {
                        const Layer& tl = h->audiodec[pre + k + 1];
                        f.coh0 = idx_of(h->Hset[k + 1], -tl.off[0]); f.coh1 = idx_of(h->Hset[k + 1], -tl.off[1]);
                        h->cone_done_total[k + 1] += (unsigned)hc_fused_holders(f.M, Bpad, f.coh0, f.coh1);
                        f.done_sig = h->d_sig + LOOP_SIG_LEVEL0 + 16 * (k + 1); f.done_val = h->cone_done_val; f.done_count = h->d_cone_count + (k + 1); f.done_target = h->cone_done_total[k + 1];
                        f.done_stamp = stamp_of(k + 1);
                    }
                    if (h->d_cldbg && t == m.max_T / 2) f.dbg = h->d_cldbg + 8 * k;
                    h->pbegin(PC_HCFUSED);
                    launch_hc_fused(f, g_cur);
                    h->pend(PC_HCFUSED, ((double)f.M * 3.0 * l.cin + (double)f.M * l.cout + (double)l.N * 3.0 * l.cin) * 4.0, 2.0 * f.M * l.N * 3.0 * l.cin);
                }
                g_cur = saved;
                return;
            }
        }
    } else {
    h->pbegin(PC_ATTN_ROWS);
    launch_attn_rows(ar, g_cur);
    h->pend(PC_ATTN_ROWS, (double)n0 * B * 3.0 * d * 4.0, (double)n0 * B * 4.0 * m.attention_win_size * d);
    }
    // k=1 layers before the highway stack, on all Hset[0] positions
    const float* x = h->coneR; int ldx = 2 * d;
    if (pre_first == 1 && pre > 1) { x = h->coneTmp; ldx = h->audiodec[1].kc; }
    for (int k = pre_first; k < pre; ++k) {
        const Layer& l = h->audiodec[k];
        GemmArgs g{};
        g.X = x; g.ldx = ldx; g.Wt = l.Wt; g.ldw = l.kc; g.bias = l.bias; g.H = h->coneRaw; g.ldh = l.Nalloc;
        g.M = n0 * Bpad; g.N = l.N; g.kc = l.kc; g.ntaps = 1; g.mode = 0; g.T = g.M; g.off[0] = 0;
        g.stop_after = stop_after; g.t = t;
        g.ksplit = h->opt.cone_ksplit(g.M); g.split_stride = (long long)g.M * l.Nalloc;
        run_gemm(h, g, l.cin);
        EpiArgs e{};
        e.nsplit = g.ksplit; e.split_stride = g.split_stride;
        e.H = h->coneRaw; e.ldh = l.Nalloc; e.M = g.M; e.C = l.cout; e.mode = PRE_CONV; e.act = l.act; e.g1 = l.g1; e.b1 = l.b1; e.nonorm = !l.ln;
        e.lcc = l.lcc_gate; e.lcc_ids = h->d_spk; e.lcc_T = 0; e.Bpad = Bpad;
        e.Bpad = Bpad; e.stop_after = stop_after; e.t = t;
        const bool spk_next = (k + 1 < pre) && h->audiodec[k + 1].ccat > 0;
        if (spk_next) {
            const Layer& nx = h->audiodec[k + 1];
            e.Y = h->coneTmp; e.ldy = nx.kc; e.ypad = nx.kc;
            e.spk_table = h->emb_spk; e.spk_ids = h->d_spk; e.spk_dim = nx.ccat; e.spk_T = 0;
            x = h->coneTmp; ldx = nx.kc;
        } else {
            const Layer& hc0 = h->audiodec[pre];
            e.Y = cone[0]; e.ldy = hc0.kc; e.ypad = hc0.kc;
            x = cone[0]; ldx = hc0.kc;
            level_done(0, e.done_sig, e.done_val, e.done_count, e.done_target, e.coh0, e.coh1);
            if (e.done_sig) e.done_stamp = stamp_of(0);
        }
        run_epi(h, e);
    }
    // Small levels (few output rows) as ONE launch each: the previous layer's LayerNorm / gate as the prologue of this
    // layer's contraction (cone_fc16) instead of ln_rows + a split-K GEMM.  From the first such level to the end.
    const int fc_rows = h->opt.fc_rows >= 0 ? h->opt.fc_rows : h->cone_fc_rows;
    const int fc_in_split = h->opt.fc_insplit;
    int fc_from = nh;             // first layer index evaluated by cone_fc16
    for (int k = nh - 2; k >= 1; --k) {
        const Layer& l = h->audiodec[pre + k]; const Layer& lp = h->audiodec[pre + k - 1];
        const bool ok = (int)h->Hset[k + 1].size() * Bpad <= fc_rows && h->fc_tab[k].valid && lp.cout <= 256 && l.cin == lp.cout && l.kc <= 512 && l.ntaps == 3 &&
                        !l.lcc_gate && !lp.lcc_gate && l.ccat == 0 && (Bpad % 16) == 0;
        if (!ok) break;
        fc_from = k;
    }
    float* raw_in = h->coneRaw; int raw_split = 1; long long raw_stride = 0;
    for (int k = 0; k + 1 < nh; ++k) {
        const Layer& l = h->audiodec[pre + k];
        const int n_out = (int)h->Hset[k + 1].size();
        float* const raw_gemm = h->coneRaw;
        if (k >= fc_from) {
            const Layer& lp = h->audiodec[pre + k - 1];
            ConeFcArgs c{};
            c.rawp = raw_in; c.ldrawp = lp.Nalloc; c.nsplit = raw_split; c.split_stride = raw_stride;
            c.g1 = lp.g1; c.b1 = lp.b1; c.g2 = lp.g2; c.b2 = lp.b2; c.nonorm = !lp.ln; c.C = lp.cout;
            c.xres = cone[k - 1]; c.ldres = lp.kc; c.n_out = n_out; c.j = t;
            const FcTables& ft = h->fc_tab[k];
            memcpy(c.tab, ft.tab, sizeof c.tab); memcpy(c.need, ft.need, sizeof c.need); memcpy(c.res, ft.res, sizeof c.res);
            memcpy(c.extra, ft.extra, sizeof c.extra); memcpy(c.extra_res, ft.extra_res, sizeof c.extra_res); c.n_extra = ft.n_extra;
            c.xstore = cone[k]; c.ldx = l.kc;
            c.Wt = l.Wt; c.ldw = 3 * l.kc; c.bias = l.bias; c.kc = l.kc; c.N = l.N;
            c.H = raw_in == h->coneRawB ? raw_gemm : h->coneRawB; c.ldh = l.Nalloc;
            c.Bpad = Bpad; c.stop_after = stop_after; c.t = t;
            if (h->cone_inline_sig && k < LOOP_MAX_LEVELS) {
                const Layer& tl = h->audiodec[pre + k];
                c.coh0 = idx_of(h->Hset[k], -tl.off[0]); c.coh1 = idx_of(h->Hset[k], -tl.off[1]);
                h->cone_done_total[k] += (unsigned)((n_out + c.n_extra) * (Bpad / 16));
                c.done_sig = h->d_sig + LOOP_SIG_LEVEL0 + 16 * k; c.done_val = h->cone_done_val; c.done_count = h->d_cone_count + k; c.done_target = h->cone_done_total[k];
                c.done_stamp = stamp_of(k);
            }
            h->pbegin(PC_DEC);
            launch_cone_fc16(c, g_cur);
            const double K = 3.0 * l.cin;
            h->pend(PC_DEC, ((double)l.N * K + (double)n_out * B * (3.0 * 3.0 * lp.cout + l.N)) * 4.0, 2.0 * n_out * B * l.N * K);
            raw_in = c.H; raw_split = 1; raw_stride = 0;
        } else {
            GemmArgs g{};
            g.X = cone[k]; g.ldx = l.kc; g.Wt = l.Wt; g.ldw = 3 * l.kc; g.bias = l.bias; g.H = raw_gemm; g.ldh = l.Nalloc;
            g.M = n_out * Bpad; g.N = l.N; g.kc = l.kc; g.ntaps = 3; g.mode = 1; g.Bpad = Bpad; g.n_out = n_out; g.j = t;
            g.tab = h->d_tab[k]; g.need = h->d_need[k]; g.stop_after = stop_after; g.t = t;
            g.ksplit = h->opt.cone_ksplit(g.M);
            if (k + 1 >= fc_from && k + 2 < nh) g.ksplit = std::min(g.ksplit, fc_in_split);     // its consumer is a cone_fc16: fewer partials to sum there
            g.split_stride = (long long)g.M * l.Nalloc;
            const int cp = (h->cone_prec && g.M >= 512) ? h->cone_prec : 0;
            g.Wh = cp == 2 ? l.Wh16 : l.Wh; g.Wl = cp == 2 ? l.Wl16 : l.Wl; g.f16 = cp == 2;
            run_gemm(h, g, l.cin, g.Wh ? cp : 0);
            raw_in = raw_gemm; raw_split = g.ksplit; raw_stride = g.split_stride;
        }
        if (k + 1 >= fc_from && k + 2 < nh) continue;       // the next level's cone_fc16 normalises these rows itself
        EpiArgs e{};
        e.nsplit = raw_split; e.split_stride = raw_stride;
        e.H = raw_in; e.ldh = l.Nalloc; e.M = n_out * Bpad; e.C = l.cout; e.mode = PRE_HC;
        e.g1 = l.g1; e.b1 = l.b1; e.g2 = l.g2; e.b2 = l.b2; e.Xres = cone[k]; e.ldres = l.kc; e.restab = h->d_res[k]; e.Bpad = Bpad;
        e.nonorm = !l.ln;
        e.lcc = l.lcc_gate; e.lcc_ids = h->d_spk; e.lcc_T = 0;
        const Layer& nx = h->audiodec[pre + k + 1];
        e.Y = cone[k + 1]; e.ldy = nx.kc; e.ypad = nx.kc; e.stop_after = stop_after; e.t = t;
        level_done(k + 1, e.done_sig, e.done_val, e.done_count, e.done_target, e.coh0, e.coh1);
        if (e.done_sig) e.done_stamp = stamp_of(k + 1);
        run_epi(h, e);
    }
    g_cur = saved;
}

