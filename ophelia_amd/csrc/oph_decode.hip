// libophelia_hip.so -- the decode loop of synth_codedtext2mel (synthesize.py:150-230): per-batch state and 16-utterance tiles, the
// whole-decode launch (dec_chain / dec_loop) with its side-stream cones and streamed SSRN chunks, and the per-step launch paths.
#include "oph_host.h"

// ------------------------------------------------------------------ decoder state
int idx_of(const std::vector<int>& v, int x) {
    auto it = std::lower_bound(v.begin(), v.end(), x);
    return (it != v.end() && *it == x) ? (int)(it - v.begin()) : -1;
}


// Point the handle's working views at tile j of the staged batch: utterances [16 j, 16 j + B_j).
void select_tile(oph_handle* h, int j) {
    const oph_dims& m = h->dm;
    Tile& t = h->tiles[j];
    const size_t r0 = (size_t)j * TILE;
    h->tile = j;
    h->B = std::min(TILE, h->nB - j * TILE); h->Bpad = TILE;
    h->d_L = h->bL[h->txt] + r0 * m.max_N; h->d_ends = h->bEnds[h->txt] + r0; h->d_spk = h->bSpk[h->txt] + r0; h->d_tends = h->bTends + r0;
    h->KV = h->bKV[h->kv_cur] + r0 * m.max_N * 2 * m.d;
    h->Yout = h->bYout[h->buf] + r0 * m.max_T * h->ldy;
    h->Z = h->bZ[h->buf] + r0 * m.max_T * m.r * m.full_dim;
    h->align = h->bAlign + r0 * m.max_N * m.max_T;
    h->d_p = t.d_p; h->d_ctl = t.d_ctl; h->d_ptab = t.d_ptab;
    h->Ytm = t.Ytm; h->Qhist = t.Qhist; h->VW = t.VW; h->QWhist = t.QWhist; h->ae_hist = t.ae_hist;
    h->d_loop_layers = t.d_loop_layers;
}

int ensure_decode_state(oph_handle* h, int B) {
    const int nBpad = round_up(B, TILE);
    if (h->nBpad == nBpad && h->bKV[0]) { h->nB = B; select_tile(h, 0); return ensure_batched_capacity(h, nBpad); }
    if (h->bKV[0]) {
        // a different number of 16-row tiles: release the per-batch state and the workspaces and rebuild them
        for (hipStream_t st : {h->stream, h->sdec, h->scone, h->sssrn, h->scopy}) if (st) hipStreamSynchronize(st);
        h->free_pool(1);
        h->ae_hist.clear(); h->ae_raw.clear(); h->ad_raw.clear(); h->ad_xrow.clear(); h->tiles.clear(); h->loop_proto.clear(); h->loop_lnp.clear();
        h->cone[0].clear(); h->cone[1].clear(); for (int pp = 0; pp < 2; ++pp) { h->coneH[pp].clear(); h->coneL[pp].clear(); } h->d_tab.clear(); h->d_need.clear(); h->d_res.clear(); h->fc_tab.clear(); h->Hset.clear();
        h->bKV[0] = h->bKV[1] = nullptr; h->preenc_valid = false; h->next_staged = false; h->kv_resident = h->y_resident = false;
        h->capB = 0; h->actA = h->actB = h->raw = h->actA2 = h->actB2 = h->raw2 = nullptr;
        h->d_loop_layers = nullptr;
        h->pipelined = false; h->buf = 0; h->ssrn_inflight[0] = h->ssrn_inflight[1] = false;
    }
    const oph_dims& m = h->dm;
    const int d = m.d, Bpad = TILE, ntiles = nBpad / TILE;
    h->nB = B; h->nBpad = nBpad;
    h->ldy = round_up(m.n_mels, 32);
    // ---- batch-level buffers (utterance-major)
    for (int i = 0; i < 2; ++i) {
        h->bL[i] = h->dalloc<int>((size_t)nBpad * m.max_N);
        h->bEnds[i] = h->dalloc<int>(nBpad);
        h->bSpk[i] = h->dalloc<int>(nBpad);
        h->bKV[i] = h->dalloc<float>((size_t)nBpad * m.max_N * 2 * d);
        h->bYout[i] = h->dalloc<float>((size_t)nBpad * m.max_T * h->ldy);
        h->bZ[i] = h->dalloc<float>((size_t)nBpad * m.max_T * m.r * m.full_dim);
    }
    h->txt = 0; h->kv_cur = 0; h->preenc_valid = false; h->next_staged = false;
    h->bTends = h->dalloc<int>(nBpad);
    h->bAlign = h->dalloc<float>((size_t)nBpad * m.max_N * m.max_T);
    h->d_amax = h->dalloc<long long>((size_t)nBpad * m.max_T);
    // ---- scratch shared by the tiles
    h->d_gbuf = h->dalloc<unsigned long long>((size_t)LOOP_MAX_LAYERS * Bpad * RUN_GCOLS);
    h->d_vbuf = h->dalloc<float>((size_t)2 * LOOP_MAX_LAYERS * Bpad * RUN_GCOLS);
    h->run_epoch = 0;
    h->d_clk = h->dalloc<long long>((size_t)2 * 512); h->clk_used = 0;
    if (h->opt.run_stamps) {
        h->d_stamps = h->dalloc<long long>((size_t)2 * 32 * LOOP_MAX_LAYERS * 8);
        h->d_sigdbg = h->dalloc<long long>((size_t)m.max_T * 8);
        h->d_lvldbg = h->dalloc<long long>((size_t)m.max_T * 8);
        if (h->d_lvldbg) hipMemset(h->d_lvldbg, 0, (size_t)m.max_T * 8 * sizeof(long long));
        h->d_cldbg = h->dalloc<long long>((size_t)(2 * m.max_T + 4) * 8 + 512);
    }
    h->Rrow = h->dalloc<float>((size_t)Bpad * 2 * d);
    for (const Layer& l : h->audioenc) h->ae_raw.push_back(h->dalloc<float>((size_t)Bpad * l.Nalloc));
    for (const Layer& l : h->audiodec) {
        h->ad_raw.push_back(h->dalloc<float>((size_t)Bpad * l.Nalloc));
        h->ad_xrow.push_back(l.kind == K_HC ? h->dalloc<float>((size_t)Bpad * l.kc) : nullptr);
    }
    // ---- history cone position sets (offsets back from the current step)
    const int nh = h->n_hc_dec, pre = h->dec_pre;
    std::vector<std::vector<int>> I(nh);
    for (int k = nh - 1; k >= 0; --k) {
        const int r = h->audiodec[pre + k].rate;
        std::vector<int> outs = (k == nh - 1) ? std::vector<int>{0} : I[k + 1];
        std::vector<int> s;
        for (int o : outs) { s.push_back(o); s.push_back(o + r); s.push_back(o + 2 * r); }
        std::sort(s.begin(), s.end());
        s.erase(std::unique(s.begin(), s.end()), s.end());
        I[k] = s;
    }
    h->Hset.assign(nh, {});
    for (int k = 0; k < nh; ++k)
        for (int o : I[k]) if (o >= 1) h->Hset[k].push_back(o);
    h->d_off0 = h->dalloc<int>(h->Hset[0].size());
    hipMemcpyAsync(h->d_off0, h->Hset[0].data(), h->Hset[0].size() * 4, hipMemcpyHostToDevice, h->stream);
    hipStreamSynchronize(h->stream);
    size_t maxrows = h->Hset[0].size();
    for (int k = 0; k < nh; ++k) {
        for (int pp = 0; pp < 2; ++pp) {
            h->cone[pp].push_back(h->dalloc<float>(h->Hset[k].size() * Bpad * (size_t)h->audiodec[pre + k].kc));
            if (h->cone_fused_ok) {
                h->coneH[pp].push_back(h->dalloc<unsigned short>(h->Hset[k].size() * Bpad * (size_t)256));
                h->coneL[pp].push_back(h->dalloc<unsigned short>(h->Hset[k].size() * Bpad * (size_t)256));
            }
        }
        if (k + 1 < nh) {
            // hc layer k evaluated at output offsets Hset[k+1]: taps (oldest first) read Hset[k]
            const int r = h->audiodec[pre + k].rate, n_out = (int)h->Hset[k + 1].size();
            std::vector<int> tab(3 * n_out), need(3 * n_out), res(n_out);
            for (int i = 0; i < n_out; ++i) {
                const int o = h->Hset[k + 1][i];
                for (int t = 0; t < 3; ++t) {
                    const int so = o + (2 - t) * r;
                    tab[t * n_out + i] = idx_of(h->Hset[k], so);
                    need[t * n_out + i] = so;
                    if (tab[t * n_out + i] < 0) { h->fail("internal: cone table hole"); return OPH_ERR_STATE; }
                }
                res[i] = idx_of(h->Hset[k], o);
            }
            int* dt = h->dalloc<int>(tab.size()); int* dn = h->dalloc<int>(need.size()); int* dr = h->dalloc<int>(res.size());
            hipMemcpyAsync(dt, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, h->stream);
            hipMemcpyAsync(dn, need.data(), need.size() * 4, hipMemcpyHostToDevice, h->stream);
            hipMemcpyAsync(dr, res.data(), res.size() * 4, hipMemcpyHostToDevice, h->stream);
            hipStreamSynchronize(h->stream);
            h->d_tab.push_back(dt); h->d_need.push_back(dn); h->d_res.push_back(dr);
            // cone_fc16's tables for layer k (k >= 1): per output its three level-k positions and their level k-1 residual
            // positions; plus the positions of level k that the loop kernel's taps read (offsets r, 2r) but no output has
            // as its current position -- normalised and stored in extra row groups
            FcTables ft{};
            ft.valid = k >= 1 && n_out <= CONE_FC_MAXOUT;
            if (ft.valid) {
                for (int i = 0; i < n_out; ++i)
                    for (int t = 0; t < 3; ++t) {
                        ft.tab[t][i] = (short)tab[t * n_out + i]; ft.need[t][i] = (short)need[t * n_out + i];
                        ft.res[t][i] = (short)idx_of(h->Hset[k - 1], h->Hset[k][tab[t * n_out + i]]);
                        if (ft.res[t][i] < 0) ft.valid = false;
                    }
                for (int o : {r, 2 * r}) {
                    const int ip = idx_of(h->Hset[k], o);
                    bool is_cur = false;
                    for (int i = 0; i < n_out; ++i) is_cur = is_cur || tab[2 * n_out + i] == ip;
                    if (ip >= 0 && !is_cur) {
                        if (ft.n_extra >= CONE_FC_MAXEXTRA) { ft.valid = false; break; }
                        ft.extra[ft.n_extra] = (short)ip; ft.extra_res[ft.n_extra] = (short)idx_of(h->Hset[k - 1], o); ++ft.n_extra;
                    }
                }
            }
            h->fc_tab.push_back(ft);
        }
    }
    const int ld_cat = round_up(d + m.speaker_embedding_size, 32);
    h->coneR = h->dalloc<float>(maxrows * Bpad * 2 * d);
    h->coneRaw = h->dalloc<float>((size_t)CONE_KSPLIT * maxrows * Bpad * (size_t)round_up(2 * d, 128));
    h->coneRawB = h->dalloc<float>((size_t)maxrows * Bpad * (size_t)round_up(2 * d, 128));
    h->coneTmp = h->dalloc<float>(maxrows * Bpad * (size_t)ld_cat);
    h->d_cone_count = h->dalloc<unsigned>(LOOP_MAX_LEVELS); for (uint32_t& v : h->cone_done_total) v = 0;
    if (h->cone_fused_ok) {
        h->hcf_stats_stride = (size_t)((maxrows * Bpad + 63) / 64) * 2 * 2 * 32 * 8 * 2;      // granules per level: [row block][2][2][32 rows][8 tiles][2 values]
        h->d_hcf_stats = h->dalloc<unsigned long long>((size_t)2 * nh * h->hcf_stats_stride);      // two per level: alternating with the step parity
        h->hcf_epoch = 0;
        if (!h->d_hcf_stats) { h->fail("out of device memory for decode state"); return OPH_ERR_DEVICE; }
    }
    // ---- per-tile state
    h->tiles.assign(ntiles, Tile());
    for (Tile& t : h->tiles) {
        t.d_p = h->dalloc<int>(2 * Bpad);
        t.d_ctl = h->dalloc<int>(4);
        t.d_ptab = h->dalloc<int>((size_t)m.max_T * Bpad);
        t.Ytm = h->dalloc<float>((size_t)(m.max_T + 1) * Bpad * h->ldy);
        t.Qhist = h->dalloc<float>((size_t)m.max_T * Bpad * d);
        for (const Layer& l : h->audioenc) t.ae_hist.push_back(l.kind == K_HC ? h->dalloc<float>((size_t)m.max_T * Bpad * l.kc) : nullptr);
        if (h->cone_head_ok) {
            t.VW = h->dalloc<float>((size_t)Bpad * m.max_N * h->ldvw);
            t.QWhist = h->dalloc<float>((size_t)m.max_T * Bpad * d);
        }
        if (!t.d_p || !t.d_ctl || !t.d_ptab || !t.Ytm || !t.Qhist || (h->cone_head_ok && !t.QWhist)) { h->fail("out of device memory for decode state"); return OPH_ERR_DEVICE; }
    }
    hipStreamSynchronize(h->stream);
    if (!h->coneTmp || !h->bZ[1] || !h->bYout[1] || !h->bAlign || !h->d_amax) { h->fail("out of device memory for decode state"); return OPH_ERR_DEVICE; }
    select_tile(h, 0);
    return ensure_batched_capacity(h, nBpad);      // the workspaces of the batched nets, for every batch size of this tile count
}

// reset the CURRENT tile's decode state (synthesize.py:157-166)
void reset_decode(oph_handle* h) {
    const oph_dims& m = h->dm;
    Tile& tl = h->tiles[h->tile];
    tl.steps = 0; tl.ssrn_done = 0; tl.z_copied = 0;
    {
        ResetTileArgs r{};
        const size_t nY = (size_t)h->Bpad * m.max_T * h->ldy, nYtm = (size_t)(m.max_T + 1) * h->Bpad * h->ldy, nAl = (size_t)h->Bpad * m.max_N * m.max_T;
        r.buf[0] = h->Yout; r.n4[0] = nY / 4; r.buf[1] = h->Ytm; r.n4[1] = nYtm / 4; r.buf[2] = h->align; r.n4[2] = nAl / 4;
        r.p = h->d_p; r.n_p = 2 * h->Bpad; r.tends = h->d_tends; r.n_tends = h->Bpad; r.max_T = m.max_T; r.ctl = h->d_ctl;
        if ((nY | nYtm | nAl) & 3) {      // (never with Bpad a multiple of 16; kept exact for any geometry)
            hipMemsetAsync(h->Yout, 0, nY * 4, h->stream); hipMemsetAsync(h->Ytm, 0, nYtm * 4, h->stream); hipMemsetAsync(h->align, 0, nAl * 4, h->stream);
            r.n4[0] = r.n4[1] = r.n4[2] = 0;
        }
        launch_reset_tile(r, h->stream);
    }
    if (h->cone_head_ok) {
        // V . Wc for every text position of the tile (one small GEMM; the cone head adds prob-weighted rows of it)
        GemmArgs g{};
        g.X = h->KV + m.d; g.ldx = 2 * m.d; g.Wt = h->Wt_c; g.ldw = h->kc_c; g.bias = h->d_zeros; g.H = h->VW; g.ldh = h->ldvw;
        g.M = h->B * m.max_N; g.N = m.d; g.kc = h->kc_c; g.ntaps = 1; g.mode = 0; g.T = g.M; g.off[0] = 0;
        hipStream_t saved = g_cur;
        g_cur = h->stream;
        run_gemm(h, g, m.d);
        g_cur = saved;
    }
    // (no host synchronisation: nothing above reads host memory any more, and the decode streams start behind an event of this one)
}
// a new batch starts decoding: in pipelined mode Y / Z ping-pong, so that the SSRN of the previous batch can still read its Y
void begin_batch(oph_handle* h) {
    if (h->pipelined) {
        h->buf ^= 1;
        if (h->ssrn_inflight[h->buf]) { hipStreamWaitEvent(h->stream, h->ev_ssrn_done[h->buf], 0); h->ssrn_inflight[h->buf] = false; }
    } else {
        // speculative SSRN chunks of a previous decode nobody asked the magnitudes of may still be running (and copying into
        // a host buffer): they read the Y / write the Z this batch is about to reuse
        hipStreamSynchronize(h->sssrn);
        hipStreamSynchronize(h->scopy);
    }
    h->chunk_inflight = false;
    h->y_resident = false;
    h->batch_gen++;
}

RowLayer row_layer(const Layer& l) {
    RowLayer r{};
    r.W = l.Wkn; r.ldn = l.ldn; r.bias = l.bias; r.g = l.g1; r.b = l.b1; r.kc = l.kc; r.N = l.cout; r.act = l.act; r.ccat = l.ccat; r.lcc = l.lcc_gate;
    return r;
}
void run_row_chain(oph_handle* h, RowChainArgs& a, int first_is_attn) {
    a.nonorm = (h->dm.flags & OPH_FLAG_NORM_NONE) ? 1 : 0;      // Text2Mel has no transposed convs: all or nothing
    a.nomono = (h->dm.flags & OPH_FLAG_NO_MONOTONIC) ? 1 : 0;
    a.has_lcc = (h->dm.flags & OPH_FLAG_LCC) ? 1 : 0;
    if (h->fixed_att && a.pro == ROW_ATTN) a.ptab = h->d_ptab;
    if (a.has_lcc) a.cat_ids = h->d_spk;
    double wbytes = 0, flops = 0;
    for (int i = 0; i < a.nlayers; ++i) { wbytes += (double)a.L[i].kc * a.L[i].N * 4.0; flops += 2.0 * a.B * a.L[i].kc * a.L[i].N; }
    h->pbegin(PC_ROWCHAIN);
    launch_row_chain(a, g_cur);
    h->pend(PC_ROWCHAIN, wbytes + (double)a.B * 4096.0 + (first_is_attn ? (double)a.B * 8.0 * a.d * 4.0 : 0.0), flops);
}

// dec_layer16 arguments of decoder layer `l` whose input rows x[t] are produced by `prev`'s raw output
void fill_pre(DecArgs& a, const Layer* prev, const float* prev_raw, const float* prev_x) {
    a.nonorm = !prev->ln;
    a.lcc = prev->lcc_gate;
    if (prev->kind == K_CONV) { a.pre = PRE_CONV; a.src = prev_raw; a.ldsrc = prev->Nalloc; a.g1 = prev->g1; a.b1 = prev->b1; a.act = prev->act; a.cin = prev->cout; }
    else { a.pre = PRE_HC; a.src = prev_raw; a.ldsrc = prev->Nalloc; a.g1 = prev->g1; a.b1 = prev->b1; a.g2 = prev->g2; a.b2 = prev->b2; a.xres = prev_x; a.ldres = prev->kc; a.cin = prev->cout; }
}

// one decoder step t.  Critical stream (19 dependent launches):
//   row_chain A  : S[t] -> AudioEnc C_1..C_3 (k=1, row-local LN)            -> x of the first highway layer
//   dec_layer16  : AudioEnc highway layers (column-split, dilated taps from the cached history)
//   row_chain B  : gate of the last highway layer -> attention row t (+ alignments, prev_max, end
//                  detection, stop flag) -> AudioDec C_1 [-> speaker concat -> C_3]
//   dec_layer16  : AudioDec highway layers (taps from the cone of this step)
//   row_chain C  : gate -> AudioDec C_8..C_11 -> LN -> sigmoid -> Y[:, t] (and S[t+1])
// Side stream: cone(t+1), released by the event recorded right after row_chain B of step t.
// ---------------------------------------------------------------- persistent runs (oph_decrun.hip)
// Prologue description of the layer that consumes `prev`'s raw output.
RunLayer run_layer(const Layer& l, const Layer* prev) {
    RunLayer r{};
    if (!prev) r.pre = RUN_COPY;
    else if (prev->kind == K_CONV) { r.pre = RUN_CONV; r.act = prev->act; }
    else r.pre = RUN_HC;
    if (prev) { r.cin = prev->cout; r.nonorm = !prev->ln; r.g1 = prev->g1; r.b1 = prev->b1; r.g2 = prev->g2; r.b2 = prev->b2; }
    r.ccat = l.ccat; r.cat_table = l.cat_table;
    r.ntaps = l.ntaps; r.kc = l.kc; r.N = l.N; r.Wt = l.Wt; r.ldw = l.ntaps * l.kc; r.bias = l.bias;
    return r;
}
void run_args_common(oph_handle* h, RunArgs& a, int t, int stop_mode) {
    const oph_dims& m = h->dm;
    a.B = h->B; a.Bpad = h->Bpad; a.t = t; a.stop_after = h->d_ctl + 1;
    const bool ms = m.flags & (OPH_FLAG_SPK_AUDIO_DECODER_INPUT | OPH_FLAG_SPK_AUDIO_ENCODER_INPUT);
    a.spk_ids = ms ? h->d_spk : nullptr;
    a.gbuf = h->d_gbuf; a.err = h->d_ctl + 2;
    a.KV = h->KV; a.N_keys = m.max_N; a.win = m.attention_win_size; a.max_T = m.max_T;
    a.pcur = h->d_p + (t & 1) * h->Bpad; a.pnext = h->d_p + ((t + 1) & 1) * h->Bpad;
    a.ends = h->d_ends; a.t_ends = h->d_tends; a.n_ended = h->d_ctl; a.stop_flag = h->d_ctl + 1; a.stop_mode = stop_mode;
    a.Qhist = h->Qhist; a.align = h->align;
    a.Yout = h->Yout; a.ldy = h->ldy; a.Ytm = h->Ytm; a.ldtm = h->ldy;
}
void run_launch(oph_handle* h, RunArgs& a) {
    int slices = 1, kmax = 32;
    double bytes = 0, flops = 0;
    for (int i = 0; i < a.nlayers; ++i) {
        const RunLayer& L = a.L[i];
        slices = std::max(slices, round_up(L.N, 16) / 16);
        kmax = std::max(kmax, L.ntaps * L.kc);
        const double K = (double)L.ntaps * L.kc;
        bytes += ((double)L.N * K + (double)a.B * (K + L.N)) * 4.0;
        flops += 2.0 * a.B * L.N * K;
    }
    a.epoch0 = h->run_epoch;
    h->run_epoch += RUN_MAX_LAYERS;
    h->pbegin(PC_DECRUN);
    launch_dec_run(a, slices, 4, kmax, g_cur);
    h->pend(PC_DECRUN, bytes, flops);
}
// First launch of step t: S[t] -> AudioEnc (k=1 head, highway layers with cached dilated taps) -> attention row t
// (+ alignments, prev_max, end detection) -> AudioDec input convs.  Leaves the raw rows of the last input conv.
void run_encoder_half(oph_handle* h, int t, int stop_mode) {
    const oph_dims& m = h->dm;
    const int Bpad = h->Bpad, pre = h->dec_pre;
    RunArgs a{};
    run_args_common(h, a, t, stop_mode);
    int n = 0;
    for (size_t li = 0; li < h->audioenc.size(); ++li) {
        const Layer& l = h->audioenc[li];
        RunLayer r = run_layer(l, li ? &h->audioenc[li - 1] : nullptr);
        if (li == 0) { r.src = h->Ytm + (size_t)t * Bpad * h->ldy; r.ldsrc = h->ldy; r.cin = m.n_mels; }
        if (l.kind == K_HC) {
            float* hist = h->ae_hist[li];
            const int o0 = -l.off[0], o1 = -l.off[1];
            r.tap0 = t - o0 >= 0 ? hist + (size_t)(t - o0) * Bpad * l.kc : nullptr;
            r.tap1 = t - o1 >= 0 ? hist + (size_t)(t - o1) * Bpad * l.kc : nullptr;
            r.ldtap = l.kc;
            r.xstore = hist + (size_t)t * Bpad * l.kc; r.ldstore = l.kc;
        }
        a.L[n++] = r;
    }
    for (int k = 0; k < pre; ++k) {
        const Layer& l = h->audiodec[k];
        RunLayer r = run_layer(l, k ? &h->audiodec[k - 1] : &h->audioenc.back());
        if (k == 0) r.pre = RUN_ATTN;
        if (l.ccat > 0) r.cat_table = h->emb_spk;      // 'audio_decoder_input' (networks.py:381-389)
        if (k + 1 == pre) { r.out = h->ad_raw[k]; r.ldout = l.Nalloc; }
        a.L[n++] = r;
    }
    a.nlayers = n;
    if (h->d_stamps && t == m.max_T / 2) a.stamps = h->d_stamps;
    run_launch(h, a);
}
// Second launch of step t: AudioDec highway layers (older taps from the cone of this step) -> k=1 tail -> mel frame t.
void run_decoder_half(oph_handle* h, int t, int stop_mode) {
    const int Bpad = h->Bpad, pre = h->dec_pre, nh = h->n_hc_dec;
    const std::vector<float*>& cone = h->cone[t & 1];
    RunArgs a{};
    run_args_common(h, a, t, stop_mode);
    int n = 0;
    for (size_t li = pre; li < h->audiodec.size(); ++li) {
        const Layer& l = h->audiodec[li];
        RunLayer r = run_layer(l, &h->audiodec[li - 1]);
        if ((int)li == pre) { r.src = h->ad_raw[pre - 1]; r.ldsrc = h->audiodec[pre - 1].Nalloc; }
        const int k = (int)li - pre;
        if (k < nh) {
            const int o0 = -l.off[0], o1 = -l.off[1];
            r.tap0 = t - o0 >= 0 ? cone[k] + (size_t)idx_of(h->Hset[k], o0) * Bpad * l.kc : nullptr;
            r.tap1 = t - o1 >= 0 ? cone[k] + (size_t)idx_of(h->Hset[k], o1) * Bpad * l.kc : nullptr;
            r.ldtap = l.kc;
        }
        a.L[n++] = r;
    }
    RunLayer e = run_layer(h->audiodec.back(), &h->audiodec.back());     // prologue only: LN + squash sigmoid of the last conv
    e.act = ACT_SIGMOID;                                                 // squash_output_t2m (networks.py:430-431)
    e.N = 0; e.ccat = 0; e.cat_table = nullptr;
    a.L[n++] = e;
    a.nlayers = n;
    if (h->d_stamps && t == h->dm.max_T / 2) a.stamps = h->d_stamps + (size_t)32 * RUN_MAX_LAYERS * 8;
    run_launch(h, a);
}
// whether this handle's configuration can take the persistent-run path
bool run_supported(const oph_handle* h) {
    const oph_dims& m = h->dm;
    if (h->opt.decode == 2) return false;
    if (h->n_hc_dec > LOOP_MAX_LEVELS || h->n_hc_dec >= 15) return false;      // one completion word per cone level; 4-bit level fields in the packed descriptors
    if (m.flags & (OPH_FLAG_LCC | OPH_FLAG_NO_MONOTONIC | OPH_FLAG_NO_SQUASH_T2M)) return false;       // variants served by the per-layer kernels
    if ((int)h->audioenc.size() + h->dec_pre > RUN_MAX_LAYERS || (int)h->audiodec.size() - h->dec_pre + 1 > RUN_MAX_LAYERS) return false;
    for (const auto* net : {&h->audioenc, &h->audiodec})
        for (const Layer& l : *net) {
            if (l.ntaps * l.kc > 768 || l.N > RUN_GCOLS || (l.ntaps == 3 && l.kc > 256) || l.cout > 256) return false;
        }
    return true;
}

// ---------------------------------------------------------------- whole-decode launch (dec_loop / dec_chain)
constexpr int CLK_SLOTS = 512;
// read the finished launches' clock pairs back (the launches must be complete) and add them to the running totals
void drain_loop_clock(oph_handle* h) {
    if (!h->d_clk || h->clk_used == 0) return;
    std::vector<long long> v((size_t)2 * h->clk_used);
    if (hipMemcpy(v.data(), h->d_clk, v.size() * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess)
        for (int k = 0; k < h->clk_used; ++k) {
            const unsigned long long t0 = (unsigned long long)v[2 * k], t1 = (unsigned long long)v[2 * k + 1];
            if (t1 > t0 && t0 != ~0ull) { h->clk_launches++; h->clk_total_us += (double)(t1 - t0) * 0.01; }      // 100 MHz constant clock
        }
    (void)hipGetLastError();
    h->clk_used = 0;
}

// Static layer table of a decode: AudioEnc (layer 0 consumes the previous step's last AudioDec layer) -> attention +
// AudioDec input convs -> AudioDec highway layers (taps from the cone ping-pong buffers) -> k=1 tail.  Built once per
// decode state: the weights in the loop kernel's fragment order and the prologue's LayerNorm parameters are shared by the
// tiles; build_loop_layers() then packs one descriptor table per tile (the history pointers differ).
int build_loop_proto(oph_handle* h) {
    const int pre = h->dec_pre, nh = h->n_hc_dec;
    std::vector<LoopLayer> v;
    auto from = [&](const Layer& l, const Layer* prev) {
        const RunLayer r = run_layer(l, prev);
        LoopLayer q{};
        q.pre = r.pre; q.act = r.act; q.cin = r.cin; q.nonorm = r.nonorm; q.g1 = r.g1; q.b1 = r.b1; q.g2 = r.g2; q.b2 = r.b2;
        q.cat_table = r.cat_table; q.ccat = r.ccat; q.ntaps = r.ntaps; q.kc = r.kc; q.N = r.N; q.Wt = r.Wt; q.ldw = r.ldw; q.bias = r.bias;
        return q;
    };
    for (size_t li = 0; li < h->audioenc.size(); ++li) {
        const Layer& l = h->audioenc[li];
        LoopLayer q = from(l, li ? &h->audioenc[li - 1] : &h->audiodec.back());
        if (li == 0) q.act = ACT_SIGMOID;                 // squash_output_t2m (networks.py:430-431): x = mel frame t-1
        if (l.kind == K_HC) { q.tapkind = 1; q.off0 = -l.off[0]; q.off1 = -l.off[1]; q.idx0 = (int)li; }     // idx0: which history (per tile)
        v.push_back(q);
    }
    h->loop_attn = (int)v.size();
    for (size_t li = 0; li < h->audiodec.size(); ++li) {
        const Layer& l = h->audiodec[li];
        LoopLayer q = from(l, li ? &h->audiodec[li - 1] : &h->audioenc.back());
        if (li == 0) q.pre = RUN_ATTN;
        if (l.ccat > 0) q.cat_table = h->emb_spk;
        const int k = (int)li - pre;
        if (k >= 0 && k < nh) {
            q.tapkind = 2; q.off0 = -l.off[0]; q.off1 = -l.off[1]; q.level1 = k + 1;
            q.idx0 = idx_of(h->Hset[k], q.off0); q.idx1 = idx_of(h->Hset[k], q.off1);
            q.cone0 = h->cone[0][k]; q.cone1 = h->cone[1][k];
            if (q.idx0 < 0 || q.idx1 < 0) { h->fail("internal: cone tap not in the position set"); return OPH_ERR_STATE; }
            if (q.level1 > LOOP_MAX_LEVELS || q.level1 > 15) { h->fail("internal: too many cone levels for the loop kernel"); return OPH_ERR_STATE; }
        }
        v.push_back(q);
    }
    if ((int)v.size() > LOOP_MAX_LAYERS) { h->fail("internal: too many decoder layers for the loop kernel"); return OPH_ERR_STATE; }
    h->loop_nlayers = (int)v.size();
    h->loop_slices = 1; h->loop_kmax = 32;
    for (const LoopLayer& q : v) { h->loop_slices = std::max(h->loop_slices, round_up(q.N, 16) / 16); h->loop_kmax = std::max(h->loop_kmax, q.ntaps * q.kc); }
    h->loop_rows = h->opt.run_rows;
    const int R = h->loop_rows, PF = (768 / 16 + R - 1) / R;       // as dec_loop<R> (RUN_KMAX = 768)
    if (h->loop_kmax > 768) { h->fail("internal: layer K exceeds the loop kernel's"); return OPH_ERR_STATE; }
    h->loop_lnp.assign(v.size(), nullptr);
    for (size_t i = 0; i < v.size(); ++i) {
        LoopLayer& q = v[i];
        // The weights in the order the loop kernel's lanes hold them: [column slice g][wave w][chunk i][lane][4] with
        // chunk = w + R i, column = 16 g + 4 (lane >> 4) + (lane & 3), k = 16 chunk + 4 ((lane >> 2) & 3) + e -- one
        // fragment request of a wave is 1 KB contiguous (8 full lines) instead of 16 half lines 3 KB apart: the CU's
        // address unit was the bottleneck of the weight prefetch (profiles/r02 ablation: 0.9 us of a 5.3 us layer).
        {
            // (packed on the device, like every other layout: oph_pack.hip)
            const int slices = round_up(q.N, 16) / 16, nch = (q.ntaps * q.kc) / 16;
            const int rows_have = std::min(slices * 16, round_up(q.N, 16));
            float* dsw = h->dalloc<float>((size_t)slices * R * PF * 64 * 4);
            if (!dsw) { h->fail("out of device memory"); return OPH_ERR_DEVICE; }
            launch_pack_loop(q.Wt, q.ldw, rows_have, nch, slices, R, PF, dsw, h->stream);
            q.Wt = dsw;
        }
        if (q.g1) {       // the prologue's LayerNorm parameters side by side: one pointer instead of four
            const int ls = round_up(std::max(q.cin, 4), 4);
            float* lnp = h->dalloc<float>((size_t)4 * ls);
            if (!lnp) { h->fail("out of device memory"); return OPH_ERR_DEVICE; }
            hipMemset(lnp, 0, (size_t)4 * ls * sizeof(float));
            const float* src[4] = {q.g1, q.b1, q.g2, q.b2};
            for (int k = 0; k < 4; ++k)
                if (src[k] && hipMemcpy(lnp + (size_t)k * ls, src[k], (size_t)q.cin * sizeof(float), hipMemcpyDeviceToDevice) != hipSuccess) { h->fail("descriptor upload failed"); return OPH_ERR_DEVICE; }
            h->loop_lnp[i] = lnp;
        }
    }
    if (hipStreamSynchronize(h->stream) != hipSuccess) { h->fail("weight repack failed"); return OPH_ERR_DEVICE; }     // (the packed copies are read from other streams)
    h->loop_proto = v;
    // dec_chain (oph_decchain.hip) is dec_loop specialised for the standard geometry: 256 channels per row, 8 rows per workgroup,
    // LayerNorm everywhere, a window of <= 4 keys, the attention layer emitting QW, k = 3 layers 3 x 256 wide, k = 1 layers <= 512 wide
    {
        const oph_dims& m = h->dm;
        bool ok = !h->opt.no_chain && R == 8 && m.d == 256 && m.attention_win_size <= 4 && m.n_mels <= 256 && !(m.flags & OPH_FLAG_NORM_NONE) &&
                  h->cone_head_ok && !h->opt.no_loop_qw && h->audiodec[0].cin == 2 * m.d;
        for (size_t i = 0; ok && i < v.size(); ++i) {
            const LoopLayer& q = v[i];
            ok = !q.nonorm && q.g1 != nullptr && (i == 0 || q.cin == 256) && (q.ntaps == 3 ? (q.kc == 256 && q.ccat == 0 && q.pre != RUN_ATTN) : (q.ntaps == 1 && q.kc <= 512 && q.kc % 16 == 0)) &&
                 (q.pre == RUN_CONV || q.pre == RUN_HC || q.pre == RUN_ATTN) && (q.pre != RUN_ATTN || (q.kc == 512 && q.ccat == 0)) && (q.ccat == 0 || (q.cat_table != nullptr && q.kc >= 256 + q.ccat)) &&
                 (i != 0 || (q.pre == RUN_CONV && q.ccat == 0 && q.ntaps == 1));
        }
        h->chain_ok = ok;
    }
    return OPH_OK;
}
// descriptor table of the CURRENT tile
int build_loop_layers(oph_handle* h) {
    if (h->loop_proto.empty()) { const int rc = build_loop_proto(h); if (rc) return rc; }
    const std::vector<LoopLayer>& v = h->loop_proto;
    std::vector<unsigned> words(v.size() * LOOP_DESC_STRIDE, 0u);
    for (size_t i = 0; i < v.size(); ++i) {
        LoopLayer q = v[i];
        if (q.tapkind == 1) { q.hist = h->ae_hist[q.idx0]; q.idx0 = 0; }
        unsigned* w = &words[i * LOOP_DESC_STRIDE];
        const int ls = round_up(std::max(q.cin, 4), 4);
        auto put = [&](int at, const void* ptr) { const uint64_t u = (uint64_t)(uintptr_t)ptr; w[at] = (unsigned)u; w[at + 1] = (unsigned)(u >> 32); };
        put(0, q.Wt); put(2, q.bias); put(4, h->loop_lnp[i]); put(6, q.cat_table); put(8, q.hist); put(10, q.cone0); put(12, q.cone1);
        const LoopLayer& nx = v[(i + 1) % v.size()];
        if (q.cin > 0xffff || q.kc > 0xffff || q.N > 0xffff || q.ldw > 0xffff || q.ccat > 0xffff || q.off0 > 0xffff || q.off1 > 0xffff || q.idx0 > 0xffff || q.idx1 > 0xffff || q.off0 < 0 || q.off1 < 0 ||
            q.level1 > 15 || nx.level1 > 15 || q.pre > 15 || q.act > 15 || q.ntaps > 3 || q.tapkind > 3) {
            h->fail("internal: layer geometry does not fit the packed descriptor"); return OPH_ERR_STATE;
        }
        w[14] = (unsigned)q.pre | (unsigned)q.act << 4 | (unsigned)(q.nonorm ? 1 : 0) << 8 | (unsigned)q.ntaps << 12 | (unsigned)q.tapkind << 16 | (unsigned)nx.pre << 20 |
                (unsigned)q.level1 << 24 | (unsigned)nx.level1 << 28;
        w[15] = (unsigned)q.cin | (unsigned)q.kc << 16;
        w[16] = (unsigned)q.N | (unsigned)q.ldw << 16;
        w[17] = (unsigned)q.ccat | (unsigned)ls << 16;
        w[18] = (unsigned)q.off0 | (unsigned)q.off1 << 16;
        w[19] = (unsigned)q.idx0 | (unsigned)q.idx1 << 16;
    }
    unsigned* dl = h->dalloc<unsigned>(words.size());
    if (!dl) { h->fail("out of device memory"); return OPH_ERR_DEVICE; }
    if (hipMemcpy(dl, words.data(), words.size() * sizeof(unsigned), hipMemcpyHostToDevice) != hipSuccess) { h->fail("descriptor upload failed"); return OPH_ERR_DEVICE; }
    h->tiles[h->tile].d_loop_layers = h->d_loop_layers = dl;
    return OPH_OK;
}

// diagnostics (OPH_TRACE) at the moment a decode's wait timed out: the level words, the levels' arrival counters against their
// targets, the tile's control words
static void trace_cone_state(oph_handle* h) {
    if (!g_trace || !h->d_sig || !h->d_cone_count) return;
    unsigned sg[LOOP_SIG_LEVEL0 + 16 * 8], cc[LOOP_MAX_LEVELS];
    int ctl[4] = {0, 0, 0, 0};
    hipMemcpy(sg, h->d_sig, sizeof sg, hipMemcpyDeviceToHost); hipMemcpy(cc, h->d_cone_count, sizeof cc, hipMemcpyDeviceToHost);
    hipMemcpy(ctl, h->d_ctl, sizeof ctl, hipMemcpyDeviceToHost);
    (void)hipGetLastError();
    TRACE("  sig_base %u: attention word %u; level words %u %u %u %u %u %u; arrivals %u/%u %u/%u %u/%u %u/%u %u/%u %u/%u; ctl %d %d %d %d", h->sig_base, sg[0],
          sg[LOOP_SIG_LEVEL0], sg[LOOP_SIG_LEVEL0 + 16], sg[LOOP_SIG_LEVEL0 + 32], sg[LOOP_SIG_LEVEL0 + 48], sg[LOOP_SIG_LEVEL0 + 64], sg[LOOP_SIG_LEVEL0 + 80],
          cc[0], h->cone_done_total[0], cc[1], h->cone_done_total[1], cc[2], h->cone_done_total[2], cc[3], h->cone_done_total[3], cc[4], h->cone_done_total[4], cc[5], h->cone_done_total[5],
          ctl[0], ctl[1], ctl[2], ctl[3]);
}

// The decode loop as ONE launch on the critical stream + the per-step cones on the side stream, chained by device words
// (in-kernel waits and signals on both ends).  Steps [t_begin, t_end); t_begin > 0 continues a decode of this tile.
int decode_loop(oph_handle* h, int t_begin, int t_end, int stop_mode) {
    const oph_dims& m = h->dm;
    if (!h->d_loop_layers) { const int rc = build_loop_layers(h); if (rc) return rc; }
    const int lookahead = h->opt.lookahead;
    h->host_prog[0] = t_begin - 1; h->host_prog[1] = INT_MAX;
    if ((uint64_t)h->run_epoch + (uint64_t)(m.max_T + 1) * LOOP_MAX_LAYERS > 0xF0000000ull) {     // tag wrap guard
        for (hipStream_t st : {h->sdec, h->scone}) hipStreamSynchronize(st);
        hipMemsetAsync(h->d_gbuf, 0, (size_t)LOOP_MAX_LAYERS * h->Bpad * RUN_GCOLS * 8, h->sdec);
        hipStreamSynchronize(h->sdec);
        h->run_epoch = 0;
    }
    LoopArgs a{};
    a.nlayers = h->loop_nlayers; a.B = h->B; a.Bpad = h->Bpad; a.t_begin = t_begin; a.t_end = t_end; a.stop_mode = stop_mode; a.attn_layer = h->loop_attn;
    a.L = h->d_loop_layers; a.ctl = h->d_ctl;
    const bool ms = m.flags & (OPH_FLAG_SPK_AUDIO_DECODER_INPUT | OPH_FLAG_SPK_AUDIO_ENCODER_INPUT);
    a.spk_ids = ms ? h->d_spk : nullptr;
    a.gbuf = h->d_gbuf; a.epoch0 = h->run_epoch;
    h->run_epoch += (uint32_t)(m.max_T + 1) * LOOP_MAX_LAYERS;
    a.stamps = h->d_stamps; a.stamp_t = m.max_T / 2;
    a.KV = h->KV; a.N_keys = m.max_N; a.win = m.attention_win_size; a.max_T = m.max_T;
    a.p = h->d_p; a.ends = h->d_ends; a.t_ends = h->d_tends;
    a.Qhist = h->Qhist; a.align = h->align; a.Yout = h->Yout; a.ldy = h->ldy; a.Ytm = h->Ytm;
    a.sig = h->d_sig; a.sig_base = h->sig_base;
    h->qw_from_loop = !h->opt.no_loop_qw && h->cone_head_ok && !h->fixed_att && h->QWhist != nullptr && h->audiodec[0].cin == 2 * m.d && (m.d % 16) == 0;
    a.QW = h->qw_from_loop ? h->QWhist : nullptr; a.attn_slices = round_up(h->audiodec[0].N, 16) / 16;
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, (void*)h->host_prog, 0) != hipSuccess) { h->fail("pinned progress words are not mapped"); return OPH_ERR_DEVICE; }
    a.host_progress = (volatile int*)dp;
    const int dbg = h->opt.loop_dbg;
    a.dbg = dbg;
    a.sigdbg = h->d_sigdbg;
    hipStreamWaitEvent(h->scone, h->ev_in, 0);
    g_cur = h->sdec;
    double bytes = 0, flops = 0;
    for (const auto* net : {&h->audioenc, &h->audiodec})
        for (const Layer& l : *net) { const double K = (double)l.ntaps * l.cin; bytes += ((double)l.N * K + (double)h->B * (K + l.N)) * 4.0; flops += 2.0 * h->B * l.N * K; }
    if (h->d_sigdbg) hipMemsetAsync(h->d_sigdbg, 0, (size_t)m.max_T * 8 * sizeof(long long), h->sdec);
    h->pbegin(PC_DECLOOP);
    h->dec_t0 = std::chrono::steady_clock::now(); h->chunk_inflight = false; h->dec_tbegin = t_begin; h->dec_tend = t_end;
    if (h->d_clk) {        // device-side witness: first workgroup in / last workgroup out of this launch, on the kernel's own clock
        if (h->clk_used == CLK_SLOTS) { hipStreamSynchronize(h->sdec); drain_loop_clock(h); }
        static const long long clk_init[2] = {-1LL, 0LL};
        a.clk = h->d_clk + 2 * h->clk_used++;
        hipMemcpyAsync(a.clk, clk_init, sizeof clk_init, hipMemcpyHostToDevice, h->sdec);
    }
    // the generic kernel when stamps or ablation bits other than "no side stream" are asked for (they live there)
    const bool use_chain = h->chain_ok && !h->fixed_att && a.QW != nullptr && (dbg & ~32) == 0 && h->d_vbuf;
    if (use_chain) {
        // every hand-off slot starts as the sentinel (what the previous launch left in them is stale)
        a.vbuf = h->d_vbuf;
        hipMemsetAsync(h->d_vbuf, 0xFF, (size_t)2 * LOOP_MAX_LAYERS * h->Bpad * RUN_GCOLS * sizeof(float), h->sdec);
        launch_dec_chain(a, h->loop_slices, h->sdec);
    }
    else launch_dec_loop(a, h->loop_slices, h->loop_rows, h->loop_kmax, h->sdec);
    h->pend(PC_DECLOOP, bytes * (t_end - t_begin), flops * (t_end - t_begin));
    if (h->want_preenc && h->next_staged && !h->preenc_valid) {
        // K,V of the NEXT batch's staged text into the other KV buffer, on the SSRN partition (own workspace; in stream order
        // behind the previous batch's SSRN and ahead of this batch's chunks)
        if (run_encode_into(h, h->bL[h->txt ^ 1], h->bSpk[h->txt ^ 1], h->next_B, h->bKV[h->kv_cur ^ 1], h->sssrn, 1) == OPH_OK &&
            hipEventRecord(h->ev_preenc, h->sssrn) == hipSuccess)
            h->preenc_valid = true;
    }
    // side stream: cone(t) after the attention of step t-1; its launches wait for / raise the device words themselves
    g_cur = h->scone;
    const auto t_host0 = std::chrono::steady_clock::now();
    const bool stream_ssrn = h->spec_ssrn && h->opt.ssrn_chunk > 0 && !h->opt.no_stream_ssrn;
    for (int t = std::max(1, t_begin); t < t_end && !(dbg & 32); ++t) {
        // bounded run-ahead, so that an early stop leaves at most `lookahead` queued cones (they early-out on the device)
        auto t_wait0 = std::chrono::steady_clock::now();
        while (h->host_prog[0] < t - 1 - lookahead && h->host_prog[1] == INT_MAX) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_wait0).count() > 10.0) {
                h->fail("decode loop kernel made no progress for 10 s (step %d)", h->host_prog[0]);
                return OPH_ERR_DEVICE;
            }
            // `lookahead` cones are queued in front of the device (~0.8 ms of work): sleep instead of spinning, so that a rank does not
            // burn a whole host core on this wait (8 ranks share a node's cores)
            struct timespec ts = {0, 30000};
            nanosleep(&ts, nullptr);
        }
        const int stopped_at = h->host_prog[1];
        if (stopped_at != INT_MAX && t > stopped_at + 1) break;       // step stop+1 still runs (stores off) and polls its cone
        if (!h->opt.skip_cone) {
            // (the first cone of a continued decode has nobody to wait for: the attention of step t_begin - 1 is long done)
            h->cone_inline_sig = true; h->cone_wait_val = (t == t_begin) ? 0u : h->sig_base + (uint32_t)t; h->cone_done_val = h->sig_base + (uint32_t)t;
            launch_cone(h, t);
            h->cone_inline_sig = false;
        }
        // SSRN over the mel frames that are final: the attention of step p is done => frames < p are stored (write-through)
        if (stream_ssrn && stopped_at == INT_MAX) { const int rc = ssrn_stream_chunks(h, h->host_prog[0], false); if (rc) return rc; g_cur = h->scone; }
    }
    if (g_trace) {
        const double enq = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_host0).count() * 1e3;
        hipStreamSynchronize(h->sdec); hipStreamSynchronize(h->scone);
        const double all = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_host0).count() * 1e3;
        TRACE("decode loop (one launch): side-stream enqueue %.2f ms, drained %.2f ms after the launch", enq, all);
    }
    g_cur = h->sdec;
    return OPH_OK;
}

void decode_step(oph_handle* h, int t, int t_last, int stop_mode) {
    const oph_dims& m = h->dm;
    const int d = m.d, Bpad = h->Bpad, B = h->B;
    int* stop_after = h->d_ctl + 1;
    g_cur = h->sdec;
    const bool run = h->use_run && !h->fixed_att;
    const int pre = h->dec_pre, nh = h->n_hc_dec;
    if (run) {
        run_encoder_half(h, t, stop_mode);
    } else {
    // ---------------- row_chain A: AudioEnc k=1 head
    size_t nk1 = 0;
    while (nk1 < h->audioenc.size() && h->audioenc[nk1].kind == K_CONV) ++nk1;
    {
        const Layer& hc0 = h->audioenc[nk1];
        RowChainArgs a{};
        a.pro = ROW_COPY; a.src = h->Ytm + (size_t)t * Bpad * h->ldy; a.ldsrc = h->ldy; a.cin = m.n_mels;
        a.nlayers = (int)nk1;
        for (size_t i = 0; i < nk1; ++i) {
            a.L[i] = row_layer(h->audioenc[i]);
            if (h->audioenc[i].cat_table) { a.cat_table = h->audioenc[i].cat_table; a.cat_ids = h->d_spk; }   // 'audio_encoder_input'
        }
        a.xout = h->ae_hist[nk1] + (size_t)t * Bpad * hc0.kc; a.ldout = hc0.kc;
        a.B = B; a.Bpad = Bpad; a.stop_after = stop_after; a.t = t; a.d = d;
        run_row_chain(h, a, 0);
    }
    // ---------------- AudioEnc highway layers, incremental (causal, mask-free => cacheable)
    const Layer* prev = nullptr;
    const float* prev_raw = nullptr;
    const float* prev_x = nullptr;    // previous layer's input rows at time t (highway residual)
    h->gbegin(PC_DEC);
    for (size_t li = nk1; li < h->audioenc.size(); ++li) {
        const Layer& l = h->audioenc[li];
        float* hist = h->ae_hist[li];
        DecArgs a{};
        if (li == nk1) { a.pre = PRE_COPY; a.src = hist + (size_t)t * Bpad * l.kc; a.ldsrc = l.kc; a.cin = l.cin; }
        else { fill_pre(a, prev, prev_raw, prev_x); a.lcc_ids = h->d_spk; a.xstore = hist + (size_t)t * Bpad * l.kc; a.ldstore = l.kc; }
        a.ntaps = l.ntaps; a.kc = l.kc; a.ldtap = l.kc;
        const int o0 = -l.off[0], o1 = -l.off[1];
        a.tap0 = t - o0 >= 0 ? hist + (size_t)(t - o0) * Bpad * l.kc : nullptr;
        a.tap1 = t - o1 >= 0 ? hist + (size_t)(t - o1) * Bpad * l.kc : nullptr;
        a.Wt = l.Wt; a.ldw = l.ntaps * l.kc; a.bias = l.bias; a.H = h->ae_raw[li]; a.ldh = l.Nalloc; a.B = B;
        a.stop_after = stop_after; a.t = t;
        run_dec(h, a, l);
        prev = &l; prev_raw = h->ae_raw[li];
        prev_x = hist + (size_t)t * Bpad * l.kc;
    }
    h->gend(PC_DEC);
    // ---------------- row_chain B: attention at row t + AudioDec k=1 head
    {
        const Layer& hca = h->audiodec[pre];
        RowChainArgs a{};
        a.pro = ROW_ATTN; a.src = prev_raw; a.ldsrc = prev->Nalloc; a.cin = d;
        a.g1 = prev->g1; a.b1 = prev->b1; a.g2 = prev->g2; a.b2 = prev->b2; a.xres = prev_x; a.ldres = prev->kc;
        a.lcc_pro = prev->lcc_gate;
        a.KV = h->KV; a.N_keys = m.max_N; a.d = d; a.win = m.attention_win_size; a.max_T = m.max_T;
        a.pcur = h->d_p + (t & 1) * Bpad; a.pnext = h->d_p + ((t + 1) & 1) * Bpad;
        a.ends = h->d_ends; a.t_ends = h->d_tends; a.n_ended = h->d_ctl; a.stop_flag = stop_after; a.stop_mode = stop_mode;
        a.Qhist = h->Qhist; a.align = h->align; a.Bpad = Bpad;
        a.nlayers = pre;
        for (int i = 0; i < pre; ++i) a.L[i] = row_layer(h->audiodec[i]);
        a.cat_table = h->emb_spk; a.cat_ids = h->d_spk;
        a.xout = h->ad_xrow[pre]; a.ldout = hca.kc;
        a.B = B; a.stop_after = stop_after; a.t = t;
        run_row_chain(h, a, 1);
    }
    }   // !run
    // cone(t) (launched during step t-1, or by decode_range for the first step) must have landed
    const bool sv = h->use_sigval;
    {
        HostTimer ht(0, g_trace);
        if (t >= 1) {
            if (sv) hipStreamWaitValue32(h->sdec, h->d_sig + 16, h->sig_base + (uint32_t)t, hipStreamWaitValueGte, 0xffffffffu);
            else hipStreamWaitEvent(h->sdec, h->ev_cone, 0);
        }
    }
    // release cone(t+1) on the side stream: needs p_{t+1} and Q[t], both written by row_chain B of step t
    if (t + 1 < t_last) {
        {
            HostTimer ht(0, g_trace);
            if (sv) {
                hipStreamWriteValue32(h->sdec, h->d_sig, h->sig_base + (uint32_t)t + 1, 0);
                hipStreamWaitValue32(h->scone, h->d_sig, h->sig_base + (uint32_t)t + 1, hipStreamWaitValueGte, 0xffffffffu);
            } else {
                hipEventRecord(h->ev_attn, h->sdec);
                hipStreamWaitEvent(h->scone, h->ev_attn, 0);
            }
        }
        if (!h->opt.skip_cone) { HostTimer ht(1, g_trace); launch_cone(h, t + 1); }
        {
            HostTimer ht(0, g_trace);
            if (sv) hipStreamWriteValue32(h->scone, h->d_sig + 16, h->sig_base + (uint32_t)t + 1, 0);
            else hipEventRecord(h->ev_cone, h->scone);
        }
    }
    if (run) { run_decoder_half(h, t, stop_mode); return; }
    const std::vector<float*>& cone = h->cone[t & 1];
    // ---------------- AudioDec highway layers, row t (taps from the cone)
    const Layer* prev = nullptr; const float* prev_raw = nullptr; const float* prev_x = nullptr;
    h->gbegin(PC_DEC);
    for (int k = 0; k < nh; ++k) {
        const size_t li = pre + k;
        const Layer& l = h->audiodec[li];
        DecArgs a{};
        if (k == 0) { a.pre = PRE_COPY; a.src = h->ad_xrow[li]; a.ldsrc = l.kc; a.cin = l.cin; }
        else { fill_pre(a, prev, prev_raw, prev_x); a.lcc_ids = h->d_spk; a.xstore = h->ad_xrow[li]; a.ldstore = l.kc; }
        a.ntaps = l.ntaps; a.kc = l.kc; a.ldtap = l.kc;
        const int o0 = -l.off[0], o1 = -l.off[1];
        a.tap0 = t - o0 >= 0 ? cone[k] + (size_t)idx_of(h->Hset[k], o0) * Bpad * l.kc : nullptr;
        a.tap1 = t - o1 >= 0 ? cone[k] + (size_t)idx_of(h->Hset[k], o1) * Bpad * l.kc : nullptr;
        a.Wt = l.Wt; a.ldw = l.ntaps * l.kc; a.bias = l.bias; a.H = h->ad_raw[li]; a.ldh = l.Nalloc; a.B = B;
        a.stop_after = stop_after; a.t = t;
        run_dec(h, a, l);
        prev = &l; prev_raw = h->ad_raw[li]; prev_x = h->ad_xrow[li];
    }
    h->gend(PC_DEC);
    // ---------------- row_chain C: AudioDec k=1 tail + mel frame t
    {
        RowChainArgs a{};
        a.pro = ROW_HC; a.src = prev_raw; a.ldsrc = prev->Nalloc; a.cin = d;
        a.g1 = prev->g1; a.b1 = prev->b1; a.g2 = prev->g2; a.b2 = prev->b2; a.xres = prev_x; a.ldres = prev->kc;
        a.lcc_pro = prev->lcc_gate;
        a.nlayers = (int)h->audiodec.size() - pre - nh;
        for (int i = 0; i < a.nlayers; ++i) a.L[i] = row_layer(h->audiodec[pre + nh + i]);
        a.L[a.nlayers - 1].act = (m.flags & OPH_FLAG_NO_SQUASH_T2M) ? ACT_NONE : ACT_SIGMOID;           // squash_output_t2m (networks.py:430-433)
        a.emit = 1; a.Yout = h->Yout; a.ldy = h->ldy; a.Ytm = h->Ytm; a.ldtm = h->ldy; a.max_T = m.max_T;
        a.B = B; a.Bpad = Bpad; a.stop_after = stop_after; a.t = t; a.d = d;
        run_row_chain(h, a, 0);
    }
}

// After a failed whole-decode launch (a hand-off timed out: its workgroups were not co-resident, or the cone never got
// CUs) the cross-stream words and counters are out of step: bring them back to a quiet state so that the handle stays usable.
void recover_loop_state(oph_handle* h) {
    for (hipStream_t st : {h->sdec, h->scone, h->sssrn, h->stream}) if (st) hipStreamSynchronize(st);
    (void)hipGetLastError();
    hipMemsetAsync(h->d_sig, 0, LOOP_SIG_WORDS * sizeof(uint32_t), h->stream);
    hipMemsetAsync(h->d_cone_count, 0, LOOP_MAX_LEVELS * sizeof(unsigned), h->stream);
    hipMemsetAsync(h->d_gbuf, 0, (size_t)LOOP_MAX_LAYERS * h->Bpad * RUN_GCOLS * 8, h->stream);
    hipStreamSynchronize(h->stream);
    for (uint32_t& v : h->cone_done_total) v = 0;
    h->sig_base = 0; h->run_epoch = 0;
}

// Steps [t_begin, t_end) of the CURRENT tile.  *steps_run = steps executed so far (stop step + 1 after an early stop).
int decode_range(oph_handle* h, int t_begin, int t_end, int stop_mode, int32_t* steps_run) {
    const oph_dims& m = h->dm;
    Tile& tl = h->tiles[h->tile];
    t_end = std::min(t_end, (int)m.max_T);
    int ctl[4] = {0, INT_MAX, 0, 0};
    int last = t_begin;
    // fork: the decode streams start after everything queued on the API stream (encode, resets)
    hipEventRecord(h->ev_in, h->stream);
    hipStreamWaitEvent(h->sdec, h->ev_in, 0);
    g_cur = h->sdec;
    if (h->use_run && h->run_epoch > 0xF0000000u) {      // tag wrap guard (once per ~10^5 batches): start over from zeroed granules
        for (hipStream_t st : {h->sdec, h->scone}) hipStreamSynchronize(st);
        hipMemsetAsync(h->d_gbuf, 0, (size_t)LOOP_MAX_LAYERS * h->Bpad * RUN_GCOLS * 8, h->sdec);
        hipStreamSynchronize(h->sdec);
        h->run_epoch = 0;
    }
    h->qw_from_loop = false;
    bool loop_mode = h->use_loop && !h->fixed_att && t_end > t_begin;
    if (loop_mode) {
        // every workgroup of the loop kernel must be resident at once on the critical stream's own CUs (the cone needs the
        // others): without that partition, or when the tile's workgroups do not fit it, take the two-launches-per-step path
        if (!h->d_loop_layers) { const int rc = build_loop_layers(h); if (rc) return rc; }
        if (h->loop_capacity < 0) h->loop_capacity = h->mask_words > 0 ? (h->chain_ok ? dec_chain_blocks_per_cu() : dec_loop_blocks_per_cu(h->loop_rows, h->loop_kmax)) * h->ndec_cus : 0;
        if (h->loop_slices * (h->Bpad / h->loop_rows) > h->loop_capacity) loop_mode = false;
    }
    if (h->use_sigval || loop_mode) {
        // a fresh value range for this loop: every value of an earlier loop is below sig_base + 1
        if (h->sig_base > 0x7fff0000u) {       // wrap guard (once per ~10 million batches): start over from a quiet state
            hipStreamSynchronize(h->sdec); hipStreamSynchronize(h->scone);
            hipMemsetAsync(h->d_sig, 0, LOOP_SIG_WORDS * sizeof(uint32_t), h->stream);
            hipStreamSynchronize(h->stream);
            h->sig_base = 0;
        }
        h->sig_base += (uint32_t)m.max_T + 2;
    }
    if (loop_mode && t_begin >= 1) {
        // the continued launch counts its attention arrivals from zero
        const int zero = 0;
        HIPCHK(h, hipMemcpyAsync(h->d_ctl + 3, &zero, 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    if (!loop_mode && t_begin >= 1 && t_begin < t_end) {      // resuming mid-utterance: cone(t_begin) has not been launched yet
        if (h->use_sigval) {
            hipStreamWriteValue32(h->sdec, h->d_sig, h->sig_base + (uint32_t)t_begin, 0);
            hipStreamWaitValue32(h->scone, h->d_sig, h->sig_base + (uint32_t)t_begin, hipStreamWaitValueGte, 0xffffffffu);
            launch_cone(h, t_begin);
            hipStreamWriteValue32(h->scone, h->d_sig + 16, h->sig_base + (uint32_t)t_begin, 0);
        } else {
            hipEventRecord(h->ev_attn, h->sdec);
            hipStreamWaitEvent(h->scone, h->ev_attn, 0);
            launch_cone(h, t_begin);
            hipEventRecord(h->ev_cone, h->scone);
        }
    }
    int rc_loop = OPH_OK;
    const auto tq0 = std::chrono::steady_clock::now();
    if (loop_mode) {
        h->n_loop_decodes++;
        if ((rc_loop = decode_loop(h, t_begin, t_end, stop_mode)) != OPH_OK) { recover_loop_state(h); return rc_loop; }
        last = t_end;
        t_begin = t_end;          // skip the per-step loop below
    }
    for (int t = t_begin; t < t_end; ++t) {
        decode_step(h, t, t_end, stop_mode);
        last = t + 1;
        // bounded look-ahead: poll the device-side stop flag every 8 steps (reference semantics keep
        // frames after the break step at zero because later steps early-out on the device)
        if (stop_mode == OPH_STOP_REFERENCE && ((t & 7) == 7)) {
            if (hipMemcpyAsync(ctl, h->d_ctl, sizeof ctl, hipMemcpyDeviceToHost, h->sdec) != hipSuccess ||
                hipStreamSynchronize(h->sdec) != hipSuccess) { rc_loop = OPH_ERR_DEVICE; break; }
            if (ctl[1] != INT_MAX) break;
        }
    }
    g_cur = h->sdec;
    if (rc_loop != OPH_OK) { h->fail("device error while polling the stop flag"); return rc_loop; }
    if (g_trace) {
        const double enq_ms = (std::chrono::duration<double>(std::chrono::steady_clock::now() - tq0).count() * 1e3);
        hipStreamSynchronize(h->sdec);
        hipStreamSynchronize(h->scone);
        const double all_ms = (std::chrono::duration<double>(std::chrono::steady_clock::now() - tq0).count() * 1e3);
        TRACE("decode loop: host enqueue %.2f ms, device drained %.2f ms after the first launch (last=%d); of the enqueue: "
              "event ops %.2f ms, cone launches %.2f ms", enq_ms, all_ms, last, g_host_us[0] * 1e-3, g_host_us[1] * 1e-3);
        g_host_us[0] = g_host_us[1] = 0;
        if (h->d_stamps) {      // phase durations of the stamped launch(es) of step max_T/2, averaged over the column slices (us)
            const int nruns = loop_mode ? 1 : 2, stride = loop_mode ? LOOP_MAX_LAYERS : RUN_MAX_LAYERS;
            std::vector<long long> st((size_t)2 * 32 * LOOP_MAX_LAYERS * 8);
            hipMemcpy(st.data(), h->d_stamps, st.size() * 8, hipMemcpyDeviceToHost);
            if (loop_mode && h->d_sigdbg) {
                std::vector<long long> sd((size_t)m.max_T * 8);
                hipMemcpy(sd.data(), h->d_sigdbg, sd.size() * 8, hipMemcpyDeviceToHost);
                for (int t : {50, 51, 100, 101, 150}) {
                    if (t >= m.max_T) continue;
                    const long long* q = &sd[(size_t)t * 8];
                    TRACE("step %d: the loop kernel spun for cone levels 0..5: %.2f %.2f %.2f %.2f %.2f %.2f us", t,
                          q[1] * 0.01, q[2] * 0.01, q[3] * 0.01, q[4] * 0.01, q[5] * 0.01, q[6] * 0.01);
                }
                if (h->d_lvldbg) {      // the cone of step t: release (attention of step t-1 done) -> each level complete, and the previous cone's end
                    std::vector<long long> lv((size_t)m.max_T * 8);
                    hipMemcpy(lv.data(), h->d_lvldbg, lv.size() * 8, hipMemcpyDeviceToHost);
                    for (int t : {50, 51, 100, 101, 150}) {
                        if (t >= m.max_T || t < 2) continue;
                        const long long rel = sd[(size_t)t * 8];
                        const long long* q = &lv[(size_t)t * 8];
                        const long long* qp = &lv[(size_t)(t - 1) * 8];
                        long long prev_end = 0;
                        for (int k = 0; k < 8; ++k) prev_end = std::max(prev_end, qp[k]);
                        TRACE("cone of step %d: levels 0..5 complete %.1f %.1f %.1f %.1f %.1f %.1f us after its release; the previous cone ended %.1f us %s it",
                              t, (q[0] - rel) * 0.01, (q[1] - rel) * 0.01, (q[2] - rel) * 0.01, (q[3] - rel) * 0.01, (q[4] - rel) * 0.01, (q[5] - rel) * 0.01,
                              std::fabs((double)(prev_end - rel)) * 0.01, prev_end > rel ? "AFTER" : "before");
                    }
                }
            }
            if (loop_mode && h->d_cldbg && h->cone_fused_ok) {
                std::vector<long long> cd(64);
                hipMemcpy(cd.data(), h->d_cldbg, cd.size() * 8, hipMemcpyDeviceToHost);
                for (int k = 0; k + 1 < h->n_hc_dec; ++k) {
                    const long long* q = &cd[(size_t)8 * k];
                    if (q[0]) TRACE("hc_fused level %d, workgroup 0: K loop %.2f  stats+publish %.2f  gather %.2f  normalise..store %.2f us", k + 1,
                                    (q[1] - q[0]) * 0.01, (q[2] - q[1]) * 0.01, (q[3] - q[2]) * 0.01, (q[4] - q[3]) * 0.01);
                }
                if (h->d_lvldbg && m.max_T / 2 + 1 < m.max_T) {      // the launches of step max_T/2 on one time line (us after level 0 completed)
                    std::vector<long long> lv((size_t)m.max_T * 8);
                    hipMemcpy(lv.data(), h->d_lvldbg, lv.size() * 8, hipMemcpyDeviceToHost);
                    const long long* q = &lv[(size_t)(m.max_T / 2) * 8];
                    const long long z = q[0];
                    for (int k = 0; k + 1 < h->n_hc_dec; ++k)
                        TRACE("step %d, level %d: workgroup 0 in at %.2f, out at %.2f; level complete at %.2f us (level 0 complete = 0)", m.max_T / 2, k + 1,
                              (cd[(size_t)8 * k] - z) * 0.01, (cd[(size_t)8 * k + 4] - z) * 0.01, (q[k + 1] - z) * 0.01);
                    TRACE("step %d: level 0 of the next cone complete at %.2f us", m.max_T / 2, (lv[(size_t)(m.max_T / 2 + 1) * 8] - z) * 0.01);
                }
            }
            if (loop_mode) {
                const long long* q = &st[(size_t)(LOOP_MAX_LAYERS - 1) * 8];
                if (q[1] > q[0]) TRACE("stamped step: %.2f us, shader clock %.0f MHz", (double)(q[1] - q[0]) * 0.01, (double)(q[3] - q[2]) / ((double)(q[1] - q[0]) * 0.01));
            }
            for (int run = 0; run < nruns; ++run)
                for (int l = 0; l < stride - (loop_mode ? 1 : 0); ++l) {
                    double d[5] = {0, 0, 0, 0, 0}, passes = 0, start = 0; int n = 0;
                    const long long t00 = st[((size_t)run * 32 + 0) * stride * 8 + 0];
                    for (int g = 0; g < 32; ++g) {
                        const long long* s_ = &st[(((size_t)run * 32 + g) * stride + l) * 8];
                        if (s_[0] == 0 || s_[5] == 0) continue;
                        for (int k = 0; k < 5; ++k) d[k] += (double)(s_[k + 1] - s_[k]) * 0.01;
                        passes += (double)s_[6]; start += (double)(s_[0] - t00) * 0.01; ++n;
                    }
                    double fma = 0;
                    for (int g = 0; g < 32; ++g) {
                        const long long* s_ = &st[(((size_t)run * 32 + g) * stride + l) * 8];
                        if (s_[0] == 0 || s_[5] == 0 || s_[7] == 0) continue;
                        fma += (double)(s_[7] - s_[3]) * 0.01;
                    }
                    if (n) TRACE("run %d layer %2d (%2d slices): start %+7.2f  sweep %.2f (%.1f passes)  prologue+stage %.2f  barrier %.2f  fma+prefetch %.2f (fma %.2f)  reduce+publish %.2f",
                                 run, l, n, start / n, d[0] / n, passes / n, d[1] / n, d[2] / n, d[3] / n, fma / n, d[4] / n);
                }
        }
    }
    // join: the API stream continues (SSRN, fetches) only after both decode streams drained
    hipEventRecord(h->ev_out, h->sdec);
    hipStreamWaitEvent(h->stream, h->ev_out, 0);
    hipEventRecord(h->ev_out, h->scone);
    hipStreamWaitEvent(h->stream, h->ev_out, 0);
    g_cur = h->stream;
    const bool need_ctl = steps_run || stop_mode == OPH_STOP_REFERENCE || h->use_run;
    if (need_ctl) {
        HIPCHK(h, hipMemcpyAsync(ctl, h->d_ctl, sizeof ctl, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (ctl[2] != 0) {
            h->last_wait_err = ctl[2];
            h->fail(ctl[2] == 3 ? "decoder loop: the side stream never saw the attention signal (time-out)" : ctl[2] == 2 ? "decoder loop: the side stream's cone never signalled (time-out)" :
                    ctl[2] == 4 ? "cone level: the column tiles of a row block never saw each other's statistics (time-out: workgroups of one launch were not co-resident)" :
                                  "decoder run: a workgroup hand-off timed out (workgroups of one run were not co-resident)");
            trace_cone_state(h);
            recover_loop_state(h);
            return OPH_ERR_DEVICE;
        }
        if (ctl[1] != INT_MAX) {
            // The reference leaves alignment columns after the break step at zero (synthesize.py:204-228).  With several row
            // groups a fast group of the loop kernel may have attended step stop+1 before the slowest one set the stop word
            // (about one step of skew, rare): clear those columns.  (A later resume rewrites the ones it decodes.)
            const int c0 = ctl[1] + 1;
            if (c0 < m.max_T)
                HIPCHK(h, hipMemset2DAsync(h->align + c0, (size_t)m.max_T * 4, 0, (size_t)(m.max_T - c0) * 4, (size_t)h->B * m.max_N, h->stream));
        }
    }
    const int steps = (need_ctl && ctl[1] != INT_MAX) ? ctl[1] + 1 : last;
    tl.steps = steps;
    if (steps_run) *steps_run = steps;
    HIPCHK(h, hipGetLastError());
    return OPH_OK;
}

// All tiles of the staged batch from step 0.  The reference's break couples the whole batch (synthesize.py:225-228: the
// loop ends after the step at which the LAST utterance has ended): every tile decodes to its own stop, then the tiles
// that stopped earlier resume to the batch's stop step -- the same fix-up the utterance shards of a multi-GPU run get.
int decode_batch(oph_handle* h, int t_end, int stop_mode, int32_t* steps_run) {
    const int ntiles = (h->nB + TILE - 1) / TILE;
    int batch_steps = 0;
    // A launch whose workgroups wait for each other (the whole-decode launch, the fused cone levels) needs them co-resident; nothing
    // guarantees that when another client holds CUs of the partition.  Every such wait is bounded (2 s) and ends in an error word; the
    // answer here is to redo the tile on a path that needs less: first the cone's levels as contraction + ln_rows launches (no
    // exchange between workgroups), then two launches per step instead of the whole-decode launch, then one launch per layer.
    // The fast paths are tried again after `rearm` decodes (16, then 4 x as many each time they fail again, at most 256);
    // oph_get_counters[10] = decodes left on the reduced paths.
    if (h->degraded_left > 0 && --h->degraded_left == 0) {
        TRACE("re-arming the whole-decode launch / fused cone after a degraded period");
        h->use_loop = h->use_loop_wanted; h->use_run = h->use_run_wanted; if (h->hcf_capacity == 0 && h->hcf_capacity_was_ok) h->hcf_capacity = -1;
    }
    // steps [t0, t1) of the current tile; after a failure the tile is redone from step 0 (a continued range without the stop rule:
    // the same frames as stop + resume)
    auto range_with_recovery = [&](int t0, int t1, int stop, int32_t* st) -> int {
        h->last_wait_err = 0;
        int rc = decode_range(h, t0, t1, stop, st);
        // only a co-residency time-out of THIS decode (ctl[2], read back by decode_loop / decode_step's wait) enters the ladder: an upload
        // failure, a device fault or the 10 s no-progress bound is not cured by another launch path and is returned as it is
        for (int attempt = 0; rc == OPH_ERR_DEVICE && h->last_wait_err != 0 && attempt < 3; ++attempt) {
            const char* what = nullptr;
            if (h->last_wait_err == 4 && h->hcf_capacity != 0) { h->hcf_capacity_was_ok = true; h->hcf_capacity = 0; what = "the cone's levels as separate contraction + LayerNorm launches"; }
            else if (h->use_loop) { h->use_loop = false; h->n_loop_fallbacks++; what = "two launches per step"; }
            else if (h->use_run) { h->use_run = false; what = "one launch per layer"; }
            else break;
            TRACE("decode failed (%s): redoing the tile with %s", h->err.c_str(), what);
            h->last_wait_err = 0;
            h->n_recoveries++;
            h->degraded_left = h->degraded_next; h->degraded_next = std::min(h->degraded_next * 4, 256);
            reset_decode(h);
            rc = decode_range(h, 0, t1, t0 == 0 ? stop : OPH_STOP_NEVER, st);
        }
        return rc;
    };
    for (int j = 0; j < ntiles; ++j) {
        select_tile(h, j);
        reset_decode(h);
        int32_t st = 0;
        int rc = range_with_recovery(0, t_end, stop_mode, &st);
        if (rc) return rc;
        batch_steps = std::max(batch_steps, (int)st);
        // a tile that ran to the end has all its frames: what SSRN has not covered yet goes to the SSRN partition now, under the
        // next tile's decode (a tile that stopped early may still be resumed: its tail waits for the batch's stop step)
        if (h->spec_ssrn && !h->opt.no_stream_ssrn && h->opt.ssrn_chunk > 0 && j + 1 < ntiles && st == t_end && t_end == h->dm.max_T &&
            (rc = ssrn_stream_chunks(h, h->dm.max_T, true, true)))
            return rc;
    }
    if (stop_mode == OPH_STOP_REFERENCE && ntiles > 1)
        for (int j = 0; j < ntiles; ++j) {
            if (h->tiles[j].steps >= batch_steps) continue;
            select_tile(h, j);
            const int ctl1 = INT_MAX;
            HIPCHK(h, hipMemcpyAsync(h->d_ctl + 1, &ctl1, 4, hipMemcpyHostToDevice, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            {   // frames from the tile's stop step on are about to change: SSRN rows that saw them are stale
                int back = 0, ahead = 0;
                ssrn_margins(h, &back, &ahead);
                h->tiles[j].ssrn_done = std::min(h->tiles[j].ssrn_done, std::max(0, h->tiles[j].steps - ahead));
                h->tiles[j].z_copied = std::min(h->tiles[j].z_copied, h->tiles[j].ssrn_done);
            }
            int32_t st_ = 0;
            const int rc = range_with_recovery(h->tiles[j].steps, batch_steps, OPH_STOP_NEVER, &st_);
            if (rc) return rc;
            h->n_tile_resumes++;
            h->tiles[j].steps = batch_steps;
        }
    select_tile(h, 0);
    if (steps_run) *steps_run = batch_steps;
    return OPH_OK;
}

