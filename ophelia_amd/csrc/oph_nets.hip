// libophelia_hip.so -- the batched nets: TextEnc (networks.py:121-212), SSRN (437-537) and the one-pass graph evaluation run as one
// conv-GEMM + one LayerNorm launch per layer over dense (utterance, frame) rows; SSRN streamed in chunks under the decode that
// produces its input frames (synth_mel2mag, synthesize.py:250-260).
#include "oph_host.h"

// ------------------------------------------------------------------ launch wrappers with accounting
void run_gemm(oph_handle* h, const GemmArgs& a, int cin_true, int prec) {       // prec: 0 fp32 MFMA, 1 split-bf16 x3, 2 split-fp16 x3 (a.Wh / a.Wl / a.f16 set to match)
    const int cls = prec ? PC_GEMM_BF16 : (conv_gemm_tile_m(a.M, a.N) == 128 ? PC_GEMM : PC_GEMM64);
    h->pbegin(cls);
    if (prec) launch_conv_gemm_bf16x3(a, g_cur);
    else launch_conv_gemm(a, g_cur);
    const double K = (double)a.ntaps * cin_true;
    h->pend(cls, ((double)a.M * cin_true + (double)a.M * a.N + (double)a.N * K) * 4.0, 2.0 * a.M * a.N * K);
}
void run_epi(oph_handle* h, const EpiArgs& a) {
    h->pbegin(PC_LN);
    launch_epilogue(a, g_cur);
    const double cols = a.mode == PRE_HC ? 4.0 * a.C : 2.0 * a.C;    // read raw (+res), write out
    h->pend(PC_LN, (double)a.M * cols * 4.0, (double)a.M * a.C * 10.0);
}
void run_dec(oph_handle* h, const DecArgs& a, const Layer& l) {
    h->pbegin(PC_DEC);
    launch_dec_layer(a, round_up(l.N, 16), g_cur);
    const double K = (double)l.ntaps * l.cin;
    h->pend(PC_DEC, ((double)l.N * K + (double)a.B * (K + l.N)) * 4.0, 2.0 * a.B * l.N * K);
}

// ------------------------------------------------------------------ batched networks (TextEnc, SSRN, ops)
float* run_batched(oph_handle* h, const std::vector<Layer>& layers, float* in, int ld_in, int B, int T, int wsi, int prec,
                   float* final_out, int final_ld, int final_pad, int* out_ld, long long* out_rows, const BatchedIO& io) {
    const int* spk_ids = io.spk ? io.spk : h->bSpk[h->txt];
    float* x = in;
    int ldx = ld_in;
    int Tcur = T;
    float* const wsA = wsi ? h->actA2 : h->actA; float* const wsB = wsi ? h->actB2 : h->actB; float* const wsraw = wsi ? h->raw2 : h->raw;
    float* bufs[2] = {wsA, wsB};
    int flip = (in == wsA) ? 1 : 0;
    // split-fp16 contractions read their activations as hi / lo planes where the previous layer's LayerNorm launch wrote them
    // (plane_gemm, oph_planegemm.hip); a layer whose input has no planes (the first one, a speaker-embedding concat, tap offsets
    // beyond the halo) runs conv_gemm_bf16x3 on the fp32 rows
    const bool planes_on = prec == 2 && !h->opt.no_plane_gemm && h->actP[wsi ? 1 : 0][0][0];
    auto wants_planes = [&](const Layer& l) {
        if (!planes_on || l.ccat > 0 || !l.Wkh || !l.Wkl) return false;
        if (l.kind == K_CONVT) return l.Wkh2 && l.Wkl2 && plane_gemm_ok(2, nullptr, l.kc, true);
        return plane_gemm_ok(l.ntaps, l.off, l.kc, false);
    };
    const _Float16 *xh = nullptr, *xl = nullptr;      // planes of x (the current layer's input), if any
    for (size_t li = 0; li < layers.size(); ++li) {
        const Layer& l = layers[li];
        const bool last = li + 1 == layers.size();
        const int M = B * Tcur;
        const bool use_planes = xh && wants_planes(l);
        const bool write_planes = !last && wants_planes(layers[li + 1]);
        float* y = (last && final_out) ? final_out : bufs[flip];
        const int cout_pad = round_up(l.cout, 32);
        const int ldy = (last && final_out) ? final_ld : cout_pad;
        const int ypad = (last && final_out) ? final_pad : cout_pad;
        GemmArgs g{};
        g.X = x; g.ldx = ldx; g.bias = l.bias; g.H = wsraw; g.kc = l.kc; g.mode = 0; g.T = Tcur;
        g.stop_after = nullptr;
        EpiArgs e{};
        e.g1 = l.g1; e.b1 = l.b1; e.g2 = l.g2; e.b2 = l.b2; e.act = l.act; e.Y = y; e.ldy = ldy; e.ypad = ypad;
        e.H = wsraw; e.stop_after = nullptr; e.nonorm = !l.ln;
        e.lcc = l.lcc_gate; e.lcc_ids = spk_ids; e.lcc_T = Tcur;
        if (last && final_out && io.out_T > 0) { e.out_T = io.out_T; e.keep_lo = io.keep_lo; e.keep_hi = io.keep_hi; e.out_bs = io.out_bs; e.out_t0 = io.out_t0; }
        if (!last && layers[li + 1].ccat > 0) {       // the next layer's input = [this output | speaker embedding]
            const Layer& nx = layers[li + 1];
            e.spk_table = nx.cat_table ? nx.cat_table : h->emb_spk;      // AudioDec 'audio_decoder_input': embed_2
            e.spk_ids = spk_ids; e.spk_dim = nx.ccat; e.spk_T = Tcur;
            e.ldy = e.ypad = nx.kc;
        }
        if (l.kind == K_CONVT) {
            // even rows: taps (x[t], x[t-1]); odd rows: tap x[t]; raw rows interleaved 2t / 2t+1
            g.N = l.N; g.ldh = 2 * l.Nalloc; g.M = M;
            const bool f16 = prec >= 2;
            g.nprod = prec == 3 ? 2 : (prec == 4 ? 1 : 3);
            g.Wt = l.Wt; g.Wh = f16 ? l.Wh16 : l.Wh; g.Wl = f16 ? l.Wl16 : l.Wl; g.f16 = f16; g.ldw = 2 * l.kc; g.ntaps = 2; g.off[0] = 0; g.off[1] = -1;
            GemmArgs g2 = g;
            g2.Wt = l.Wt2; g2.Wh = f16 ? l.Wh2_16 : l.Wh2; g2.Wl = f16 ? l.Wl2_16 : l.Wl2; g2.ldw = l.kc; g2.ntaps = 1; g2.off[0] = 0; g2.H = wsraw + l.Nalloc;
            // LayerNorm inside the launch (round 6): the launch writes the normalised rows and the next layer's planes itself
            const bool fuse_ln = use_planes && !h->opt.no_fused_convt_ln && !h->pg_ln_off && l.ln && !last && l.N % 64 == 0 && l.N <= 1024 && !l.lcc_gate && layers[li + 1].ccat == 0 &&
                                 h->d_pg_stats[wsi ? 1 : 0] && h->d_pg_err && plane_gemm_ln_stats_bytes(M, l.N) <= h->pg_stats_bytes && convt_ln_fits(h, wsi);
            if (use_planes) {    // both phases as one problem on the planes
                PlaneGemmArgs pg{};
                pg.Ah = xh; pg.Al = xl; pg.Wh = (const _Float16*)l.Wkh; pg.Wl = (const _Float16*)l.Wkl; pg.Wh2 = (const _Float16*)l.Wkh2; pg.Wl2 = (const _Float16*)l.Wkl2;
                pg.bias = l.bias; pg.H = wsraw; pg.M = M; pg.N = l.N; pg.kc = l.kc; pg.T = Tcur; pg.nalloc = l.Nalloc; pg.ldh = 2 * l.Nalloc;
                pg.ntaps = 2; pg.off[0] = 0; pg.off[1] = -1; pg.convt = 1; pg.waves = h->opt.pg_waves;
                if (fuse_ln) {
                    if (h->pg_epoch > 0xF0000000u) {          // tag wrap guard: start over from zeroed regions
                        hipDeviceSynchronize();
                        for (float* p_ : h->d_pg_stats) if (p_) hipMemset(p_, 0, h->pg_stats_bytes);
                        h->pg_epoch = 0;
                    }
                    pg.ln_gamma = l.g1; pg.ln_beta = l.b1; pg.Y = y; pg.ldy = ldy;
                    if (write_planes) { pg.Yh = (_Float16*)h->actP[wsi ? 1 : 0][flip][0]; pg.Yl = (_Float16*)h->actP[wsi ? 1 : 0][flip][1]; }
                    pg.ln_stats = h->d_pg_stats[wsi ? 1 : 0]; pg.ln_epoch = ++h->pg_epoch; pg.ln_err = h->d_pg_err;
                    if (h->opt.fake_ln_timeout) pg.dbg = 64;       // (measurement builds: fault injection)
                }
                h->pbegin(PC_PLANEGEMM);
                launch_plane_gemm(pg, g_cur);
                h->pend(PC_PLANEGEMM, ((double)g.M * l.cin + 2.0 * g.M * g.N + 3.0 * g.N * l.cin) * 4.0, 2.0 * g.M * g.N * 3.0 * l.cin);
            } else {   // both phases in one launch
                const int p2 = (prec && g.Wh && g2.Wh) ? std::min(prec, 2) : 0;
                const int cls = p2 ? PC_GEMM_BF16 : (conv_gemm_tile_m(g.M, g.N) == 128 ? PC_GEMM : PC_GEMM64);
                h->pbegin(cls);
                launch_conv_gemm_pair(g, g2, p2, g_cur);
                h->pend(cls, ((double)g.M * l.cin + 2.0 * g.M * g.N + 3.0 * g.N * l.cin) * 4.0, 2.0 * g.M * g.N * 3.0 * l.cin);
            }
            Tcur *= 2;
            e.ldh = l.Nalloc; e.M = B * Tcur; e.C = l.cout; e.mode = PRE_CONV; e.act = ACT_NONE;
            if (write_planes) { e.planes = 1; e.Yh = h->actP[wsi ? 1 : 0][flip][0]; e.Yl = h->actP[wsi ? 1 : 0][flip][1]; }
            if (!fuse_ln) run_epi(h, e);
        } else {
            const bool f16 = prec >= 2;
            g.nprod = prec == 3 ? 2 : (prec == 4 ? 1 : 3);
            g.N = l.N; g.ldh = l.Nalloc; g.M = M; g.Wt = l.Wt; g.Wh = f16 ? l.Wh16 : l.Wh; g.Wl = f16 ? l.Wl16 : l.Wl; g.f16 = f16; g.ldw = l.ntaps * l.kc; g.ntaps = l.ntaps;
            for (int t = 0; t < 3; ++t) g.off[t] = l.off[t];
            if (use_planes) {
                PlaneGemmArgs pg{};
                pg.Ah = xh; pg.Al = xl; pg.Wh = (const _Float16*)l.Wkh; pg.Wl = (const _Float16*)l.Wkl;
                pg.bias = l.bias; pg.H = wsraw; pg.M = M; pg.N = l.N; pg.kc = l.kc; pg.T = Tcur; pg.nalloc = l.Nalloc; pg.ldh = l.Nalloc;
                pg.ntaps = l.ntaps;
                for (int t = 0; t < 3; ++t) pg.off[t] = l.off[t];
                pg.waves = h->opt.pg_waves;
                h->pbegin(PC_PLANEGEMM);
                launch_plane_gemm(pg, g_cur);
                const double K = (double)l.ntaps * l.cin;
                h->pend(PC_PLANEGEMM, ((double)M * l.cin + (double)M * l.N + (double)l.N * K) * 4.0, 2.0 * M * l.N * K);
            } else
                run_gemm(h, g, l.cin, g.Wh ? std::min(prec, 2) : 0);
            e.ldh = l.Nalloc; e.M = M; e.C = l.cout;
            if (l.kind == K_HC) { e.mode = PRE_HC; e.Xres = x; e.ldres = ldx; }
            else e.mode = PRE_CONV;
            if (write_planes) { e.planes = 1; e.Yh = h->actP[wsi ? 1 : 0][flip][0]; e.Yl = h->actP[wsi ? 1 : 0][flip][1]; }
            run_epi(h, e);
            if (last && io.final_logits && l.kind == K_CONV) {      // the fetch surface's g.Z_logits / g.Y_logits: the same rows before the squash
                EpiArgs el = e;
                el.act = ACT_NONE; el.Y = io.final_logits;
                run_epi(h, el);
            }
        }
        x = y; ldx = e.ldy;
        xh = write_planes ? (const _Float16*)e.Yh : nullptr; xl = write_planes ? (const _Float16*)e.Yl : nullptr;
        flip ^= 1;
    }
    if (out_ld) *out_ld = ldx;
    if (out_rows) *out_rows = (long long)B * Tcur;
    return x;
}

// Can the column tiles of a row tile of the fused conv1d_transpose + LayerNorm launch be resident together on every XCD of the stream
// the workspace set `wsi` runs on?  They wait for each other's statistics (up to 8 workgroups of one row tile per XCD, one per CU).
bool convt_ln_fits(const oph_handle* h, int wsi) {
    int ncu = 0;
    if (wsi && h->mask_words > 0) for (int i = 0; i < h->mask_words; ++i) ncu += __builtin_popcount(h->m_ssrn[i]);
    else { hipDeviceProp_t prop; ncu = hipGetDeviceProperties(&prop, h->device) == hipSuccess ? prop.multiProcessorCount : 0; }
    return ncu / 8 >= 8;
}
// After a synchronisation of the streams SSRN ran on: did a fused conv1d_transpose + LayerNorm launch of this batch time out (the column
// tiles of a row tile never saw each other's statistics: workgroups of one launch were not co-resident -- another tenant on the
// SSRN CUs)?  Its rows are garbage then.  Like the decode's ladder: the batch's SSRN is redone in the two-launch form, which needs no
// exchange between workgroups (z_host: the host destination the redone rows are copied to, or null), the handle stays on that form, and
// the event is counted (oph_get_counters[9]).
int check_convt_ln(oph_handle* h, float* z_host) {
    if (!h->host_prog || h->host_prog[8] == 0) return OPH_OK;
    for (hipStream_t st : {h->stream, h->sssrn, h->scopy}) if (st) hipStreamSynchronize(st);
    h->host_prog[8] = 0;
    h->pg_ln_off = true;
    h->n_recoveries++;
    TRACE("conv1d_transpose + LayerNorm: statistics exchange timed out -- SSRN of this batch is redone with two launches per transposed layer");
    if (h->pipelined || !h->y_resident) {
        // (pipelined batches: the frames of the batch whose SSRN tail timed out may already have been replaced by the next decode's --
        //  nothing to redo it from.  Loud, once: the handle is on the two-launch form from here on.)
        h->fail("conv1d_transpose + LayerNorm: the statistics exchange timed out (workgroups of one launch were not co-resident) and the batch's "
                "frames are no longer resident to redo SSRN from: that batch's spectrogram is invalid; later batches use the two-launch form");
        return OPH_ERR_DEVICE;
    }
    for (Tile& tl : h->tiles) { tl.ssrn_done = 0; tl.z_copied = 0; }
    float* const saved = h->z_host;
    h->z_host = z_host;
    int rc = finish_ssrn(h);           // (not pipelined: on the API stream, now)
    h->z_host = saved;
    for (hipStream_t st : {h->stream, h->sssrn, h->scopy}) if (st && hipStreamSynchronize(st) != hipSuccess && !rc) { h->fail("SSRN redo failed"); rc = OPH_ERR_DEVICE; }
    return rc;
}

int ensure_batched_capacity(oph_handle* h, int B) {
    if (B <= h->capB) return 0;
    const oph_dims& m = h->dm;
    const long long rows_ssrn = (long long)B * m.max_T * m.r, rows_text = (long long)B * m.max_N;
    const long long rows = std::max(rows_ssrn, rows_text);
    const int ld_act = round_up(std::max({2 * m.c, m.full_dim, 2 * m.d}), 32);
    const int ld_raw = round_up(std::max({4 * m.c, m.full_dim, 4 * m.d}), 128);
    h->act_elems = (size_t)rows * ld_act;
    h->raw_elems = (size_t)rows * ld_raw;
    h->actA = h->dalloc<float>(h->act_elems);
    h->actB = h->dalloc<float>(h->act_elems);
    h->raw = h->dalloc<float>(h->raw_elems);
    h->actA2 = h->dalloc<float>(h->act_elems);
    h->actB2 = h->dalloc<float>(h->act_elems);
    h->raw2 = h->dalloc<float>(h->raw_elems);
    for (int w_ = 0; w_ < 2; ++w_) for (int b_ = 0; b_ < 2; ++b_) for (int p_ = 0; p_ < 2; ++p_) {
        h->actP[w_][b_][p_] = h->dalloc<unsigned short>(h->act_elems);
        if (!h->actP[w_][b_][p_]) { h->fail("out of device memory for batch %d", B); return OPH_ERR_DEVICE; }
    }
    if (!h->actA || !h->actB || !h->raw || !h->actA2 || !h->actB2 || !h->raw2) { h->fail("out of device memory for batch %d", B); return OPH_ERR_DEVICE; }
    {   // exchange regions of the fused conv1d_transpose + LayerNorm launches (the largest: the last transposed layer's input rows)
        h->pg_stats_bytes = plane_gemm_ln_stats_bytes((int)(rows_ssrn / 2), m.c);
        for (int w_ = 0; w_ < 2; ++w_) h->d_pg_stats[w_] = (float*)h->dalloc<unsigned char>(h->pg_stats_bytes);      // (zero-filled)
        void* dp = nullptr;
        if (h->host_prog && hipHostGetDevicePointer(&dp, (void*)h->host_prog, 0) == hipSuccess) { h->d_pg_err = (int*)dp + 8; h->host_prog[8] = 0; }
        (void)hipGetLastError();
    }
    h->capB = B;
    return 0;
}

// TextEnc (networks.py:121-212) of B staged utterances (ids dL, speakers dSpk) into `KVdst` on `stream` with workspace set `wsi`
int run_encode_into(oph_handle* h, const int* dL, const int* dSpk, int B, float* KVdst, hipStream_t stream, int wsi) {
    const oph_dims& m = h->dm;
    hipStream_t saved = g_cur;
    g_cur = stream;
    h->n_textenc++;
    float* ws = wsi ? h->actA2 : h->actA;
    // embed_1 (modules.py:15-44) -> rows [B*max_N][e]
    const Layer& first = h->textenc[0];
    const int ld0 = first.kc;                     // round_up(e [+ speaker embedding], 32)
    h->pbegin(PC_MISC);
    launch_embed(dL, (long long)B * m.max_N, h->emb_text, m.e, ws, ld0, stream);
    h->pend(PC_MISC, (double)B * m.max_N * m.e * 4.0, 0);
    if (first.cat_table)                          // 'text_encoder_input': [embed(L) | embed(speaker)]  networks.py:138-144
        launch_spk_append_rows(ws, ld0, (long long)B * m.max_N, m.max_N, m.e, first.cat_table, dSpk, first.ccat, stream);
    // last highway layer writes K|V rows straight into the resident KV buffer [B][N][2d]
    BatchedIO io{};
    io.spk = dSpk;
    run_batched(h, h->textenc, ws, ld0, B, m.max_N, wsi, h->textenc_prec, KVdst, 2 * m.d, 2 * m.d, nullptr, nullptr, io);
    g_cur = saved;
    HIPCHK(h, hipGetLastError());
    return OPH_OK;
}
// the whole staged batch, on the API stream
int run_encode(oph_handle* h) { return run_encode_into(h, h->bL[h->txt], h->bSpk[h->txt], h->nB, h->bKV[h->kv_cur], h->stream, 0); }

int run_ssrn_on(oph_handle* h, const float* Yrows, int ldy, int B, int T, float* Zout, int wsi, float* Zlogits, const int* dSpk) {
    const oph_dims& m = h->dm;
    BatchedIO io{};
    io.final_logits = Zlogits;
    io.spk = dSpk;                    // 'ssrn_input' (networks.py:457-465); null: the staged batch's codes
    run_batched(h, h->ssrn, const_cast<float*>(Yrows), ldy, B, T, wsi, h->ssrn_prec, Zout, m.full_dim, m.full_dim, nullptr, nullptr, io);
    HIPCHK(h, hipGetLastError());
    return OPH_OK;
}

// ---------------------------------------------------------------- streamed SSRN
// SSRN (networks.py:437-537) is not causal, but its receptive field is short: output rows of mel frame f depend on mel
// frames [f - SSRN_BACK, f + SSRN_AHEAD) only (HC r1, r3 at T; D_4; HC r1, r3 at 2T; D_7; HC r1, r3 at 4T; HC_11, HC_12 r1:
// back 1+3 + ceil((1 + 1+3 + ceil((1 + 1+3+1+1) / 2)) / 2) = 9, ahead 1+3 + (1+3 + (1+3+1+1+1)/2)/2 = 7.x -> 8).
// So the rows of frames [a, b) can be computed from frames [a - 9, b + 8) as soon as those exist, while the decoder
// is still producing later frames: the chunk is run as a dense batch over the extended range (values near the range's
// ends are wrong and are not stored; at the true sequence ends the range is clamped and SAME padding applies as in the
// one-shot run).  Every output element is the same dot product in the same order as in the one-shot run: bitwise equal.
// The margins are derived from the layer list (ssrn_margins) rather than hard-coded.
void ssrn_margins(const oph_handle* h, int* back, int* ahead) {
    // walk the layers from the output back to the input: an output row u needs input rows [u - lo, u + hi]
    int lo = 0, hi = 0;
    for (size_t i = h->ssrn.size(); i-- > 0;) {
        const Layer& l = h->ssrn[i];
        if (l.kind == K_CONVT) { lo = (lo + 1) / 2 + 1; hi = (hi + 1) / 2; }     // out[2t] reads x[t], x[t-1]; out[2t+1] reads x[t]
        else if (l.size == 3) { lo += l.rate; hi += l.rate; }
    }
    *back = lo + 1; *ahead = hi + 1;     // lj_tutorial: 9 + 1 and 8 + 1 (one frame of slack each side)
}

// SSRN rows of mel frames [a, b) of the CURRENT tile -> the batch's host destination, on the copy stream, after everything queued on `after`
int copy_mag_rows(oph_handle* h, int a, int b, hipStream_t after) {
    if (!h->z_host || b <= a) return OPH_OK;
    const oph_dims& m = h->dm;
    HIPCHK(h, hipEventRecord(h->ev_chunk, after));
    HIPCHK(h, hipStreamWaitEvent(h->scopy, h->ev_chunk, 0));
    const size_t rowb = (size_t)m.full_dim * 4, pitch = (size_t)m.max_T * m.r * rowb, r0 = (size_t)h->tile * TILE;
    HIPCHK(h, hipMemcpy2DAsync((char*)h->z_host + r0 * pitch + (size_t)a * m.r * rowb, pitch, (const char*)h->Z + (size_t)a * m.r * rowb, pitch,
                               (size_t)(b - a) * m.r * rowb, (size_t)h->B, hipMemcpyDeviceToHost, h->scopy));
    return OPH_OK;
}

// Z rows of mel frames [a, b) of the CURRENT tile, from its resident Yout, on stream `st` with workspace `wsi`.
int run_ssrn_chunk(oph_handle* h, int a, int b, hipStream_t st, int wsi) {
    const oph_dims& m = h->dm;
    int back = 0, ahead = 0;
    ssrn_margins(h, &back, &ahead);
    const int lo = std::max(0, a - back), hi = std::min((int)m.max_T, b + ahead), Tc = hi - lo;
    hipStream_t saved = g_cur;
    g_cur = st;
    float* ws = wsi ? h->actB2 : h->actB;
    // the chunk's input frames as a dense [B][Tc] batch
    launch_copy_rows_strided(h->Yout + (size_t)lo * h->ldy, (long long)m.max_T * h->ldy, h->ldy, ws, h->B, Tc, h->ldy, st);
    BatchedIO io{};
    io.out_T = Tc * m.r; io.keep_lo = (a - lo) * m.r; io.keep_hi = (b - lo) * m.r;
    io.out_bs = (long long)m.max_T * m.r; io.out_t0 = lo * m.r;
    io.spk = h->d_spk;                // this tile's slice of the staged speaker codes ('ssrn_input')
    run_batched(h, h->ssrn, ws, h->ldy, h->B, Tc, wsi, h->ssrn_prec, h->Z, m.full_dim, m.full_dim, nullptr, nullptr, io);
    g_cur = saved;
    if (h->z_host) {
        // the chunk's rows leave for the host on the copy stream while the decode goes on.  The copied frontier only moves over a
        // contiguous range: rows computed earlier without a destination (a resumed decode) are picked up by finish_ssrn
        Tile& tl = h->tiles[h->tile];
        const int from = std::min(a, tl.z_copied);
        const int rc = copy_mag_rows(h, from, b, st);
        if (rc) return rc;
        tl.z_copied = b;
    }
    HIPCHK(h, hipGetLastError());
    return OPH_OK;
}

// Launch the chunks of the current tile whose input frames exist: `frames_ready` = mel frames stored so far.  Chunks of
// opt.ssrn_chunk frames on the SSRN partition while the decode runs; final: everything that is left (the decode is over).
int ssrn_stream_chunks(oph_handle* h, int frames_ready, bool final, bool side_tail) {
    const oph_dims& m = h->dm;
    Tile& tl = h->tiles[h->tile];
    int back = 0, ahead = 0;
    ssrn_margins(h, &back, &ahead);
    const int ch = h->opt.ssrn_chunk;
    while (tl.ssrn_done < m.max_T) {
        int a = tl.ssrn_done, b;
        if (final) b = m.max_T;
        else {
            b = a + ch;
            if (b + ahead > frames_ready || b >= m.max_T) break;
            // one chunk in flight on the partition; its measured duration tells whether another one can still finish before
            // the decode does -- if not, those frames are cheaper in the final piece on the whole chip
            if (h->chunk_inflight) {
                if (hipEventQuery(h->ev_ce) != hipSuccess) { (void)hipGetLastError(); break; }
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, h->ev_cs, h->ev_ce) == hipSuccess) h->chunk_ms = ms;
                h->chunk_inflight = false;
            }
            if (h->chunk_ms > 0.f && frames_ready - h->dec_tbegin > 8 && !h->pipelined) {
                // (dec_t0 is the launch of steps [dec_tbegin, dec_tend): a resumed decode counts its own frames only)
                const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - h->dec_t0).count() * 1e3;
                const double remaining = elapsed / std::max(1, frames_ready - h->dec_tbegin) * std::max(0, h->dec_tend - frames_ready);
                if (h->chunk_ms > remaining && b - a >= ch) break;       // (the short chunk in front of a bounded final piece goes anyway)
            }
        }
        // while the decode runs: the SSRN partition; afterwards, not pipelined: the whole chip through the API stream (which
        // the decode streams have joined)
        const bool side = !final || h->pipelined || side_tail;       // side_tail: a finished tile's last piece, under the next tile's decode
        if (!final) hipEventRecord(h->ev_cs, h->sssrn);
        const int rc = run_ssrn_chunk(h, a, b, side ? h->sssrn : h->stream, side ? 1 : 0);
        if (rc) return rc;
        if (!final) { hipEventRecord(h->ev_ce, h->sssrn); h->chunk_inflight = true; h->n_chunks_streamed++; }
        tl.ssrn_done = b;
    }
    return OPH_OK;
}
// SSRN of every tile brought up to date (what streaming has not covered yet); the API stream has joined the decode.
int finish_ssrn(oph_handle* h) {
    const int ntiles = (h->nB + TILE - 1) / TILE;
    if (h->pipelined) {      // the tails run on the SSRN partition behind this batch's decode, under the next batch's
        HIPCHK(h, hipEventRecord(h->ev_dec_done, h->stream));
        HIPCHK(h, hipStreamWaitEvent(h->sssrn, h->ev_dec_done, 0));
    }
    for (int j = 0; j < ntiles; ++j) {
        select_tile(h, j);
        Tile& tl = h->tiles[j];
        if (h->z_host && tl.z_copied < tl.ssrn_done) {
            // rows that were computed while no host destination was set (chunks streamed under a resumed decode, oph_decode_steps):
            // they are final, and every stream that may have produced them is ordered before the copy
            int rc = copy_mag_rows(h, tl.z_copied, tl.ssrn_done, h->sssrn);
            if (rc) return rc;
            tl.z_copied = tl.ssrn_done;
        }
        const int rc = ssrn_stream_chunks(h, h->dm.max_T, true);
        if (rc) return rc;
    }
    select_tile(h, 0);
    if (h->pipelined) {
        HIPCHK(h, hipEventRecord(h->ev_ssrn_done[h->buf], h->sssrn));
        h->ssrn_inflight[h->buf] = true;
    } else {
        // chunks streamed during the decode ran on the SSRN partition: the API stream waits for them
        HIPCHK(h, hipEventRecord(h->ev_ssrn_done[h->buf], h->sssrn));
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_ssrn_done[h->buf], 0));
    }
    return OPH_OK;
}

