// libophelia_hip.so -- the networks of the synthesis path as layer lists (networks.py: TextEnc 121-212, AudioEnc 214-284, AudioDec
// 360-435, SSRN 437-537; variable names and creation order of architectures.py / modules.py), weight packing into the kernels' layouts,
// and the weight entry points of the C ABI (replaces tf.train.Saver.restore, synthesize.py:302-330).
#include "oph_host.h"

// ------------------------------------------------------------------ network description
static void add_conv(std::vector<Layer>& v, const std::string& scope, int cin, int cout, bool causal, int act, int ccat = 0) {
    Layer l;
    l.scope = scope; l.kind = K_CONV; l.cin = cin; l.cout = cout; l.size = 1; l.rate = 1;
    l.causal = causal; l.act = act; l.ccat = ccat;
    v.push_back(l);
}
static void add_hc(std::vector<Layer>& v, const std::string& scope, int c, int size, int rate, bool causal) {
    Layer l;
    l.scope = scope; l.kind = K_HC; l.cin = c; l.cout = c; l.size = size; l.rate = rate; l.causal = causal;
    v.push_back(l);
}
static std::string sc(const char* net, const char* pfx, int i) {
    char b[128];
    snprintf(b, sizeof b, "%s/%s_%d", net, pfx, i);
    return b;
}

void build_networks(oph_handle* h) {
    const oph_dims& m = h->dm;
    const int d = m.d, c = m.c;
    {   // TextEnc  networks.py:121-212
        const char* n = "Text2Mel/TextEnc";
        int i = 2;                                    // embed_1 handled separately
        const int se = m.speaker_embedding_size;
        if (m.flags & OPH_FLAG_SPK_TEXT_ENCODER_INPUT) {          // networks.py:138-144: embed_2, concat, C_3
            const std::string es = sc(n, "embed", i++);
            add_conv(h->textenc, sc(n, "C", i++), m.e + se, 2 * d, false, ACT_RELU, se);
            h->textenc.back().cat_scope = es;
        } else {
            add_conv(h->textenc, sc(n, "C", i++), m.e, 2 * d, false, ACT_RELU);
        }
        add_conv(h->textenc, sc(n, "C", i++), 2 * d, 2 * d, false, ACT_NONE);
        for (int o = 0; o < 2; ++o)
            for (int j = 0, r = 1; j < 4; ++j, r *= 3) add_hc(h->textenc, sc(n, "HC", i++), 2 * d, 3, r, false);
        for (int o = 0; o < 2; ++o) add_hc(h->textenc, sc(n, "HC", i++), 2 * d, 3, 1, false);
        if (m.flags & OPH_FLAG_SPK_TEXT_ENCODER_TOWARDS_END) {    // networks.py:184-199: embed, concat, 1x1 conv back to 2d
            const std::string es = sc(n, "embed", i++);
            add_conv(h->textenc, sc(n, "C", i++), 2 * d + se, 2 * d, false, ACT_RELU, se);
            h->textenc.back().cat_scope = es;
        }
        for (int o = 0; o < 2; ++o) add_hc(h->textenc, sc(n, "HC", i++), 2 * d, 1, 1, false);
    }
    {   // AudioEnc  networks.py:214-284
        const char* n = "Text2Mel/AudioEnc";
        int i = 1;
        add_conv(h->audioenc, sc(n, "C", i++), m.n_mels, d, true, ACT_RELU);
        if (m.flags & OPH_FLAG_SPK_AUDIO_ENCODER_INPUT) {         // networks.py:237-245: embed, concat, 1x1 conv (no act)
            const std::string es = sc(n, "embed", i++);
            add_conv(h->audioenc, sc(n, "C", i++), d + m.speaker_embedding_size, d, false, ACT_NONE, m.speaker_embedding_size);
            h->audioenc.back().cat_scope = es;
        }
        add_conv(h->audioenc, sc(n, "C", i++), d, d, true, ACT_RELU);
        add_conv(h->audioenc, sc(n, "C", i++), d, d, true, ACT_NONE);
        for (int o = 0; o < 2; ++o)
            for (int j = 0, r = 1; j < 4; ++j, r *= 3) add_hc(h->audioenc, sc(n, "HC", i++), d, 3, r, true);
        for (int o = 0; o < 2; ++o) add_hc(h->audioenc, sc(n, "HC", i++), d, 3, 3, true);
    }
    {   // AudioDec  networks.py:360-435
        const char* n = "Text2Mel/AudioDec";
        int i = 1;
        add_conv(h->audiodec, sc(n, "C", i++), 2 * d, d, true, ACT_NONE);
        // hp.concatenate_query False (networks.py:317-321): R = the context alone and C_1's kernel has d input rows.  The device keeps
        // R' = [ctx | Q] and packs zeros for the Q rows: every kernel of the decode path is unchanged, ctx . Wc + 0 is exact
        if (m.flags & OPH_FLAG_NO_CONCAT_QUERY) h->audiodec.back().cin_var = d;
        h->dec_pre = 1;
        if (m.flags & OPH_FLAG_SPK_AUDIO_DECODER_INPUT) {
            i++;                                      // embed_2
            add_conv(h->audiodec, sc(n, "C", i++), d + m.speaker_embedding_size, d, false, ACT_NONE,
                     m.speaker_embedding_size);
            h->dec_pre = 2;
        }
        for (int j = 0, r = 1; j < 4; ++j, r *= 3) add_hc(h->audiodec, sc(n, "HC", i++), d, 3, r, true);
        for (int o = 0; o < 2; ++o) add_hc(h->audiodec, sc(n, "HC", i++), d, 3, 1, true);
        h->n_hc_dec = 6;
        for (int o = 0; o < 3; ++o) add_conv(h->audiodec, sc(n, "C", i++), d, d, true, ACT_RELU);
        add_conv(h->audiodec, sc(n, "C", i++), d, m.n_mels, true, ACT_NONE);   // sigmoid applied by emit (squash_output_t2m)
    }
    {   // SSRN  networks.py:437-537
        const char* n = "SSRN";
        int i = 1;
        add_conv(h->ssrn, sc(n, "C", i++), m.n_mels, c, false, ACT_NONE);
        if (m.flags & OPH_FLAG_SPK_SSRN_INPUT) {                  // networks.py:457-465: embed_2, concat, C_3 (no activation)
            const std::string es = sc(n, "embed", i++);
            add_conv(h->ssrn, sc(n, "C", i++), c + m.speaker_embedding_size, c, false, ACT_NONE, m.speaker_embedding_size);
            h->ssrn.back().cat_scope = es;
        }
        for (int j = 0, r = 1; j < 2; ++j, r *= 3) add_hc(h->ssrn, sc(n, "HC", i++), c, 3, r, false);
        const int ntr = m.r == 4 ? 2 : 3;
        for (int o = 0; o < ntr; ++o) {
            Layer l;
            l.scope = sc(n, "D", i++); l.kind = K_CONVT; l.cin = c; l.cout = c; l.size = 3;
            h->ssrn.push_back(l);
            for (int j = 0, r = 1; j < 2; ++j, r *= 3) add_hc(h->ssrn, sc(n, "HC", i++), c, 3, r, false);
        }
        add_conv(h->ssrn, sc(n, "C", i++), c, 2 * c, false, ACT_NONE);
        for (int o = 0; o < 2; ++o) add_hc(h->ssrn, sc(n, "HC", i++), 2 * c, 3, 1, false);
        add_conv(h->ssrn, sc(n, "C", i++), 2 * c, m.full_dim, false, ACT_NONE);
        for (int o = 0; o < 2; ++o) add_conv(h->ssrn, sc(n, "C", i++), m.full_dim, m.full_dim, false, ACT_RELU);
        add_conv(h->ssrn, sc(n, "C", i++), m.full_dim, m.full_dim, false, (m.flags & OPH_FLAG_NO_SQUASH_SSRN) ? ACT_NONE : ACT_SIGMOID);   // squash_output_ssrn (networks.py:533-536)
    }
    // hp.norm None concerns Text2Mel only: synthesize() sets hp.norm = 'layer' while it builds SSRNGraph and restores None
    // afterwards (synthesize.py:513-534), so the SSRN of such a config is normalised like any other and its checkpoint
    // holds the SSRN gamma / beta variables
    if (m.flags & OPH_FLAG_NORM_NONE)
        for (auto* net : {&h->textenc, &h->audioenc, &h->audiodec})
            for (Layer& l : *net) l.ln = false;
    if (m.flags & OPH_FLAG_LCC) {
        // the layers the reference passes lcc=/codes= to: all of TextEnc except the 'towards_end' squash conv
        // (networks.py:191-198), all of AudioEnc, AudioDec after its input convs (networks.py:373-389 pass none); SSRN none
        for (Layer& l : h->textenc) l.lcc = l.cat_scope.empty() || (m.flags & OPH_FLAG_SPK_TEXT_ENCODER_INPUT && &l == &h->textenc[0]);
        for (Layer& l : h->audioenc) l.lcc = l.cat_scope.empty();      // the 'audio_encoder_input' conv gets none (networks.py:244-245)
        for (size_t i = (size_t)h->dec_pre; i < h->audiodec.size(); ++i) h->audiodec[i].lcc = true;
    }
    // inventory of TF variables, in graph-creation order
    auto inv = [&](const std::string& name, std::vector<int64_t> shp) { h->inventory.emplace_back(name, shp); };
    auto inv_layers = [&](const std::vector<Layer>& v) {
        for (const Layer& l : v) {
            if (!l.cat_scope.empty()) inv(l.cat_scope + "/lookup_table", {m.nspeakers, m.speaker_embedding_size});
            if (l.kind == K_CONV) {
                inv(l.scope + "/conv1d/kernel", {1, l.cin_var > 0 ? l.cin_var : l.cin, l.cout});
                inv(l.scope + "/conv1d/bias", {l.cout});
                if (l.ln) {
                    inv(l.scope + "/normalize/beta", {l.cout});
                    inv(l.scope + "/normalize/gamma", {l.cout});
                }
                if (l.lcc) inv(l.scope + "/lcc_embed/lookup_table", {m.nspeakers, l.cout});
            } else if (l.kind == K_HC) {
                inv(l.scope + "/conv1d/kernel", {l.size, l.cin, 2 * l.cout});
                inv(l.scope + "/conv1d/bias", {2 * l.cout});
                if (l.ln) {
                    inv(l.scope + "/H1/beta", {l.cout});
                    inv(l.scope + "/H1/gamma", {l.cout});
                    inv(l.scope + "/H2/beta", {l.cout});
                    inv(l.scope + "/H2/gamma", {l.cout});
                }
                if (l.lcc) inv(l.scope + "/lcc_embed/lookup_table", {m.nspeakers, l.cout});
            } else {
                inv(l.scope + "/conv2d_transpose/kernel", {1, 3, l.cout, l.cin});
                inv(l.scope + "/conv2d_transpose/bias", {l.cout});
                inv(l.scope + "/normalize/beta", {l.cout});
                inv(l.scope + "/normalize/gamma", {l.cout});
            }
        }
    };
    inv("Text2Mel/TextEnc/embed_1/lookup_table", {m.vocab, m.e});
    inv_layers(h->textenc);
    inv_layers(h->audioenc);
    if (m.flags & OPH_FLAG_SPK_AUDIO_DECODER_INPUT) {
        // creation order inside AudioDec: C_1, embed_2, C_3, ...
        std::vector<Layer> first(h->audiodec.begin(), h->audiodec.begin() + 1), rest(h->audiodec.begin() + 1, h->audiodec.end());
        inv_layers(first);
        inv("Text2Mel/AudioDec/embed_2/lookup_table", {m.nspeakers, m.speaker_embedding_size});
        inv_layers(rest);
    } else {
        inv_layers(h->audiodec);
    }
    inv_layers(h->ssrn);
}

// ------------------------------------------------------------------ weight packing
const std::vector<float>* getw(oph_handle* h, const std::string& name) {
    auto it = h->hostw.find(name);
    return it == h->hostw.end() ? nullptr : &it->second;
}

float* upload(oph_handle* h, const std::vector<float>& v) {
    float* p = h->dalloc<float>(v.size());
    if (p) hipMemcpyAsync(p, v.data(), v.size() * 4, hipMemcpyHostToDevice, h->stream);
    hipStreamSynchronize(h->stream);     // host vector may be a temporary
    return p;
}
float* upload_padded(oph_handle* h, const std::vector<float>& v, int padto) {
    std::vector<float> t((size_t)round_up((int)v.size(), padto), 0.f);
    std::copy(v.begin(), v.end(), t.begin());
    return upload(h, t);
}

// conv kernel (size, cin, cout) -> Wt[Nalloc][size*kc], k contiguous; tap order = kernel order
std::vector<float> pack_conv(const float* k, int size, int cin, int cout, int kc, int Nalloc) {
    std::vector<float> w((size_t)Nalloc * size * kc, 0.f);
    for (int t = 0; t < size; ++t)
        for (int c = 0; c < cin; ++c) {
            const float* src = k + ((size_t)t * cin + c) * cout;
            for (int n = 0; n < cout; ++n) w[(size_t)n * size * kc + (size_t)t * kc + c] = src[n];
        }
    return w;
}

// ---- every variable as a DEVICE tensor: a slice of the flat buffer oph_set_weights_device was given, or the host copy oph_set_weight
// collected, uploaded into a staging buffer that the next call reuses (in stream order: the pack kernel that reads it is enqueued first)
static const float* dev_tensor(oph_handle* h, const std::string& name, size_t* n_out = nullptr) {
    size_t n = 1, off = 0;
    bool found = false;
    for (const auto& it : h->inventory) {
        n = 1;
        for (int64_t d : it.second) n *= (size_t)d;
        if (it.first == name) { found = true; break; }
        off += n;
    }
    if (!found) return nullptr;
    if (n_out) *n_out = n;
    if (h->d_flat) return h->d_flat + off;
    const std::vector<float>* v = getw(h, name);
    if (!v || v->size() != n) return nullptr;
    if (h->stage_cap < n) {
        if (h->d_stage) { hipStreamSynchronize(h->stream); hipFree(h->d_stage); h->d_stage = nullptr; }
        if (hipMalloc((void**)&h->d_stage, n * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); h->stage_cap = 0; return nullptr; }
        h->stage_cap = n;
    }
    if (hipMemcpyAsync(h->d_stage, v->data(), n * sizeof(float), hipMemcpyHostToDevice, h->stream) != hipSuccess) return nullptr;
    return h->d_stage;
}
// a vector / table, zero-padded to a multiple of `padto` elements (mode 1: LCC gates sigmoid(table), row 0 reading as zeros)
static float* dev_padded(oph_handle* h, const std::string& name, int padto, int mode = 0, int row0 = 0) {
    size_t n = 0;
    const float* src = dev_tensor(h, name, &n);
    if (!src) return nullptr;
    const size_t npad = (n + padto - 1) / padto * padto;
    float* dst = h->dalloc<float>(npad);
    if (dst) launch_pad_copy(src, dst, n, npad, mode, row0, h->stream);
    return dst;
}

int pack_layer(oph_handle* h, Layer& l) {
    l.kc = round_up(l.cin, 32);
    if (l.kind == K_CONVT) {
        // [TF-sem] o[2t] = x[t].Kt[0,0]^T + x[t-1].Kt[0,2]^T ; o[2t+1] = x[t].Kt[0,1]^T   (modules.py:242-250); kernel (1,3,cout,cin)
        l.N = l.cout; l.Nalloc = round_up(l.N, 128); l.ntaps = 2; l.off[0] = 0; l.off[1] = -1;
        l.Wt = h->dalloc<float>((size_t)l.Nalloc * 2 * l.kc);
        l.Wt2 = h->dalloc<float>((size_t)l.Nalloc * l.kc);
        const float* kt = dev_tensor(h, l.scope + "/conv2d_transpose/kernel");
        if (!l.Wt || !l.Wt2 || !kt) return -1;
        launch_pack_convT(kt, l.Wt, l.Wt2, l.cin, l.cout, l.kc, l.Nalloc, h->stream);
        l.bias = dev_padded(h, l.scope + "/conv2d_transpose/bias", l.Nalloc);
        l.g1 = dev_padded(h, l.scope + "/normalize/gamma", 256);
        l.b1 = dev_padded(h, l.scope + "/normalize/beta", 256);
        return (l.bias && l.g1 && l.b1) ? 0 : -1;
    }
    l.N = l.kind == K_HC ? 2 * l.cout : l.cout;
    l.Nalloc = round_up(l.N, 128);
    l.ntaps = l.size;
    for (int t = 0; t < l.size; ++t)   // causal: x[t-(size-1-k)*rate] (modules.py:123-127); SAME: centred [TF-sem]
        l.off[t] = l.causal ? -(l.size - 1 - t) * l.rate : (t - (l.size - 1) / 2) * l.rate;
    {
        const float* k = dev_tensor(h, l.scope + "/conv1d/kernel");
        l.Wt = h->dalloc<float>((size_t)l.Nalloc * l.size * l.kc);
        if (!k || !l.Wt) return -1;
        const int cin_k = l.cin_var > 0 ? l.cin_var : l.cin;      // rows the variable has; the packed copies are zero beyond them
        launch_pack_conv(k, l.Wt, l.size, cin_k, l.N, l.kc, l.Nalloc, h->stream);
        if (l.kind == K_CONV && l.size == 1 && l.N <= 256) {
            l.ldn = round_up(l.N, 4);
            l.Wkn = h->dalloc<float>((size_t)l.kc * l.ldn);
            if (!l.Wkn) return -1;
            launch_pack_wkn(k, l.Wkn, cin_k, l.N, l.kc, l.ldn, h->stream);
        }
    }
    l.bias = dev_padded(h, l.scope + "/conv1d/bias", l.Nalloc);
    if (!l.cat_scope.empty()) {
        l.cat_table = dev_padded(h, l.cat_scope + "/lookup_table", 1);
        if (!l.cat_table) return -1;
    }
    if (l.lcc) {          // (nspeakers, cout) -> gates sigmoid(table), row 0 = sigmoid(0)
        l.lcc_gate = dev_padded(h, l.scope + "/lcc_embed/lookup_table", 1, 1, l.cout);
        if (!l.lcc_gate) return -1;
    }
    if (!l.ln) {
        l.g1 = l.g2 = h->d_ones;
        l.b1 = l.b2 = h->d_zeros;
    } else if (l.kind == K_HC) {
        l.g1 = dev_padded(h, l.scope + "/H1/gamma", 256);
        l.b1 = dev_padded(h, l.scope + "/H1/beta", 256);
        l.g2 = dev_padded(h, l.scope + "/H2/gamma", 256);
        l.b2 = dev_padded(h, l.scope + "/H2/beta", 256);
        if (!l.g2 || !l.b2) return -1;
    } else {
        l.g1 = dev_padded(h, l.scope + "/normalize/gamma", 256);
        l.b1 = dev_padded(h, l.scope + "/normalize/beta", 256);
    }
    return (l.Wt && l.bias && l.g1 && l.b1) ? 0 : -1;
}

extern "C" {

int oph_num_weights(const oph_handle* h) { return h ? (int)h->inventory.size() : OPH_ERR_INVALID; }

int oph_weight_info(const oph_handle* h, int index, char* name, int name_cap, int64_t* shape, int* rank) {
    if (!h || index < 0 || index >= (int)h->inventory.size()) return OPH_ERR_INVALID;
    const auto& it = h->inventory[index];
    if (name && name_cap > 0) { strncpy(name, it.first.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (rank) *rank = (int)it.second.size();
    if (shape) for (size_t i = 0; i < it.second.size() && i < 4; ++i) shape[i] = it.second[i];
    return OPH_OK;
}

int oph_set_weight(oph_handle* h, const char* name, const float* data, const int64_t* shape, int rank) {
    if (!h) return OPH_ERR_INVALID;
    if (!name || !data || !shape) { h->fail("null argument"); return OPH_ERR_INVALID; }
    if (h->finalized) { h->fail("weights already finalized"); return OPH_ERR_STATE; }
    for (const auto& it : h->inventory) {
        if (it.first != name) continue;
        if ((int)it.second.size() != rank) { h->fail("variable %s: rank %d, expected %d", name, rank, (int)it.second.size()); return OPH_ERR_INVALID; }
        size_t n = 1;
        for (int i = 0; i < rank; ++i) {
            if (shape[i] != it.second[i]) { h->fail("variable %s: dim %d is %lld, expected %lld", name, i, (long long)shape[i], (long long)it.second[i]); return OPH_ERR_INVALID; }
            n *= (size_t)shape[i];
        }
        h->hostw[name].assign(data, data + n);
        return OPH_OK;
    }
    h->fail("unknown variable %s", name);
    return OPH_ERR_INVALID;
}

// The whole inventory at once, already on this handle's device: `d_flat` = every variable's float32 data back to back in
// oph_weight_info order (n_floats = their total).  It is repacked in place by device kernels inside oph_finalize_weights -- nothing
// passes through the host -- and must stay valid until that call returns.  Multi-GPU start-up: the RCCL receive buffer of the one
// weight broadcast is handed over as it is.
int oph_set_weights_device(oph_handle* h, const float* d_flat, int64_t n_floats) {
    if (!h) return OPH_ERR_INVALID;
    if (!d_flat) { h->fail("null argument"); return OPH_ERR_INVALID; }
    if (h->finalized) { h->fail("weights already finalized"); return OPH_ERR_STATE; }
    int64_t want = 0;
    for (const auto& it : h->inventory) { int64_t n = 1; for (int64_t d : it.second) n *= d; want += n; }
    if (n_floats != want) { h->fail("flat weight buffer holds %lld floats, the inventory needs %lld", (long long)n_floats, (long long)want); return OPH_ERR_INVALID; }
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, d_flat) != hipSuccess || attr.type != hipMemoryTypeDevice || attr.device != h->device) {
        (void)hipGetLastError();
        h->fail("oph_set_weights_device needs a pointer to memory of device %d", h->device);
        return OPH_ERR_INVALID;
    }
    h->d_flat = d_flat;
    return OPH_OK;
}

static int finalize_weights_impl(oph_handle* h);
int oph_finalize_weights(oph_handle* h) {
    if (!h) return OPH_ERR_INVALID;
    if (h->finalized) return OPH_OK;
    const int rc = finalize_weights_impl(h);
    // whatever happened, the caller's device buffer (oph_set_weights_device) is not referenced beyond this call: after a failure
    // the weights have to be handed over again
    h->d_flat = nullptr;
    return rc;
}
static int finalize_weights_impl(oph_handle* h) {
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->d_flat)
        for (const auto& it : h->inventory)
            if (!h->hostw.count(it.first)) { h->fail("missing variable %s", it.first.c_str()); return OPH_ERR_STATE; }
    {
        const int n = round_up(std::max({2 * h->dm.c, h->dm.full_dim, 2 * h->dm.d, 256}), 256);
        h->d_ones = upload(h, std::vector<float>((size_t)n, 1.f));
        h->d_zeros = upload(h, std::vector<float>((size_t)n, 0.f));
        if (!h->d_ones || !h->d_zeros) { h->fail("out of device memory"); return OPH_ERR_DEVICE; }
    }
    for (auto* net : {&h->textenc, &h->audioenc, &h->audiodec, &h->ssrn})
        for (Layer& l : *net)
            if (pack_layer(h, l) != 0) { h->fail("out of device memory packing %s", l.scope.c_str()); return OPH_ERR_DEVICE; }
    // Range guard of the fp16 split (hi = fp16(w) overflows to inf above 65504, lo = w - hi to nan): trained weights are orders of
    // magnitude below that, but a net that has one falls back to the fp32-operand MFMA instead of propagating NaNs silently.
    // (Small values are safe: the lo term of a tiny weight lands in fp16's subnormals, which the gfx950 MFMA does not flush.)
    {
        unsigned* d_max = h->dalloc<unsigned>(4);
        if (!d_max) { h->fail("out of device memory"); return OPH_ERR_DEVICE; }
        int which = 0;
        auto scan = [&](const std::vector<Layer>& net, size_t from, size_t to) {      // over the packed copies (zero padding does not change a maximum)
            for (size_t i = from; i < to && i < net.size(); ++i) {
                const Layer& l = net[i];
                const size_t taps = l.kind == K_CONVT ? 2 : (size_t)l.ntaps;
                if (l.Wt) launch_maxabs(l.Wt, (size_t)l.Nalloc * taps * l.kc, d_max + which, h->stream);
                if (l.Wt2) launch_maxabs(l.Wt2, (size_t)l.Nalloc * l.kc, d_max + which, h->stream);
            }
            ++which;
        };
        scan(h->ssrn, 0, h->ssrn.size());
        scan(h->audiodec, (size_t)h->dec_pre, (size_t)(h->dec_pre + h->n_hc_dec));
        scan(h->textenc, 0, h->textenc.size());
        unsigned mx[4] = {0, 0, 0, 0};
        HIPCHK(h, hipMemcpyAsync(mx, d_max, sizeof mx, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        auto too_big = [&](int i) { float f; memcpy(&f, &mx[i], 4); return !(f <= 6.0e4f); };
        h->guard_ssrn = too_big(0);
        h->guard_cone = too_big(1);
        h->guard_text = too_big(2);
        if (h->guard_ssrn) h->ssrn_prec = 0;
    }
    // SSRN contractions run on the 16-bit MFMAs with every fp32 operand as hi + lo: the weights are split here, once, into
    // fp16 planes (the default arithmetic) and bf16 planes (oph_set_ssrn_precision(h, 1))
    auto split = [&](const float* wsrc, size_t n, bool f16, void*& hi, void*& lo) {
        hi = h->dalloc<unsigned short>(n); lo = h->dalloc<unsigned short>(n);
        if (!hi || !lo) return false;
        if (f16) launch_split_f16(wsrc, hi, lo, n, h->stream); else launch_split_bf16(wsrc, hi, lo, n, h->stream);
        return true;
    };
    for (Layer& l : h->ssrn) {
        const size_t taps = l.kind == K_CONVT ? 2 : (size_t)l.ntaps;
        const size_t n1 = (size_t)l.Nalloc * taps * l.kc, n2 = (size_t)l.Nalloc * l.kc;
        if (l.Wt && (!split(l.Wt, n1, false, l.Wh, l.Wl) || !split(l.Wt, n1, true, l.Wh16, l.Wl16))) { h->fail("out of device memory"); return OPH_ERR_DEVICE; }
        if (l.Wt2 && (!split(l.Wt2, n2, false, l.Wh2, l.Wl2) || !split(l.Wt2, n2, true, l.Wh2_16, l.Wl2_16))) { h->fail("out of device memory"); return OPH_ERR_DEVICE; }
    }
    // plane_gemm reads the fp16 planes K-blocked (batched nets only: SSRN here, TextEnc below)
    auto kblock = [&](const void* src, int rows, int ld, void*& dst) {
        dst = h->dalloc<unsigned short>((size_t)rows * ld);
        if (!dst) return false;
        launch_kblock_planes(src, dst, rows, ld, h->stream);
        return true;
    };
    for (Layer& l : h->ssrn) {
        const int taps = l.kind == K_CONVT ? 2 : l.ntaps;
        if (l.Wh16 && (!kblock(l.Wh16, l.Nalloc, taps * l.kc, l.Wkh) || !kblock(l.Wl16, l.Nalloc, taps * l.kc, l.Wkl))) { h->fail("out of device memory"); return OPH_ERR_DEVICE; }
        if (l.Wh2_16 && (!kblock(l.Wh2_16, l.Nalloc, l.kc, l.Wkh2) || !kblock(l.Wl2_16, l.Nalloc, l.kc, l.Wkl2))) { h->fail("out of device memory"); return OPH_ERR_DEVICE; }
    }
    // The two many-row levels of the AudioDec history cone (1312 and 704 rows x 768 x 512 per step) on the split contraction.
    // Text2Mel feeds an argmax back into itself, so only fp32-class arithmetic qualifies as its default: split-fp16 x3
    // (22 significant bits per operand; measured against the fp32 MFMA flavour in tests/test_gpu_decode_modes.py) -- the
    // split-bf16 flavour (16 bits) stays an experiment (OPH_CONE_PREC=1).
    h->cone_prec = h->guard_cone ? 0 : (h->opt.cone_prec >= 0 ? h->opt.cone_prec : CONE_PREC_DEFAULT);
    for (int k = 0; k + 1 < h->n_hc_dec; ++k) {
        Layer& l = h->audiodec[h->dec_pre + k];
        const size_t n = (size_t)l.Nalloc * l.ntaps * l.kc;
        if (!split(l.Wt, n, true, l.Wh16, l.Wl16) || (h->cone_prec == 1 && !split(l.Wt, n, false, l.Wh, l.Wl))) { h->fail("out of device memory"); return OPH_ERR_DEVICE; }
    }
    // TextEnc (82 GFLOP per 16-utterance batch, once per batch) on the split-fp16 contraction as well: K,V feed the attention
    // argmax, so again only the fp32-class flavour is offered (oph_set_precision(h, 2, 0) selects the fp32 MFMA)
    h->textenc_prec = h->guard_text ? 0 : (h->opt.textenc_prec >= 0 ? h->opt.textenc_prec : TEXTENC_PREC_DEFAULT);
    for (Layer& l : h->textenc)
        if (!split(l.Wt, (size_t)l.Nalloc * l.ntaps * l.kc, true, l.Wh16, l.Wl16) || !kblock(l.Wh16, l.Nalloc, l.ntaps * l.kc, l.Wkh) ||
            !kblock(l.Wl16, l.Nalloc, l.ntaps * l.kc, l.Wkl)) { h->fail("out of device memory"); return OPH_ERR_DEVICE; }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->cone_head_ok = !h->opt.no_cone_head && !(h->dm.flags & OPH_FLAG_NO_MONOTONIC) && h->audiodec[0].Wkn != nullptr && h->dm.d <= 256 && (h->dm.d % 4) == 0;
    if (h->cone_head_ok) {
        // Wc = the rows of AudioDec C_1's kernel (1, 2d, d) that multiply the attention context (R' = [ctx | Q], networks.py:316-319)
        const Layer& c1 = h->audiodec[0];
        const int d = h->dm.d;
        h->kc_c = round_up(d, 32); h->ldvw = round_up(d, 128);
        h->Wt_c = h->dalloc<float>((size_t)h->ldvw * h->kc_c);
        const float* k = dev_tensor(h, c1.scope + "/conv1d/kernel");
        if (!h->Wt_c || !k) { h->fail("out of device memory"); return OPH_ERR_DEVICE; }
        launch_pack_wtc(k, h->Wt_c, d, h->kc_c, h->ldvw, h->stream);
    }
    // hc_fused: the cone's levels as one launch each.  Kernel of AudioDec highway layer k as planes with the output columns permuted
    // per 64-tile to [32 H1 channels | the same 32 channels of H2]
    {
        const int pre = h->dec_pre, nh = h->n_hc_dec, d = h->dm.d;
        bool ok = !h->opt.no_fused_cone && h->cone_head_ok && pre == 1 && d == 256 && !(h->dm.flags & (OPH_FLAG_LCC | OPH_FLAG_NORM_NONE | OPH_FLAG_NO_MONOTONIC)) &&
                  nh >= 2 && !h->guard_cone;
        for (int k = 0; ok && k + 1 < nh; ++k) {
            const Layer& l = h->audiodec[pre + k];
            ok = l.kind == K_HC && l.ntaps == 3 && l.kc == 256 && l.cout == 256 && l.cin == 256 && l.ln && !l.lcc && l.ccat == 0 && l.causal;
        }
        for (int k = 0; ok && k + 1 < nh; ++k) {
            Layer& l = h->audiodec[pre + k];
            float* dwp = h->dalloc<float>((size_t)512 * 768);      // [column tile jt][K-step ks][64 columns][64 k]
            l.bias_p = h->dalloc<float>(512);
            float* dbs = dev_padded(h, l.scope + "/conv1d/bias", 1);
            const float* kr = dev_tensor(h, l.scope + "/conv1d/kernel");      // (3, 256, 512)
            if (!dwp || !l.bias_p || !dbs || !kr) { h->fail("out of device memory"); return OPH_ERR_DEVICE; }
            launch_pack_hcf(kr, dbs, dwp, l.bias_p, h->stream);
            if (!split(dwp, (size_t)512 * 768, true, l.Wph, l.Wpl)) { h->fail("out of device memory"); return OPH_ERR_DEVICE; }
        }
        h->cone_fused_ok = ok;
    }
    h->emb_text = dev_padded(h, "Text2Mel/TextEnc/embed_1/lookup_table", 1);
    if (h->dm.flags & OPH_FLAG_SPK_AUDIO_DECODER_INPUT) h->emb_spk = dev_padded(h, "Text2Mel/AudioDec/embed_2/lookup_table", 1);
    if (!h->emb_text || ((h->dm.flags & OPH_FLAG_SPK_AUDIO_DECODER_INPUT) && !h->emb_spk)) { h->fail("out of device memory"); return OPH_ERR_DEVICE; }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    h->hostw.clear();
    if (h->d_stage) { hipFree(h->d_stage); h->d_stage = nullptr; h->stage_cap = 0; }
    h->d_flat = nullptr;          // the caller's buffer is no longer referenced
    h->pool = 1;          // everything allocated from here on is per-batch-size state
    h->use_run = run_supported(h);
    // OPH_DECODE = loop (default where possible) | runs (two launches per step) | layers (one launch per layer, round 1).
    // The whole-decode launch needs its own CU partition (all its workgroups resident while the cone runs beside it) and
    // the mapped progress words.
    h->use_loop = h->use_run && h->opt.decode == 0 && h->d_sig && h->host_prog && h->mask_words > 0;
    h->use_loop_wanted = h->use_loop; h->use_run_wanted = h->use_run;
    h->finalized = true;
    return OPH_OK;
}

}  // extern "C"
