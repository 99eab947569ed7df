// Host-side declarations shared by the translation units of libophelia_hip.so that assemble the model, run the batched nets and
// the decode loop, and export the C ABI (include/ophelia_hip.h): the layer / option / tile / handle structures, the launch
// accounting, and the functions that cross file boundaries.  Internal: nothing here is part of the ABI.
//   oph_model.hip   network description (networks.py:121-537), weight packing, oph_set_weight / oph_finalize_weights
//   oph_nets.hip    batched nets (TextEnc, SSRN, graph evaluation): conv-GEMM + LayerNorm per layer, streamed SSRN
//   oph_cone.hip    the AudioDec history cone of one step (side stream)
//   oph_decode.hip  decode state, tiles, the whole-decode launch and the per-step launch paths
//   oph_api.hip     handle lifetime and the session / resident / measurement entry points of the C ABI
//   oph_ops.hip     per-operator entry points (unit parity) and the conv1d_transpose timing probe
#pragma once
#include "oph_internal.h"
#pragma GCC visibility push(default)      // the library is built with -fvisibility=hidden: only the C ABI is exported
#include "../../include/ophelia_hip.h"
#pragma GCC visibility pop

#include <algorithm>
#include <chrono>
#include <ctime>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace oph;

extern thread_local std::string g_create_error;
extern thread_local hipStream_t g_cur;    // stream the launch wrappers of THIS host thread target
extern thread_local int g_group_cls;
extern double g_host_us[4];     // OPH_TRACE: host time spent enqueuing {event ops, cone, critical launches, other}
struct HostTimer {
    int slot; std::chrono::steady_clock::time_point t0; bool on;
    HostTimer(int s, bool enable) : slot(s), on(enable) { if (on) t0 = std::chrono::steady_clock::now(); }
    ~HostTimer() { if (on) g_host_us[slot] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6; }
};
extern const bool g_trace;
#define TRACE(...) do { if (g_trace) { fprintf(stderr, "[oph] " __VA_ARGS__); fputc('\n', stderr); fflush(stderr); } } while (0)
extern thread_local std::string g_op_error;

#define HIPCHK(h, expr)                                                                        \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (h)->fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return OPH_ERR_DEVICE;                                                             \
        }                                                                                      \
    } while (0)


enum Kind { K_CONV = 0, K_HC = 1, K_CONVT = 2 };

struct Layer {
    std::string scope;
    int kind = K_CONV;
    int cin = 0, cout = 0;       // cout = filters (hc: C ; raw conv output has 2C columns)
    int size = 1, rate = 1;
    bool causal = false;
    int act = ACT_NONE;
    int ccat = 0;                // speaker-embedding channels concatenated to the input (cin includes them)
    int cin_var = 0;             // > 0: input channels of the TF VARIABLE when fewer than cin -- hp.concatenate_query False: AudioDec C_1's kernel is
                                 // (1, d, d) and is packed into the first d of the layer's 2d input channels (the query half multiplies zeros)
    bool ln = true;              // false: hp.norm None -> no gamma/beta variables, normalisation is the identity
    std::string cat_scope;       // TextEnc layers with ccat: TF scope of the speaker lookup table concatenated to the input
    float* cat_table = nullptr;
    bool lcc = false;            // learned channel contributions: variable <scope>/lcc_embed/lookup_table (nspeakers, cout)
    float* lcc_gate = nullptr;   // device table [nspeakers][cout] = sigmoid(lookup_table), row 0 = sigmoid(0) (embed zero_pad)
    // packed
    int kc = 0, N = 0, Nalloc = 0, ntaps = 1;
    int off[3] = {0, 0, 0};
    float *Wt = nullptr, *bias = nullptr;       // conv / hc ; convT: even phase (taps x[t], x[t-1])
    float *Wt2 = nullptr;                        // convT odd phase (tap x[t])
    float *Wkn = nullptr; int ldn = 0;           // k=1 layers with N<=256: [kc][ldn] n-contiguous copy for row_chain
    void *Wh = nullptr, *Wl = nullptr, *Wh2 = nullptr, *Wl2 = nullptr;   // SSRN layers: Wt / Wt2 split into hi + lo bf16 planes (conv_gemm_bf16x3)
    void *Wh16 = nullptr, *Wl16 = nullptr, *Wh2_16 = nullptr, *Wl2_16 = nullptr;   // the same as fp16 planes (split-fp16 x3: fp32-class accuracy)
    void *Wkh = nullptr, *Wkl = nullptr, *Wkh2 = nullptr, *Wkl2 = nullptr;         // the fp16 planes K-blocked [ntaps kc / 32][Nalloc][32] (plane_gemm)
    void *Wph = nullptr, *Wpl = nullptr; float* bias_p = nullptr;   // AudioDec highway layers: kernel as fp16 planes [2C][3 kc] with the columns
                                                                     // permuted per 64-tile to [32 H1 | the same 32 channels of H2] (hc_fused)
    float *g1 = nullptr, *b1 = nullptr, *g2 = nullptr, *b2 = nullptr;
};

struct FcTables { bool valid; short tab[3][CONE_FC_MAXOUT], need[3][CONE_FC_MAXOUT], res[3][CONE_FC_MAXOUT]; short extra[CONE_FC_MAXEXTRA], extra_res[CONE_FC_MAXEXTRA]; int n_extra; };

struct ProfClass {
    const char* name;
    long long launches = 0;
    double bytes = 0, flops = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    size_t used = 0;
    double ms = 0;
};
constexpr int TEXTENC_PREC_DEFAULT = 2;   // TextEnc contractions when OPH_TEXTENC_PREC is not set
constexpr int CONE_PREC_DEFAULT = 2;      // arithmetic of the cone's two many-row contractions when OPH_CONE_PREC is not set (see oph_finalize_weights)
constexpr int CONE_KSPLIT = 4;   // split-K of the latency-bound decoder-cone GEMMs (partials summed by ln_rows); buffers are sized for it

// Launch-path and arithmetic options of a handle.  They are NOT read from the environment: a production build takes them only from
// the option string of oph_create_opts ("NAME=value NAME ...", names below without any prefix; oph_create = no options), so an
// inherited environment variable cannot change what a handle computes or how fast.  Every option here selects a launch path the
// library also takes by itself (non-standard geometries, the recovery ladder) or an arithmetic flavour of oph_set_precision: results
// stay within the cross-flavour bar (tests/test_gpu_decode_modes.py).  Measurement builds (-DOPH_ABLATE, ophelia_amd/_lib.py with
// OPH_HIPCC_FLAGS) additionally accept the ablation options -- SKIP_CONE, LOOP_ALONE, LOOP_DBG: wrong or unused results -- and read
// OPH_<NAME> from the environment for every name; a production build refuses the ablation names and never calls getenv for options.
struct Options {
    int decode = 0;                  // DECODE = loop (0, default where possible) | runs (1: two launches per step) | layers (2: one launch per layer)
    int run_rows = 8;                // RUN_ROWS: utterance rows per workgroup of dec_loop (8 or 4)
    // split-K of the cone GEMMs: many-row levels have enough tiles to fill the cone's CUs with less splitting (fewer partials
    // to write and re-read).  CONE_KSPLIT="big,small", each 1..4; measured (profiles/r02): 3,4 best
    int ksplit_big = 3, ksplit_small = CONE_KSPLIT;
    int fc_rows = -1, fc_insplit = 2;   // CONE_FC_ROWS (cone levels of at most this many rows run as cone_fc16), CONE_FC_INSPLIT
    int lookahead = 8;               // LOOP_LOOKAHEAD: cones the host may queue ahead of the loop kernel's progress
    int loop_dbg = 0;                // -DOPH_ABLATE only: LOOP_ALONE -> 32 (the whole-decode launch without its side stream: counter passes; mel unused),
                                     // LOOP_DBG = the other ablation bits of dec_loop
    int cu_dec = 0, cu_cone = 0;     // CU_SPLIT="chain,cone" CUs of the three partitions (rest: SSRN)
    bool no_cu_mask = false;         // NO_CU_MASK
    bool no_cone_head = false, no_loop_qw = false, no_preencode = false, no_stream_ssrn = false;
    int cone_prec = -1;              // CONE_PREC: the two many-row cone contractions: 0 fp32 MFMA, 1 split-bf16 x3 (experiment), 2 split-fp16 x3; -1 = the default
    int ssrn_prec = -1;              // SSRN_PREC: 0 fp32 MFMA, 1 split-bf16 x3, 2 split-fp16 x3; -1 = the default
    int textenc_prec = -1;           // TEXTENC_PREC: 0 fp32 MFMA, 2 split-fp16 x3; -1 = the default
    bool skip_cone = false;          // -DOPH_ABLATE only: SKIP_CONE: timing experiments, results are wrong
    bool stream_value = false;       // STREAM_VALUE: per-step launch paths chain their two streams with stream value operations
    bool run_stamps = false;         // RUN_STAMPS: clock stamps of the decode kernels' phases (printed under OPH_TRACE); takes dec_loop
    int ssrn_chunk = 40;             // SSRN_CHUNK: mel frames per streamed SSRN chunk (0 = SSRN only after the decode)
    bool no_fused_cone = false;      // NO_FUSED_CONE: the cone's levels as contraction + ln_rows launches instead of hc_fused
    int pg_waves = 0;                // PG_WAVES=4|8: the transposed convolution's plane_gemm form forced (64 channels per workgroup on 4 waves | 128 on 8;
                                     // 0 = the launcher's choice); in -DOPH_ABLATE builds 8 also selects the 8-wave forms of the other layers
    bool no_plane_gemm = false;      // NO_PLANE_GEMM: the batched nets' split-fp16 contractions on fp32 rows (conv_gemm_bf16x3) instead of planes (plane_gemm)
    bool no_chain = false;           // NO_CHAIN: the whole-decode launch as dec_loop (generic) even where dec_chain (specialised) fits
    bool fake_ln_timeout = false;    // -DOPH_ABLATE only: FAKE_LN_TIMEOUT: the fused conv1d_transpose launch raises its time-out word although nothing timed out
                                     // (fault injection for the redo path: tests/test_gpu_second_client.py)
    bool no_fused_convt_ln = false;  // NO_FUSED_CONVT_LN: conv1d_transpose as plane_gemm + ln_rows (two launches, round 5) instead of LayerNorm inside the launch
    // spec: the option string or NULL.  Returns false (and says why) on a name this build does not know.
    bool read(const char* spec, std::string* why) {
        static const char* const known[] = {"DECODE", "RUN_ROWS", "CONE_KSPLIT", "CONE_FC_ROWS", "CONE_FC_INSPLIT", "LOOP_LOOKAHEAD", "CU_SPLIT", "NO_CU_MASK", "NO_CONE_HEAD",
                                            "NO_LOOP_QW", "NO_PREENCODE", "NO_STREAM_SSRN", "CONE_PREC", "SSRN_PREC", "TEXTENC_PREC", "STREAM_VALUE", "RUN_STAMPS", "SSRN_CHUNK",
                                            "NO_FUSED_CONE", "PG_WAVES", "NO_PLANE_GEMM", "NO_CHAIN", "NO_FUSED_CONVT_LN",
#ifdef OPH_ABLATE
                                            "SKIP_CONE", "LOOP_ALONE", "LOOP_DBG", "FAKE_LN_TIMEOUT",
#endif
        };
        std::map<std::string, std::string> kv;
        for (const char* p = spec ? spec : ""; *p;) {
            while (*p == ' ' || *p == ';' || *p == '\t' || *p == '\n') ++p;
            const char* e = p;
            while (*e && *e != ' ' && *e != ';' && *e != '\t' && *e != '\n') ++e;
            if (e == p) break;
            std::string tok(p, e);
            const size_t eq = tok.find('=');
            const std::string name = tok.substr(0, eq), val = eq == std::string::npos ? std::string("1") : tok.substr(eq + 1);
            bool ok = false;
            for (const char* k : known) ok = ok || name == k;
            if (!ok) { if (why) *why = "unknown option '" + name + "' (oph_create_opts)"; return false; }
            kv[name] = val;
            p = e;
        }
#ifdef OPH_ABLATE
        for (const char* k : known)
            if (!kv.count(k)) { const std::string en = std::string("OPH_") + k; if (const char* e = getenv(en.c_str())) kv[k] = e; }
#endif
        auto str = [&](const char* n) -> const char* { auto it = kv.find(n); return it == kv.end() ? nullptr : it->second.c_str(); };
        auto flag = [&](const char* n) { return str(n) != nullptr; };
        auto num = [&](const char* n, int dflt) { const char* e = str(n); return e ? atoi(e) : dflt; };
        if (const char* m = str("DECODE")) decode = !strcmp(m, "runs") ? 1 : (!strcmp(m, "layers") ? 2 : 0);
        run_rows = num("RUN_ROWS", 8) == 4 ? 4 : 8;
        if (const char* e = str("CONE_KSPLIT")) { int a_ = 0, b_ = 0; if (sscanf(e, "%d,%d", &a_, &b_) == 2 && a_ >= 1 && a_ <= CONE_KSPLIT && b_ >= 1 && b_ <= CONE_KSPLIT) { ksplit_big = a_; ksplit_small = b_; } }
        fc_rows = num("CONE_FC_ROWS", -1); fc_insplit = std::max(1, num("CONE_FC_INSPLIT", 2));
        lookahead = num("LOOP_LOOKAHEAD", 8);
        loop_dbg = num("LOOP_DBG", flag("LOOP_ALONE") ? 32 : 0);        // (names a production build has refused above)
        skip_cone = flag("SKIP_CONE"); fake_ln_timeout = flag("FAKE_LN_TIMEOUT");
        if (const char* sp = str("CU_SPLIT")) { int a_ = 0, b_ = 0; if (sscanf(sp, "%d,%d", &a_, &b_) == 2 && a_ > 0 && b_ > 0) { cu_dec = a_; cu_cone = b_; } }
        no_cu_mask = flag("NO_CU_MASK");
        no_cone_head = flag("NO_CONE_HEAD"); no_loop_qw = flag("NO_LOOP_QW"); no_preencode = flag("NO_PREENCODE");
        no_stream_ssrn = flag("NO_STREAM_SSRN");
        cone_prec = num("CONE_PREC", -1); if (cone_prec > 2) cone_prec = -1;
        ssrn_prec = num("SSRN_PREC", -1); if (ssrn_prec > 2) ssrn_prec = -1;
        textenc_prec = num("TEXTENC_PREC", -1); if (textenc_prec != 0 && textenc_prec != 2) textenc_prec = -1;
        stream_value = num("STREAM_VALUE", 0) != 0;
        run_stamps = flag("RUN_STAMPS");
        ssrn_chunk = std::max(0, num("SSRN_CHUNK", 40));
        no_fused_convt_ln = flag("NO_FUSED_CONVT_LN");
        no_chain = flag("NO_CHAIN"); no_fused_cone = flag("NO_FUSED_CONE"); no_plane_gemm = flag("NO_PLANE_GEMM"); pg_waves = num("PG_WAVES", 0); if (pg_waves != 4 && pg_waves != 8) pg_waves = 0;
        return true;
    }
    int cone_ksplit(int M) const { return M >= 512 ? ksplit_big : ksplit_small; }
};
enum { PC_GEMM = 0, PC_GEMM64, PC_GEMM_BF16, PC_LN, PC_DEC, PC_ROWCHAIN, PC_ATTN_ROWS, PC_MISC, PC_DECRUN, PC_DECLOOP, PC_CONEHEAD, PC_HCFUSED, PC_PLANEGEMM, PC_COUNT };   // PC_GEMM = the <128,128> instance


// Decode state of one 16-utterance tile that has to survive between decode calls on the same utterances (a batch of
// more than 16 utterances decodes tile by tile; oph_decode_steps resumes tiles at the step they stopped).  Everything a
// step only uses as scratch (raw rows, cone buffers, granules) is shared by the tiles and lives in the handle.
struct Tile {
    int *d_p = nullptr, *d_ctl = nullptr, *d_ptab = nullptr;    // prev_max ping-pong [2][16]; ctl[0]=n_ended ctl[1]=stop_after ctl[2]=error ctl[3]=attention arrivals
    float *Ytm = nullptr, *Qhist = nullptr, *VW = nullptr, *QWhist = nullptr;
    std::vector<float*> ae_hist;            // AudioEnc per-highway-layer input history
    unsigned* d_loop_layers = nullptr;      // dec_loop's packed layer descriptors (they hold this tile's history pointers)
    int steps = 0;                          // decoder steps executed so far on this tile's utterances
    int ssrn_done = 0;                      // mel frames whose SSRN output is up to date (streamed SSRN)
    int z_copied = 0;                       // mel frames whose SSRN rows have been copied to this batch's host destination (z_host);
                                            // <= ssrn_done: chunks computed while no destination was set (a resumed decode) are not copied
};

struct oph_handle {
    oph_dims dm{};
    Options opt;
    int device = 0;
    hipStream_t stream = nullptr;      // API stream (unmasked): TextEnc / SSRN, timers, copies
    hipStream_t sdec = nullptr;        // decode critical path: CU-masked to a private slice of every XCD
    hipStream_t scone = nullptr;       // side stream: AudioDec history cone, overlapped with the AudioEnc chain
    hipStream_t sssrn = nullptr;       // SSRN partition: streamed SSRN chunks of the running decode, pipelined SSRN tails, the next batch's TextEnc
    hipStream_t scopy = nullptr;       // copies only (unmasked): results leave for the host while the decode runs
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    hipEvent_t ev_dec_done = nullptr, ev_ssrn_done[2] = {nullptr, nullptr}, ev_copy = nullptr, ev_chunk = nullptr;
    // streamed SSRN chunks are launched only while the SSRN partition keeps up: one chunk in flight, and none that could not
    // finish before the decode does (what is left then runs on the whole chip)
    hipEvent_t ev_cs = nullptr, ev_ce = nullptr; bool chunk_inflight = false; float chunk_ms = 0.f;
    std::chrono::steady_clock::time_point dec_t0;
    bool ssrn_inflight[2] = {false, false};
    int buf = 0; bool pipelined = false;
    uint32_t m_cone[16] = {0}, m_conep[16] = {0}, m_ssrn[16] = {0}; int mask_words = 0;   // CU partitions (0 words = no masking)
    bool masked_borrowed = false;      // sdec / scone / sssrn are the device's process-wide CU-masked streams (oph_api.hip): returned, not destroyed
    int ssrn_prec = 2;                 // SSRN contractions: 2 = split-fp16 x3 (fp32 accumulate, fp32-class accuracy), 1 = split-bf16 x3, 0 = fp32 MFMA
    hipEvent_t ev_attn = nullptr, ev_cone = nullptr;
    // The two per-step cross-stream dependencies (cone(t+1) after row_chain B(t); AudioDec(t) after cone(t)) as stream
    // write-value / wait-value operations on two device words instead of event record / wait pairs: an event operation
    // interleaved with launches costs the host ~15 us and the device ~10 us on this runtime, a stream-value operation
    // ~4 us / ~2 us (profiles/launch_rate_probe.hip).  Values grow monotonically: sig_base + step.
    bool fixed_att = false;             // the current decode uses d_ptab instead of the attention softmax
    uint32_t* d_sig = nullptr;          // [0] attention of step t done (written on sdec), [16] cone of step t done (scone), [LOOP_SIG_LEVEL0 + 16 k] cone level k done
    uint32_t sig_base = 0;
    bool use_sigval = false;
    bool can_sigval = false;            // stream value operations work on this device
    // persistent runs of decoder layers (oph_decrun.hip): two launches per step instead of nineteen
    bool use_run = false;               // this configuration takes the dec_run path
    unsigned long long* d_gbuf = nullptr;   // hand-off granules [LOOP_MAX_LAYERS][16][RUN_GCOLS]
    float* d_vbuf = nullptr;                // dec_chain's tag-free hand-off slots [2][LOOP_MAX_LAYERS][16][RUN_GCOLS]
    uint32_t run_epoch = 0;             // advanced per launch: a tag value is never reused
    long long* d_sigdbg = nullptr;      // OPH_RUN_STAMPS diagnostics: [max_T][8] stamps of the cross-stream signals
    long long* d_lvldbg = nullptr;      // ... and [max_T][8]: when the side stream completed cone level k of step t
    long long* d_stamps = nullptr;      // OPH_RUN_STAMPS diagnostics: [2 launches][32 slices][LOOP_MAX_LAYERS][8]
    // whole-decode persistent launch (dec_loop): static layer descriptions in device memory, progress words in pinned host memory
    bool use_loop = false;
    std::vector<LoopLayer> loop_proto;  // the decode's layers with pre-swizzled weights; per tile only the history pointers differ
    std::vector<float*> loop_lnp;
    int loop_nlayers = 0, loop_attn = 0, loop_slices = 0, loop_kmax = 0;
    volatile int* host_prog = nullptr;  // [0] last step whose attention is done  [1] stop step or INT_MAX
    int ndec_cus = 0;                   // CUs the critical stream may use (its CU mask, or the whole chip)
    int loop_capacity = -1;             // workgroups of dec_loop that can be resident at once (-1: not yet asked)
    int loop_rows = 8;                  // rows (utterances) per workgroup of dec_loop: 8 (default) or 4 (OPH_RUN_ROWS)
    bool chain_ok = false;              // the decode's geometry fits dec_chain (oph_decchain.hip), the specialised whole-decode launch
    long long* d_clk = nullptr;         // device-side witness of the whole-decode launches: [2 k] first workgroup in, [2 k + 1] last workgroup out (100 MHz clock)
    int clk_used = 0; long long clk_launches = 0; double clk_total_us = 0;     // launches not yet read back; accumulated over read-back ones
    std::string err;
    bool finalized = false;
    // expected variables (TF names) and host copies
    std::vector<std::pair<std::string, std::vector<int64_t>>> inventory;
    std::map<std::string, std::vector<float>> hostw;
    const float* d_flat = nullptr;      // oph_set_weights_device: every variable back to back on the device, inventory order (until finalize)
    float* d_stage = nullptr; size_t stage_cap = 0;      // host-collected variables pass through this staging buffer on their way to the pack kernels
    // networks
    std::vector<Layer> textenc, audioenc, audiodec, ssrn;
    float* emb_text = nullptr;       // (vocab, e)
    float* emb_spk = nullptr;        // (nspeakers, spk_emb)   AudioDec/embed_2
    float *d_ones = nullptr, *d_zeros = nullptr;   // gamma / beta stand-ins of layers without LayerNorm (hp.norm None)
    // batched workspaces
    int capB = 0;
    float *actA = nullptr, *actB = nullptr, *raw = nullptr;   // workspace of the API stream (TextEnc, host-buffer SSRN)
    float *actA2 = nullptr, *actB2 = nullptr, *raw2 = nullptr; // workspace of the SSRN-partition stream
    size_t act_elems = 0, raw_elems = 0;
    // the activation buffers once more as fp16 hi / lo planes, K-blocked [channel / 32][rows][32] (plane_gemm's operand): [workspace][A | B][hi | lo]
    void* actP[2][2][2] = {{{nullptr, nullptr}, {nullptr, nullptr}}, {{nullptr, nullptr}, {nullptr, nullptr}}};
    // conv1d_transpose with its LayerNorm inside the launch (plane_gemm<.., LNF>): exchange regions per workspace set (the API stream's
    // and the SSRN partition's launches may overlap), launch tags, and the error word (pinned: host_prog[8]) a timed-out exchange raises
    float* d_pg_stats[2] = {nullptr, nullptr}; size_t pg_stats_bytes = 0; unsigned pg_epoch = 0; int* d_pg_err = nullptr; bool pg_ln_off = false;
    long long* d_amax = nullptr;        // oph_text2mel_graph: argmax per (utterance, frame)
    // ---- the staged batch: nB utterances, resident in HBM, utterance-major.  Text is double-buffered so that the NEXT
    // batch can be staged (oph_stage_text_next) and pre-encoded while this one decodes.
    int nB = 0, nBpad = 0;
    int *bL[2] = {nullptr, nullptr}, *bEnds[2] = {nullptr, nullptr}, *bSpk[2] = {nullptr, nullptr}; int txt = 0;     // bX[txt]: current text
    int next_B = 0; bool next_staged = false;      // bX[txt ^ 1] holds a staged next batch of next_B utterances
    bool txt_ran = false;               // the current text has been through a run (a staged next text may take its place)
    bool kv_pre = false;                // bKV[kv_cur] already holds the current text's K,V (pre-encoded while the previous batch decoded)
    long long n_textenc = 0, n_preenc_used = 0, n_chunks_streamed = 0, n_loop_decodes = 0, n_loop_fallbacks = 0, n_tile_resumes = 0;   // oph_get_counters
    int last_wait_err = 0;             // ctl[2] of the last failed decode (1 hand-off, 2 cone level, 3 attention signal, 4 hc_fused statistics)
    bool use_loop_wanted = false, use_run_wanted = false, hcf_capacity_was_ok = false;
    int degraded_left = 0, degraded_next = 16;     // decodes until the fast paths are tried again after a recovery (decode_batch)
    long long n_recoveries = 0;        // decodes in which a co-residency time-out was recovered on the per-step path (oph_get_counters[9])
    int* bTends = nullptr;
    float* bKV[2] = {nullptr, nullptr}; int kv_cur = 0;      // K | V rows [nB][max_N][2d]; the other buffer receives the next batch's pre-encode
    float *bYout[2] = {nullptr, nullptr}, *bZ[2] = {nullptr, nullptr}, *bAlign = nullptr;      // Y / Z ping-pong over pipelined batches
    hipEvent_t ev_preenc = nullptr; bool preenc_valid = false;     // bKV[kv_cur ^ 1] holds (or will hold, after ev_preenc) the K,V of the staged next text
    bool want_preenc = false;           // set around a decode: queue the next batch's TextEnc once the loop kernel is launched
    // residency of the three session calls (oph_encode_text -> oph_text2mel -> oph_ssrn): a NULL K/V (Y) argument means "what
    // the previous call left in HBM"
    bool kv_resident = false, y_resident = false;
    bool spec_ssrn = true;              // oph_text2mel streams SSRN over the frames it has produced (consumed by oph_ssrn(Y = NULL))
    float* z_host = nullptr;            // host destination the streamed SSRN chunks are copied to as they complete (or null)
    float* z_spec = nullptr;            // oph_set_mag_destination: where oph_text2mel's speculative SSRN copies its chunks
    unsigned long long batch_gen = 0;   // advanced by every decode that starts at step 0 (begin_batch)
    unsigned long long z_spec_gen = 0;  // the batch whose speculative SSRN streamed into z_spec (0: none)
    int dec_tbegin = 0, dec_tend = 0;   // step range of the running / last whole-decode launch (the chunk scheduler's time estimate)
    bool guard_ssrn = false, guard_cone = false, guard_text = false;      // a weight of that net is outside fp16's range (|w| > 6e4: hi = inf, lo = nan): its
                                                                          // split contractions are pinned to the fp32-operand MFMA (oph_get_counters [7])
    // ---- decode tiles: utterances [16 j, 16 j + 16) of the batch; `tile` is the one the views below point into
    std::vector<Tile> tiles; int tile = 0;
    int B = 0, Bpad = 0;                // the CURRENT tile: utterances, rows (16)
    int *d_L = nullptr, *d_ends = nullptr, *d_spk = nullptr, *d_p = nullptr, *d_tends = nullptr, *d_ctl = nullptr, *d_ptab = nullptr;
    float *KV = nullptr, *Yout = nullptr, *Ytm = nullptr, *align = nullptr, *Z = nullptr;
    float *Qhist = nullptr, *Rrow = nullptr;
    unsigned* d_loop_layers = nullptr;
    std::vector<float*> ae_hist, ae_raw;          // AudioEnc per-layer input history (tile) / raw outputs (scratch)
    std::vector<float*> ad_raw, ad_xrow;          // AudioDec row chain (scratch)
    // AudioDec history cone
    int n_hc_dec = 0, dec_pre = 0;                // #hc layers, #k=1 layers before them
    std::vector<std::vector<int>> Hset;           // Hset[h] sorted offsets (>=1) at which hc layer h's INPUT is needed
    std::vector<int*> d_tab, d_need, d_res;       // per hc layer h<n-1: tables for computing layer h over Hset[h+1]
    std::vector<FcTables> fc_tab;                 // per hc layer: cone_fc16's index tables (kernel arguments)
    int cone_prec = 0;                           // the two many-row cone contractions: 0 fp32 MFMA, 1 split-bf16 x3, 2 split-fp16 x3
    int textenc_prec = 0;                        // TextEnc contractions: 0 fp32 MFMA, 2 split-fp16 x3
    bool qw_from_loop = false;                   // this decode's QW cache is filled by the loop kernel (cone_head computes nothing)
    float* coneRawB = nullptr;                    // second raw buffer: consecutive cone_fc16 launches ping-pong
    int cone_fc_rows = 64;                        // cone levels with at most this many output rows run as cone_fc16 (0: never)
    int* d_off0 = nullptr;                        // Hset[0] on device
    std::vector<float*> cone[2];                  // cone[t&1][h]: [|Hset[h]|][16][256], ping-pong over steps
    // cone head in one launch (cone_head): V . Wc per batch, Q . Wq + bias per position
    bool cone_head_ok = false;
    float *Wt_c = nullptr, *VW = nullptr, *QWhist = nullptr; int kc_c = 0, ldvw = 0;
    // dec_loop mode: the cone waits / signals inside its own first / last launch
    bool cone_inline_sig = false; uint32_t cone_wait_val = 0, cone_done_val = 0, cone_done_total[LOOP_MAX_LEVELS] = {0};     // per cone level: arrivals so far
    unsigned* d_cone_count = nullptr;
    long long* d_cldbg = nullptr;       // OPH_RUN_STAMPS: hc_fused's phase stamps of workgroup 0, [level][8]
    float *coneR = nullptr, *coneRaw = nullptr, *coneTmp = nullptr;
    // the cone's levels as one launch each (hc_fused): every level also as fp16 hi / lo planes, the LayerNorm exchange granules
    bool cone_fused_ok = false;         // weights packed for it (standard geometry)
    std::vector<void*> coneH[2], coneL[2];
    unsigned long long* d_hcf_stats = nullptr; uint32_t hcf_epoch = 0; int hcf_capacity = -1;
    size_t hcf_stats_stride = 0;        // granules of one level's statistics region
    // hc_fused_pair (the cone's last two levels in one launch): the two levels' step-independent arguments per step parity, the word
    // their workgroups count into between the levels, how often they have (8 per launch)
    int ldy = 0;
    // timing
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int profiling = 0;                  // 0 off; 1 every kernel class; 2 only the whole-decode launch (one event pair per decode: cheap enough for a timed region)
    bool prof_on(int cls) const { return profiling == 1 || (profiling == 2 && cls == PC_DECLOOP); }
    ProfClass prof[PC_COUNT];

    void fail(const char* fmt, ...) {
        char buf[1024];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
    }
    // Device memory comes out of a few large slabs (bump allocation, zero-filled, 256-byte aligned) instead of one hipMalloc per
    // buffer: hundreds of small allocations are backed by small page fragments, and the streaming kernels' rows then miss the TLB
    // all the time; a slab is one large-fragment mapping.  Pool 0: the packed weights (live as long as the handle); pool 1: the
    // per-batch-size state (released and rebuilt when the number of 16-row tiles changes).
    struct Slab { char* base; size_t size, used; };
    std::vector<Slab> slabs[2];
    int pool = 0;
    template <class T>
    T* dalloc(size_t n) {
        const size_t bytes = (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~(size_t)255;
        std::vector<Slab>& v = slabs[pool];
        if (v.empty() || v.back().used + bytes > v.back().size) {
            const size_t want = std::max<size_t>(bytes, (size_t)256 << 20);
            void* p = nullptr;
            if (hipMalloc(&p, want) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            v.push_back(Slab{(char*)p, want, 0});
        }
        Slab& sl = v.back();
        void* p = sl.base + sl.used;
        sl.used += bytes;
        hipMemsetAsync(p, 0, bytes, stream);
        hipStreamSynchronize(stream);     // setup path only; keeps later copies on any stream ordered
        return (T*)p;
    }
    void free_pool(int which) { for (Slab& sl : slabs[which]) hipFree(sl.base); slabs[which].clear(); }
    // ---- profiling brackets
    void pbegin(int cls) {
        if (!prof_on(cls) || g_group_cls == cls) return;
        ProfClass& pc = prof[cls];
        if (pc.used == pc.ev.size()) {
            hipEvent_t a, b;
            hipEventCreate(&a);
            hipEventCreate(&b);
            pc.ev.emplace_back(a, b);
        }
        hipEventRecord(pc.ev[pc.used].first, g_cur);
    }
    // group bracket: ONE event pair around a run of consecutive launches of class `cls` on `cur`
    // (per-launch event records would add ~3 us to 5-9 us kernels and disagree with rocprof)
    void gbegin(int cls) { pbegin(cls); g_group_cls = cls; }
    void gend(int cls) {
        g_group_cls = -1;
        if (!prof_on(cls)) return;
        ProfClass& pc = prof[cls];
        hipEventRecord(pc.ev[pc.used].second, g_cur);
        pc.used++;
    }
    void pend(int cls, double bytes, double flops) {
        ProfClass& pc = prof[cls];
        pc.launches++;
        pc.bytes += bytes;
        pc.flops += flops;
        if (!prof_on(cls) || g_group_cls == cls) return;
        hipEventRecord(pc.ev[pc.used].second, g_cur);
        pc.used++;
    }
};


// ---- oph_model.hip
void build_networks(oph_handle* h);
const std::vector<float>* getw(oph_handle* h, const std::string& name);
float* upload(oph_handle* h, const std::vector<float>& v);
float* upload_padded(oph_handle* h, const std::vector<float>& v, int padto);
std::vector<float> pack_conv(const float* k, int size, int cin, int cout, int kc, int Nalloc);
int pack_layer(oph_handle* h, Layer& l);

// ---- oph_nets.hip
// Runs `layers` over dense rows (B utterances x T frames).  in: [B*T][ld_in] padded rows.
// final_out/final_ld: where the LAST layer's epilogue writes (e.g. Z with ld = full_dim).
// Returns pointer to the final activation rows and their ld via *out_ld; rows via *out_rows.
struct BatchedIO {
    const int* spk = nullptr;        // speaker id per utterance of THIS batch of rows (LCC gates, appended embeddings); null: the staged batch's
    // row mapping of the LAST layer's output (streamed SSRN chunks): its M rows are [B][out_T]; row (b, u) with
    // keep_lo <= u < keep_hi is stored at output row b * out_bs + out_t0 + u, the others are not stored.  out_T == 0: dense.
    int out_T = 0, keep_lo = 0, keep_hi = 0; long long out_bs = 0; int out_t0 = 0;
    float* final_logits = nullptr;   // also store the last layer's PRE-activation rows here (same mapping and row stride)
};
void run_gemm(oph_handle* h, const GemmArgs& a, int cin_true, int prec = 0);       // prec: 0 fp32 MFMA, 1 split-bf16 x3, 2 split-fp16 x3
void run_epi(oph_handle* h, const EpiArgs& a);
void run_dec(oph_handle* h, const DecArgs& a, const Layer& l);
float* run_batched(oph_handle* h, const std::vector<Layer>& layers, float* in, int ld_in, int B, int T, int wsi, int prec,
                   float* final_out, int final_ld, int final_pad, int* out_ld, long long* out_rows, const BatchedIO& io = BatchedIO());
int ensure_batched_capacity(oph_handle* h, int B);
int check_convt_ln(oph_handle* h, float* z_host = nullptr);       // after a synchronisation: did a fused conv1d_transpose + LayerNorm launch time out?  (redoes SSRN)
bool convt_ln_fits(const oph_handle* h, int wsi);       // after a synchronisation: did a fused conv1d_transpose + LayerNorm launch time out?
int run_encode_into(oph_handle* h, const int* dL, const int* dSpk, int B, float* KVdst, hipStream_t stream, int wsi);
int run_encode(oph_handle* h);
int run_ssrn_on(oph_handle* h, const float* Yrows, int ldy, int B, int T, float* Zout, int wsi = 0, float* Zlogits = nullptr, const int* dSpk = nullptr);
void ssrn_margins(const oph_handle* h, int* back, int* ahead);
int copy_mag_rows(oph_handle* h, int a, int b, hipStream_t after);
int run_ssrn_chunk(oph_handle* h, int a, int b, hipStream_t st, int wsi);
int ssrn_stream_chunks(oph_handle* h, int frames_ready, bool final, bool side_tail = false);
int finish_ssrn(oph_handle* h);

// ---- oph_cone.hip
void launch_cone(oph_handle* h, int t);
bool hcf_fits(oph_handle* h);

// ---- oph_decode.hip
constexpr int TILE = 16;       // utterances per decode tile = rows of every decode kernel's row block
int idx_of(const std::vector<int>& v, int x);
void select_tile(oph_handle* h, int j);
int ensure_decode_state(oph_handle* h, int B);
void reset_decode(oph_handle* h);
void begin_batch(oph_handle* h);
bool run_supported(const oph_handle* h);
void drain_loop_clock(oph_handle* h);
void recover_loop_state(oph_handle* h);
int decode_range(oph_handle* h, int t_begin, int t_end, int stop_mode, int32_t* steps_run);
int decode_batch(oph_handle* h, int t_end, int stop_mode, int32_t* steps_run);

