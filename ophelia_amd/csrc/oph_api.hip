// libophelia_hip.so -- model assembly, weight packing, decode loop and the C ABI
// (include/ophelia_hip.h).  Mirrors, for the synthesis path only:
//   networks.py  TextEnc 121-212, AudioEnc 214-284, Attention 286-325, AudioDec 360-435, SSRN 437-537
//   architectures.py 69-81, 139-142, 188-239 ; synthesize.py 150-260
// There is no CPU fallback anywhere in this file: every numeric result comes from the
// HIP kernels in oph_kernels.hip.
#include "oph_internal.h"
#include "../../include/ophelia_hip.h"

#include <algorithm>
#include <chrono>
#include <ctime>
#include <climits>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace oph;

static thread_local std::string g_create_error;
static thread_local hipStream_t g_cur = nullptr;    // stream the launch wrappers of THIS host thread target
static thread_local int g_group_cls = -1;
static double g_host_us[4] = {0, 0, 0, 0};     // OPH_TRACE: host time spent enqueuing {event ops, cone, critical launches, other}
struct HostTimer {
    int slot; std::chrono::steady_clock::time_point t0; bool on;
    HostTimer(int s, bool enable) : slot(s), on(enable) { if (on) t0 = std::chrono::steady_clock::now(); }
    ~HostTimer() { if (on) g_host_us[slot] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6; }
};
static const bool g_trace = getenv("OPH_TRACE") != nullptr;
#define TRACE(...) do { if (g_trace) { fprintf(stderr, "[oph] " __VA_ARGS__); fputc('\n', stderr); fflush(stderr); } } while (0)
static thread_local std::string g_op_error;

#define HIPCHK(h, expr)                                                                        \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (h)->fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return OPH_ERR_DEVICE;                                                             \
        }                                                                                      \
    } while (0)

namespace {

enum Kind { K_CONV = 0, K_HC = 1, K_CONVT = 2 };

struct Layer {
    std::string scope;
    int kind = K_CONV;
    int cin = 0, cout = 0;       // cout = filters (hc: C ; raw conv output has 2C columns)
    int size = 1, rate = 1;
    bool causal = false;
    int act = ACT_NONE;
    int ccat = 0;                // speaker-embedding channels concatenated to the input (cin includes them)
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
    float* Wsw_cone = nullptr;                   // AudioDec highway layers: kernel in cone_loop's fragment order (oph_coneloop.hip)
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

// The OPH_* environment switches (debugging / measurement knobs, README.md lists them), read ONCE per handle in
// oph_create -- nothing on a launch path calls getenv.
struct Options {
    int decode = 0;                  // OPH_DECODE = loop (0, default where possible) | runs (1: two launches per step) | layers (2: one launch per layer)
    int run_rows = 8;                // OPH_RUN_ROWS: utterance rows per workgroup of dec_loop (8 or 4)
    // split-K of the cone GEMMs: many-row levels have enough tiles to fill the cone's CUs with less splitting (fewer partials
    // to write and re-read).  OPH_CONE_KSPLIT="big,small", each 1..4; measured (profiles/r02): 3,4 best
    int ksplit_big = 3, ksplit_small = CONE_KSPLIT;
    int fc_rows = -1, fc_insplit = 2;   // OPH_CONE_FC_ROWS (cone levels of at most this many rows run as cone_fc16), OPH_CONE_FC_INSPLIT
    int lookahead = 8;               // OPH_LOOP_LOOKAHEAD: cones the host may queue ahead of the loop kernel's progress
    int loop_dbg = 0;                // OPH_LOOP_DBG: ablation bits of dec_loop (results are wrong when set, except 16)
    int cu_dec = 0, cu_cone = 0;     // OPH_CU_SPLIT="chain,cone" CUs of the three partitions (rest: SSRN)
    bool no_cu_mask = false, ssrn_all = false, cone_all = false;      // OPH_NO_CU_MASK, OPH_SSRN_ALL, OPH_CONE_ALL
    bool no_cone_head = false, no_loop_qw = false, no_preencode = false, no_stream_ssrn = false, no_cone_loop = false;
    int cone_prec = -1;              // OPH_CONE_PREC: the two many-row cone contractions: 0 fp32 MFMA, 1 split-bf16 x3 (experiment), 2 split-fp16 x3; -1 = the default
    int ssrn_prec = -1;              // OPH_SSRN_PREC: 0 fp32 MFMA, 1 split-bf16 x3, 2 split-fp16 x3; -1 = the default (OPH_SSRN_FP32 = 0)
    int textenc_prec = -1;           // OPH_TEXTENC_PREC: 0 fp32 MFMA, 2 split-fp16 x3; -1 = the default
    bool ssrn_fp32 = false;          // OPH_SSRN_FP32
    bool skip_cone = false;          // OPH_SKIP_CONE: timing experiments only, results are wrong
    bool stream_value = false;       // OPH_STREAM_VALUE: per-step launch paths chain their two streams with stream value operations
    bool run_stamps = false;         // OPH_RUN_STAMPS: clock stamps of the decode kernels' phases (printed under OPH_TRACE)
    int ssrn_chunk = 40;             // OPH_SSRN_CHUNK: mel frames per streamed SSRN chunk (0 = SSRN only after the decode)
    int cl_wgs_per_cu = 2, cl_dbg = 0;   // OPH_CL_WGS_PER_CU (cone_loop workgroups per CU: 1 or 2), OPH_CL_DBG
    bool no_fused_cone = false;      // OPH_NO_FUSED_CONE: the cone's levels as contraction + ln_rows launches instead of hc_fused
    bool no_chain = false;           // OPH_NO_CHAIN: the whole-decode launch as dec_loop (generic) even where dec_chain (specialised) fits
    void read() {
        auto flag = [](const char* n) { return getenv(n) != nullptr; };
        auto num = [](const char* n, int dflt) { const char* e = getenv(n); return e ? atoi(e) : dflt; };
        if (const char* m = getenv("OPH_DECODE")) decode = !strcmp(m, "runs") ? 1 : (!strcmp(m, "layers") ? 2 : 0);
        if (flag("OPH_NO_DECRUN")) decode = 2;
        else if (flag("OPH_NO_DECLOOP") && decode == 0) decode = 1;
        run_rows = num("OPH_RUN_ROWS", 8) == 4 ? 4 : 8;
        if (const char* e = getenv("OPH_CONE_KSPLIT")) { int a_ = 0, b_ = 0; if (sscanf(e, "%d,%d", &a_, &b_) == 2 && a_ >= 1 && a_ <= CONE_KSPLIT && b_ >= 1 && b_ <= CONE_KSPLIT) { ksplit_big = a_; ksplit_small = b_; } }
        fc_rows = num("OPH_CONE_FC_ROWS", -1); fc_insplit = std::max(1, num("OPH_CONE_FC_INSPLIT", 2));
        lookahead = num("OPH_LOOP_LOOKAHEAD", 8); loop_dbg = num("OPH_LOOP_DBG", 0);
        if (const char* sp = getenv("OPH_CU_SPLIT")) { int a_ = 0, b_ = 0; if (sscanf(sp, "%d,%d", &a_, &b_) == 2 && a_ > 0 && b_ > 0) { cu_dec = a_; cu_cone = b_; } }
        no_cu_mask = flag("OPH_NO_CU_MASK"); ssrn_all = flag("OPH_SSRN_ALL"); cone_all = flag("OPH_CONE_ALL");
        no_cone_head = flag("OPH_NO_CONE_HEAD"); no_loop_qw = flag("OPH_NO_LOOP_QW"); no_preencode = flag("OPH_NO_PREENCODE");
        no_stream_ssrn = flag("OPH_NO_STREAM_SSRN");
        // the cone as one persistent launch (cone_loop) is opt-in: measured 25.5-26.1 ms per batch against 25.0 ms with the nine
        // launches per step; it frees the host thread from enqueuing, which matters with 8 ranks on one node
        no_cone_loop = !flag("OPH_CONE_LOOP") || flag("OPH_NO_CONE_LOOP");
        ssrn_fp32 = flag("OPH_SSRN_FP32"); skip_cone = flag("OPH_SKIP_CONE");
        cone_prec = num("OPH_CONE_PREC", flag("OPH_CONE_BF16X3") ? 1 : -1); if (cone_prec > 2) cone_prec = -1;
        ssrn_prec = num("OPH_SSRN_PREC", ssrn_fp32 ? 0 : -1); if (ssrn_prec > 2) ssrn_prec = -1;
        textenc_prec = num("OPH_TEXTENC_PREC", -1); if (textenc_prec != 0 && textenc_prec != 2) textenc_prec = -1;
        { const char* sv = getenv("OPH_STREAM_VALUE"); stream_value = sv && atoi(sv) != 0; }
        run_stamps = flag("OPH_RUN_STAMPS");
        ssrn_chunk = std::max(0, num("OPH_SSRN_CHUNK", 40));
        cl_wgs_per_cu = num("OPH_CL_WGS_PER_CU", 2) == 1 ? 1 : 2; cl_dbg = num("OPH_CL_DBG", 0);
        no_chain = flag("OPH_NO_CHAIN"); no_fused_cone = flag("OPH_NO_FUSED_CONE");
    }
    int cone_ksplit(int M) const { return M >= 512 ? ksplit_big : ksplit_small; }
};
enum { PC_GEMM = 0, PC_GEMM64, PC_GEMM_BF16, PC_LN, PC_DEC, PC_ROWCHAIN, PC_ATTN_ROWS, PC_MISC, PC_DECRUN, PC_DECLOOP, PC_CONEHEAD, PC_COUNT };   // PC_GEMM = the <128,128> instance

}  // namespace

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
    long long* d_amax = nullptr;        // oph_text2mel_graph: argmax per (utterance, frame)
    // ---- the staged batch: nB utterances, resident in HBM, utterance-major.  Text is double-buffered so that the NEXT
    // batch can be staged (oph_stage_text_next) and pre-encoded while this one decodes.
    int nB = 0, nBpad = 0;
    int *bL[2] = {nullptr, nullptr}, *bEnds[2] = {nullptr, nullptr}, *bSpk[2] = {nullptr, nullptr}; int txt = 0;     // bX[txt]: current text
    int next_B = 0; bool next_staged = false;      // bX[txt ^ 1] holds a staged next batch of next_B utterances
    bool txt_ran = false;               // the current text has been through a run (a staged next text may take its place)
    bool kv_pre = false;                // bKV[kv_cur] already holds the current text's K,V (pre-encoded while the previous batch decoded)
    long long n_textenc = 0, n_preenc_used = 0, n_chunks_streamed = 0, n_loop_decodes = 0, n_loop_fallbacks = 0, n_tile_resumes = 0;   // oph_get_counters
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
    // the cone as ONE persistent launch beside dec_loop (cone_loop, oph_coneloop.hip)
    bool cone_loop_ok = false;          // this model's geometry fits it (d = 256, no speaker concat / LCC / nonorm in AudioDec)
    int cone_loop_wgs = -1;             // its grid: workgroups that are resident at once on the cone partition (multiple of 8; -1: not asked yet)
    unsigned* d_cl_flags = nullptr; unsigned long long* d_cl_stats = nullptr;     // [flags | level counters], statistics granules
    uint32_t cl_epoch = 0;
    long long* d_cldbg = nullptr;       // OPH_RUN_STAMPS: cone_loop's per-step stamps
    long long n_cone_loops = 0;
    float *coneR = nullptr, *coneRaw = nullptr, *coneTmp = nullptr;
    // the cone's levels as one launch each (hc_fused): every level also as fp16 hi / lo planes, the LayerNorm exchange granules
    bool cone_fused_ok = false;         // weights packed for it (standard geometry)
    std::vector<void*> coneH[2], coneL[2];
    unsigned long long* d_hcf_stats = nullptr; uint32_t hcf_epoch = 0; int hcf_capacity = -1;
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

namespace {

// ------------------------------------------------------------------ network description
void add_conv(std::vector<Layer>& v, const std::string& scope, int cin, int cout, bool causal, int act, int ccat = 0) {
    Layer l;
    l.scope = scope; l.kind = K_CONV; l.cin = cin; l.cout = cout; l.size = 1; l.rate = 1;
    l.causal = causal; l.act = act; l.ccat = ccat;
    v.push_back(l);
}
void add_hc(std::vector<Layer>& v, const std::string& scope, int c, int size, int rate, bool causal) {
    Layer l;
    l.scope = scope; l.kind = K_HC; l.cin = c; l.cout = c; l.size = size; l.rate = rate; l.causal = causal;
    v.push_back(l);
}
std::string sc(const char* net, const char* pfx, int i) {
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
        add_conv(h->ssrn, sc(n, "C", i++), m.full_dim, m.full_dim, false, ACT_SIGMOID);   // squash_output_ssrn
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
                inv(l.scope + "/conv1d/kernel", {1, l.cin, l.cout});
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

int pack_layer(oph_handle* h, Layer& l) {
    l.kc = round_up(l.cin, 32);
    if (l.kind == K_CONVT) {
        // [TF-sem] o[2t] = x[t].Kt[0,0]^T + x[t-1].Kt[0,2]^T ; o[2t+1] = x[t].Kt[0,1]^T   (modules.py:242-250)
        const std::vector<float>& kt = *getw(h, l.scope + "/conv2d_transpose/kernel");   // (1,3,cout,cin)
        l.N = l.cout; l.Nalloc = round_up(l.N, 128); l.ntaps = 2; l.off[0] = 0; l.off[1] = -1;
        std::vector<float> we((size_t)l.Nalloc * 2 * l.kc, 0.f), wo((size_t)l.Nalloc * l.kc, 0.f);
        for (int n = 0; n < l.cout; ++n)
            for (int c = 0; c < l.cin; ++c) {
                we[(size_t)n * 2 * l.kc + c] = kt[((size_t)0 * l.cout + n) * l.cin + c];
                we[(size_t)n * 2 * l.kc + l.kc + c] = kt[((size_t)2 * l.cout + n) * l.cin + c];
                wo[(size_t)n * l.kc + c] = kt[((size_t)1 * l.cout + n) * l.cin + c];
            }
        l.Wt = upload(h, we);
        l.Wt2 = upload(h, wo);
        l.bias = upload_padded(h, *getw(h, l.scope + "/conv2d_transpose/bias"), l.Nalloc);
        l.g1 = upload_padded(h, *getw(h, l.scope + "/normalize/gamma"), 256);
        l.b1 = upload_padded(h, *getw(h, l.scope + "/normalize/beta"), 256);
        return (l.Wt && l.Wt2 && l.bias && l.g1 && l.b1) ? 0 : -1;
    }
    l.N = l.kind == K_HC ? 2 * l.cout : l.cout;
    l.Nalloc = round_up(l.N, 128);
    l.ntaps = l.size;
    for (int t = 0; t < l.size; ++t)   // causal: x[t-(size-1-k)*rate] (modules.py:123-127); SAME: centred [TF-sem]
        l.off[t] = l.causal ? -(l.size - 1 - t) * l.rate : (t - (l.size - 1) / 2) * l.rate;
    const std::vector<float>& k = *getw(h, l.scope + "/conv1d/kernel");
    l.Wt = upload(h, pack_conv(k.data(), l.size, l.cin, l.N, l.kc, l.Nalloc));
    if (l.kind == K_CONV && l.size == 1 && l.N <= 256) {
        l.ldn = round_up(l.N, 4);
        std::vector<float> wk((size_t)l.kc * l.ldn, 0.f);
        for (int c = 0; c < l.cin; ++c)
            for (int n = 0; n < l.N; ++n) wk[(size_t)c * l.ldn + n] = k[(size_t)c * l.N + n];
        l.Wkn = upload(h, wk);
        if (!l.Wkn) return -1;
    }
    l.bias = upload_padded(h, *getw(h, l.scope + "/conv1d/bias"), l.Nalloc);
    if (!l.cat_scope.empty()) {
        l.cat_table = upload(h, *getw(h, l.cat_scope + "/lookup_table"));
        if (!l.cat_table) return -1;
    }
    if (l.lcc) {
        const std::vector<float>& tb = *getw(h, l.scope + "/lcc_embed/lookup_table");    // (nspeakers, cout)
        std::vector<float> gate(tb.size());
        for (size_t i = 0; i < tb.size(); ++i) gate[i] = 1.0f / (1.0f + expf(-(i < (size_t)l.cout ? 0.0f : tb[i])));
        l.lcc_gate = upload(h, gate);
        if (!l.lcc_gate) return -1;
    }
    if (!l.ln) {
        l.g1 = l.g2 = h->d_ones;
        l.b1 = l.b2 = h->d_zeros;
    } else if (l.kind == K_HC) {
        l.g1 = upload_padded(h, *getw(h, l.scope + "/H1/gamma"), 256);
        l.b1 = upload_padded(h, *getw(h, l.scope + "/H1/beta"), 256);
        l.g2 = upload_padded(h, *getw(h, l.scope + "/H2/gamma"), 256);
        l.b2 = upload_padded(h, *getw(h, l.scope + "/H2/beta"), 256);
        if (!l.g2 || !l.b2) return -1;
    } else {
        l.g1 = upload_padded(h, *getw(h, l.scope + "/normalize/gamma"), 256);
        l.b1 = upload_padded(h, *getw(h, l.scope + "/normalize/beta"), 256);
    }
    return (l.Wt && l.bias && l.g1 && l.b1) ? 0 : -1;
}

// ------------------------------------------------------------------ launch wrappers with accounting
void run_gemm(oph_handle* h, const GemmArgs& a, int cin_true, int prec = 0) {       // prec: 0 fp32 MFMA, 1 split-bf16 x3, 2 split-fp16 x3 (a.Wh / a.Wl / a.f16 set to match)
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
float* run_batched(oph_handle* h, const std::vector<Layer>& layers, float* in, int ld_in, int B, int T, int wsi, int prec,
                   float* final_out, int final_ld, int final_pad, int* out_ld, long long* out_rows, const BatchedIO& io = BatchedIO()) {
    const int* spk_ids = io.spk ? io.spk : h->bSpk[h->txt];
    float* x = in;
    int ldx = ld_in;
    int Tcur = T;
    float* const wsA = wsi ? h->actA2 : h->actA; float* const wsB = wsi ? h->actB2 : h->actB; float* const wsraw = wsi ? h->raw2 : h->raw;
    float* bufs[2] = {wsA, wsB};
    int flip = (in == wsA) ? 1 : 0;
    for (size_t li = 0; li < layers.size(); ++li) {
        const Layer& l = layers[li];
        const bool last = li + 1 == layers.size();
        const int M = B * Tcur;
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
            {   // both phases in one launch
                const int p2 = (prec && g.Wh && g2.Wh) ? std::min(prec, 2) : 0;
                const int cls = p2 ? PC_GEMM_BF16 : (conv_gemm_tile_m(g.M, g.N) == 128 ? PC_GEMM : PC_GEMM64);
                h->pbegin(cls);
                launch_conv_gemm_pair(g, g2, p2, g_cur);
                h->pend(cls, ((double)g.M * l.cin + 2.0 * g.M * g.N + 3.0 * g.N * l.cin) * 4.0, 2.0 * g.M * g.N * 3.0 * l.cin);
            }
            Tcur *= 2;
            e.ldh = l.Nalloc; e.M = B * Tcur; e.C = l.cout; e.mode = PRE_CONV; e.act = ACT_NONE;
            run_epi(h, e);
        } else {
            const bool f16 = prec >= 2;
            g.nprod = prec == 3 ? 2 : (prec == 4 ? 1 : 3);
            g.N = l.N; g.ldh = l.Nalloc; g.M = M; g.Wt = l.Wt; g.Wh = f16 ? l.Wh16 : l.Wh; g.Wl = f16 ? l.Wl16 : l.Wl; g.f16 = f16; g.ldw = l.ntaps * l.kc; g.ntaps = l.ntaps;
            for (int t = 0; t < 3; ++t) g.off[t] = l.off[t];
            run_gemm(h, g, l.cin, g.Wh ? std::min(prec, 2) : 0);
            e.ldh = l.Nalloc; e.M = M; e.C = l.cout;
            if (l.kind == K_HC) { e.mode = PRE_HC; e.Xres = x; e.ldres = ldx; }
            else e.mode = PRE_CONV;
            run_epi(h, e);
            if (last && io.final_logits && l.kind == K_CONV) {      // the fetch surface's g.Z_logits / g.Y_logits: the same rows before the squash
                EpiArgs el = e;
                el.act = ACT_NONE; el.Y = io.final_logits;
                run_epi(h, el);
            }
        }
        x = y; ldx = e.ldy;
        flip ^= 1;
    }
    if (out_ld) *out_ld = ldx;
    if (out_rows) *out_rows = (long long)B * Tcur;
    return x;
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
    if (!h->actA || !h->actB || !h->raw || !h->actA2 || !h->actB2 || !h->raw2) { h->fail("out of device memory for batch %d", B); return OPH_ERR_DEVICE; }
    h->capB = B;
    return 0;
}

// ------------------------------------------------------------------ decoder state
int idx_of(const std::vector<int>& v, int x) {
    auto it = std::lower_bound(v.begin(), v.end(), x);
    return (it != v.end() && *it == x) ? (int)(it - v.begin()) : -1;
}

constexpr int TILE = 16;       // utterances per decode tile = rows of every decode kernel's row block

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
        const size_t mt = (maxrows * Bpad + 63) / 64;
        h->d_hcf_stats = h->dalloc<unsigned long long>((size_t)nh * mt * 2 * 2 * 32 * 8 * 2);      // one region per level: launches of two streams overlap
        h->hcf_epoch = 0;
        if (!h->d_hcf_stats) { h->fail("out of device memory for decode state"); return OPH_ERR_DEVICE; }
    }
    if (h->cone_loop_ok) {
        bool fits = nh <= CL_MAX_LEVELS;
        for (int k = 0; k < nh; ++k) fits = fits && (int)h->Hset[k].size() <= CL_MAX_POS;
        h->d_cl_flags = fits ? h->dalloc<unsigned>((size_t)2 * CL_MAX_LEVELS * CL_MAX_POS + 2 * CL_MAX_LEVELS + 8 * 16) : nullptr;      // flags | level counters | task queues
        h->d_cl_stats = fits ? h->dalloc<unsigned long long>((size_t)2 * CL_MAX_LEVELS * CL_MAX_POS * 8 * 64) : nullptr;
        h->cl_epoch = 0;
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
    hipMemsetAsync(h->d_p, 0, 2 * h->Bpad * 4, h->stream);
    hipMemsetAsync(h->Yout, 0, (size_t)h->Bpad * m.max_T * h->ldy * 4, h->stream);
    hipMemsetAsync(h->Ytm, 0, (size_t)(m.max_T + 1) * h->Bpad * h->ldy * 4, h->stream);
    hipMemsetAsync(h->align, 0, (size_t)h->Bpad * m.max_N * m.max_T * 4, h->stream);
    launch_fill_int(h->d_tends, m.max_T, h->Bpad, h->stream);
    const int ctl[4] = {0, INT_MAX, 0, 0};
    hipMemcpyAsync(h->d_ctl, ctl, sizeof ctl, hipMemcpyHostToDevice, h->stream);
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
    hipStreamSynchronize(h->stream);
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

// AudioDec history cone for step t under the mask p_t (= max_attentions of step t-1).
// The reference re-evaluates R[t'] for ALL t' <= t with the current prev_max (networks.py:311
// tiles one mask over max_T), so AudioDec's hidden history cannot be cached across steps; what is
// recomputed here is the sparse receptive cone of row t: the highway-layer inputs at the history
// offsets Hset[k] (84, 82, 44, 14, 4, 2 positions for rates 1,3,9,27,1,1).  It depends only on
// p_t and Q[<t], both known right after attn_step(t-1): it runs on the SIDE stream, concurrently
// with the AudioEnc chain of step t, into the ping-pong buffer cone[t&1].
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
        ch.stop_after = stop_after; ch.t = t;
        const bool fused = h->cone_fused_ok && !spk_next && h->cone_prec == 2 && h->hcf_capacity != 0;
        if (fused) { ch.Yh = h->coneH[t & 1][0]; ch.Yl = h->coneL[t & 1][0]; }
        ch.wait_sig = ar.wait_sig; ch.wait_val = ar.wait_val; ch.wait_err = ar.wait_err;
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
            bool fits = h->hcf_capacity != 0;
            if (h->hcf_capacity < 0) {
                int ncu = 0;
                for (int i = 0; i < h->mask_words; ++i) ncu += __builtin_popcount(h->m_conep[i]);
                if (h->mask_words == 0) { hipDeviceProp_t prop; ncu = hipGetDeviceProperties(&prop, h->device) == hipSuccess ? prop.multiProcessorCount : 0; }
                fits = true;
                for (int k = 0; k + 1 < nh; ++k) {
                    const int Mk = (int)h->Hset[k + 1].size() * Bpad;
                    fits = fits && hc_fused_grid(Mk) <= hc_fused_blocks_per_cu(Mk) * ncu;
                }
                h->hcf_capacity = fits ? 1 : 0;
            }
            if (!fits) h->hcf_capacity = 0;        // (this launch already wrote the planes; harmless) -> the unfused path from here on
            else {
                if (h->hcf_epoch > 0xF0000000u) {
                    hipStreamSynchronize(h->scone);
                    hipMemsetAsync(h->d_hcf_stats, 0, (size_t)nh * ((h->Hset[0].size() * Bpad + 63) / 64) * 2 * 2 * 32 * 8 * 2 * sizeof(unsigned long long), g_cur);
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
                    f.stats = h->d_hcf_stats + (size_t)k * ((h->Hset[0].size() * Bpad + 63) / 64) * 2 * 2 * 32 * 8 * 2; f.epoch = ++h->hcf_epoch; f.err = h->d_ctl + 2; f.zeros = h->d_zeros;
                    f.stop_after = stop_after; f.t = t;
                    if (h->cone_inline_sig && k + 1 < LOOP_MAX_LEVELS) {
                        const Layer& tl = h->audiodec[pre + k + 1];
                        f.coh0 = idx_of(h->Hset[k + 1], -tl.off[0]); f.coh1 = idx_of(h->Hset[k + 1], -tl.off[1]);
                        h->cone_done_total[k + 1] += (unsigned)hc_fused_holders(f.M, Bpad, f.coh0, f.coh1);
                        f.done_sig = h->d_sig + LOOP_SIG_LEVEL0 + 16 * (k + 1); f.done_val = h->cone_done_val; f.done_count = h->d_cone_count + (k + 1); f.done_target = h->cone_done_total[k + 1];
                        f.done_stamp = stamp_of(k + 1);
                    }
                    if (h->d_cldbg && t == m.max_T / 2) f.dbg = h->d_cldbg + 8 * k;
                    h->pbegin(PC_GEMM_BF16);
                    launch_hc_fused(f, g_cur);
                    h->pend(PC_GEMM_BF16, ((double)f.M * 3.0 * l.cin + (double)f.M * l.cout + (double)l.N * 3.0 * l.cin) * 4.0, 2.0 * f.M * l.N * 3.0 * l.cin);
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
    if (m.flags & (OPH_FLAG_LCC | OPH_FLAG_NO_MONOTONIC)) return false;       // variants served by the per-layer kernels
    if ((int)h->audioenc.size() + h->dec_pre > RUN_MAX_LAYERS || (int)h->audiodec.size() - h->dec_pre + 1 > RUN_MAX_LAYERS) return false;
    for (const auto* net : {&h->audioenc, &h->audiodec})
        for (const Layer& l : *net) {
            if (l.ntaps * l.kc > 768 || l.N > RUN_GCOLS || (l.ntaps == 3 && l.kc > 256) || l.cout > 256) return false;
        }
    return true;
}

int run_encode_into(oph_handle* h, const int* dL, const int* dSpk, int B, float* KVdst, hipStream_t stream, int wsi);      // defined with the batched networks below
int ssrn_stream_chunks(oph_handle* h, int frames_ready, bool final, bool side_tail = false);                                  // streamed SSRN of the current tile
void ssrn_margins(const oph_handle* h, int* back, int* ahead);
int copy_mag_rows(oph_handle* h, int a, int b, hipStream_t after);

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
            const int slices = round_up(q.N, 16) / 16, nch = (q.ntaps * q.kc) / 16;
            std::vector<float> Wh((size_t)slices * 16 * q.ldw, 0.f);
            const size_t rows_have = (size_t)std::min(slices * 16, round_up(q.N, 16));
            if (hipMemcpy(Wh.data(), q.Wt, rows_have * q.ldw * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { h->fail("weight read-back failed"); return OPH_ERR_DEVICE; }
            std::vector<float> Ws((size_t)slices * R * PF * 64 * 4, 0.f);
            for (int g = 0; g < slices; ++g)
                for (int w = 0; w < R; ++w)
                    for (int pf = 0; pf < PF; ++pf) {
                        const int ch = std::min(w + R * pf, nch - 1);
                        for (int lane = 0; lane < 64; ++lane) {
                            const int col = 16 * g + 4 * (lane >> 4) + (lane & 3), k = 16 * ch + 4 * ((lane >> 2) & 3);
                            float* dst = &Ws[((((size_t)g * R + w) * PF + pf) * 64 + lane) * 4];
                            for (int e = 0; e < 4; ++e) dst[e] = Wh[(size_t)col * q.ldw + k + e];
                        }
                    }
            float* dsw = h->dalloc<float>(Ws.size());
            if (!dsw) { h->fail("out of device memory"); return OPH_ERR_DEVICE; }
            if (hipMemcpy(dsw, Ws.data(), Ws.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { h->fail("weight upload failed"); return OPH_ERR_DEVICE; }
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
    if (h->chain_ok && !h->fixed_att && a.QW != nullptr && (dbg & ~32) == 0) launch_dec_chain(a, h->loop_slices, h->sdec);
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
    // ---- the cone of every step as ONE persistent launch (cone_loop) where the model fits it and its workgroups can all be resident
    bool cone_in_loop = false;
    if (h->cone_loop_ok && h->qw_from_loop && h->d_cl_flags && h->d_cl_stats && !h->opt.skip_cone && !(dbg & 32) && t_end > 1 && t_begin == 0) {
        if (h->cone_loop_wgs < 0) {
            int ncu = 0;
            for (int i = 0; i < h->mask_words; ++i) ncu += __builtin_popcount(h->m_conep[i]);
            h->cone_loop_wgs = std::min(h->opt.cl_wgs_per_cu, cone_loop_blocks_per_cu()) * ncu / 8 * 8;
        }
        cone_in_loop = h->cone_loop_wgs >= 64;
    }
    if (cone_in_loop) {
        const int pre = h->dec_pre, nh = h->n_hc_dec;
        if ((uint64_t)h->cl_epoch + (uint64_t)(m.max_T + 2) * CL_MAX_LEVELS > 0xF0000000ull) {
            hipStreamSynchronize(h->scone);
            hipMemsetAsync(h->d_cl_stats, 0, (size_t)2 * CL_MAX_LEVELS * CL_MAX_POS * 8 * 64 * 8, h->scone);
            h->cl_epoch = 0;
        }
        ConeLoopArgs c{};
        c.nlevels = nh; c.t_begin = std::max(1, t_begin); c.t_end = t_end; c.B = h->B; c.d = m.d;
        c.npos0 = (int)h->Hset[0].size(); c.off0 = h->d_off0; c.rows0[0] = h->cone[0][0]; c.rows0[1] = h->cone[1][0];
        const Layer& tl0 = h->audiodec[pre];
        c.sig0_pos0 = idx_of(h->Hset[0], -tl0.off[0]); c.sig0_pos1 = idx_of(h->Hset[0], -tl0.off[1]);
        c.Q = h->Qhist; c.QW = h->QWhist; c.KV = h->KV; c.VW = h->VW; c.ldvw = h->ldvw; c.N_keys = m.max_N; c.win = m.attention_win_size;
        c.gamma0 = h->audiodec[0].g1; c.beta0 = h->audiodec[0].b1;
        c.p = h->d_p;
        for (int k = 1; k < nh; ++k) {
            const Layer& l = h->audiodec[pre + k - 1];       // the highway layer that produces level k from level k-1
            const Layer& tl = h->audiodec[pre + k];           // the chain layer whose taps read level k
            ConeLoopLevel& L = c.L[k];
            L.npos = (int)h->Hset[k].size(); L.Wsw = l.Wsw_cone; L.bias = l.bias; L.g1 = l.g1; L.b1 = l.b1; L.g2 = l.g2; L.b2 = l.b2;
            L.tab = h->d_tab[k - 1]; L.need = h->d_need[k - 1];
            L.rows[0] = h->cone[0][k]; L.rows[1] = h->cone[1][k];
            L.sig_pos0 = idx_of(h->Hset[k], -tl.off[0]); L.sig_pos1 = idx_of(h->Hset[k], -tl.off[1]);
        }
        c.flags = h->d_cl_flags; c.levelcnt = h->d_cl_flags + (size_t)2 * CL_MAX_LEVELS * CL_MAX_POS; c.stats = h->d_cl_stats;
        c.epoch0 = h->cl_epoch; h->cl_epoch += (uint32_t)(m.max_T + 2) * CL_MAX_LEVELS;
        c.sig = h->d_sig; c.sig_base = h->sig_base; c.ctl = h->d_ctl;
        c.dbg = h->opt.cl_dbg;
        if (h->d_cldbg) { c.stamps = h->d_cldbg; hipMemsetAsync(h->d_cldbg, 0, ((size_t)(2 * m.max_T + 4) * 8 + 512) * sizeof(long long), h->scone); }
        hipMemsetAsync(h->d_cl_flags, 0, ((size_t)2 * CL_MAX_LEVELS * CL_MAX_POS + 2 * CL_MAX_LEVELS + 8 * 16) * sizeof(unsigned), h->scone);
        launch_cone_loop(c, h->cone_loop_wgs, h->scone);
        h->n_cone_loops++;
        // the host has nothing to enqueue per step: it only watches the progress word for the SSRN chunks
        auto t_prog = std::chrono::steady_clock::now();
        int last_prog = -2;
        while (stream_ssrn) {
            const int prog = h->host_prog[0], stopped_at = h->host_prog[1];
            if (stopped_at != INT_MAX || prog >= t_end - 1) break;
            Tile& tl = h->tiles[h->tile];
            if (tl.ssrn_done + h->opt.ssrn_chunk >= m.max_T) break;          // only the final chunk is left
            { const int rc = ssrn_stream_chunks(h, prog, false); if (rc) return rc; }
            if (prog != last_prog) { last_prog = prog; t_prog = std::chrono::steady_clock::now(); }
            else if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_prog).count() > 10.0) {
                h->fail("decode loop kernel made no progress for 10 s (step %d)", prog);
                return OPH_ERR_DEVICE;
            }
            struct timespec ts = {0, 50000};         // 50 us: a chunk boundary comes every few milliseconds
            nanosleep(&ts, nullptr);
        }
    }
    for (int t = std::max(1, t_begin); t < t_end && !(dbg & 32) && !cone_in_loop; ++t) {
        // bounded run-ahead, so that an early stop leaves at most `lookahead` queued cones (they early-out on the device)
        auto t_wait0 = std::chrono::steady_clock::now();
        while (h->host_prog[0] < t - 1 - lookahead && h->host_prog[1] == INT_MAX) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_wait0).count() > 10.0) {
                h->fail("decode loop kernel made no progress for 10 s (step %d)", h->host_prog[0]);
                return OPH_ERR_DEVICE;
            }
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
        a.L[a.nlayers - 1].act = ACT_SIGMOID;           // squash_output_t2m (networks.py:430-431)
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
            }
            if (loop_mode && h->d_cldbg && !h->cone_fused_ok) {
                std::vector<long long> cd((size_t)(2 * m.max_T + 4) * 8 + 512);
                hipMemcpy(cd.data(), h->d_cldbg, cd.size() * 8, hipMemcpyDeviceToHost);
                {   // on which XCD did the workgroups of each column group (block % 8) run?
                    char line[256]; int n = 0;
                    for (int c8 = 0; c8 < 8; ++c8) {
                        unsigned mask = 0;
                        for (int b = c8; b < h->cone_loop_wgs && b < 512; b += 8) mask |= 1u << (unsigned)cd[(size_t)(2 * last + 4) * 8 + b];
                        n += snprintf(line + n, sizeof line - n, " %d:0x%x", c8, mask);
                    }
                    TRACE("cone_loop: XCD mask per column group (block %% 8):%s", line);
                }
                for (int t : {50, 100, 150}) {
                    if (t >= last) continue;
                    const long long* q = &cd[(size_t)(last + 1 + t) * 8];
                    if (q[0]) TRACE("cone_loop step %d, sample task (level 4): wait deps %.2f  gather %.2f  mfma %.2f  local stats %.2f  exchange %.2f  normalise %.2f  store+flag %.2f us", t,
                                    (q[1] - q[0]) * 0.01, (q[2] - q[1]) * 0.01, (q[3] - q[2]) * 0.01, (q[4] - q[3]) * 0.01, (q[5] - q[4]) * 0.01, (q[6] - q[5]) * 0.01, (q[7] - q[6]) * 0.01);
                }
                for (int t : {50, 51, 100, 101, 150}) {
                    if (t >= m.max_T) continue;
                    const long long* q = &cd[(size_t)t * 8];
                    if (q[0]) TRACE("cone_loop step %d: levels 0..5 written %.2f %.2f %.2f %.2f %.2f %.2f us after its release", t,
                                    (q[1] - q[0]) * 0.01, (q[2] - q[0]) * 0.01, (q[3] - q[0]) * 0.01, (q[4] - q[0]) * 0.01, (q[5] - q[0]) * 0.01, (q[6] - q[0]) * 0.01);
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
            h->fail(ctl[2] == 3 ? "decoder loop: the side stream never saw the attention signal (time-out)" : ctl[2] == 2 ? "decoder loop: the side stream's cone never signalled (time-out)" : "decoder run: a workgroup hand-off timed out (workgroups of one run were not co-resident)");
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
    bool retried = false;
    for (int j = 0; j < ntiles; ++j) {
        select_tile(h, j);
        reset_decode(h);
        int32_t st = 0;
        int rc = decode_range(h, 0, t_end, stop_mode, &st);
        if (rc == OPH_ERR_DEVICE && h->use_loop && !retried) {
            // the whole-decode launch could not run here (e.g. another process holds CUs of its partition): fall back to the
            // two-launches-per-step path for the rest of this handle's life and redo the tile
            TRACE("whole-decode launch failed (%s): falling back to two launches per step", h->err.c_str());
            h->use_loop = false; retried = true; h->n_loop_fallbacks++;
            reset_decode(h);
            rc = decode_range(h, 0, t_end, stop_mode, &st);
        }
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
            const int rc = decode_range(h, h->tiles[j].steps, batch_steps, OPH_STOP_NEVER, nullptr);
            if (rc) return rc;
            h->n_tile_resumes++;
            h->tiles[j].steps = batch_steps;
        }
    select_tile(h, 0);
    if (steps_run) *steps_run = batch_steps;
    return OPH_OK;
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

int run_ssrn_on(oph_handle* h, const float* Yrows, int ldy, int B, int T, float* Zout, int wsi = 0, float* Zlogits = nullptr) {
    const oph_dims& m = h->dm;
    BatchedIO io{};
    io.final_logits = Zlogits;
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
            if (b + ahead > frames_ready || b >= m.max_T) break;     // (the last frames always belong to the final chunk)
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
                if (h->chunk_ms > remaining) break;
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

}  // namespace

// ====================================================================================== C ABI
extern "C" {

int oph_abi_version(void) { return OPH_ABI_VERSION; }

const char* oph_last_error(const oph_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int oph_create(const oph_dims* dims, int device, oph_handle** out) {
    if (!dims || !out) { g_create_error = "null argument"; return OPH_ERR_INVALID; }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_create_error = "no HIP device available (libophelia_hip has no CPU fallback)";
        return OPH_ERR_DEVICE;
    }
    if (device < 0 || device >= ndev) { g_create_error = "device index out of range"; return OPH_ERR_INVALID; }
    const oph_dims& m = *dims;
    if (m.r != 4 && m.r != 8) { g_create_error = "reduction factor not handled by SSRN (networks.py:474-479)"; return OPH_ERR_UNSUPPORTED; }
    if (m.d % 4 || m.d > 256 || m.c % 4 || 2 * m.c > 1024 || m.n_mels > 256 || m.full_dim > 1280 || m.e % 4 ||
        m.attention_win_size < 1 || m.attention_win_size > 8 || m.max_N < 1 || m.max_T < 1 || m.vocab < 1) {
        g_create_error = "dimensions outside the supported hot path (d<=256, c<=512, n_mels<=256, full_dim<=1280, win<=8)";
        return OPH_ERR_UNSUPPORTED;
    }
    if ((m.flags & (OPH_FLAG_SPK_AUDIO_DECODER_INPUT | OPH_FLAG_SPK_TEXT_ENCODER_INPUT | OPH_FLAG_SPK_TEXT_ENCODER_TOWARDS_END | OPH_FLAG_LCC | OPH_FLAG_SPK_AUDIO_ENCODER_INPUT)) &&
        (m.nspeakers < 1 || m.speaker_embedding_size < 1 || m.speaker_embedding_size % 4)) {
        g_create_error = "multispeaker flag set but nspeakers/speaker_embedding_size invalid";
        return OPH_ERR_INVALID;
    }
    if (hipSetDevice(device) != hipSuccess) { g_create_error = "hipSetDevice failed"; return OPH_ERR_DEVICE; }
    oph_handle* h = new oph_handle();
    h->dm = m;
    h->device = device;
    h->opt.read();
    static const char* names[PC_COUNT] = {"conv_gemm_f32<128,128>", "conv_gemm_f32<64,64>", "conv_gemm_bf16x3", "ln_rows", "dec_layer16", "row_chain", "attn_rows", "misc", "dec_run", "dec_loop", "cone_head"};
    for (int i = 0; i < PC_COUNT; ++i) h->prof[i].name = names[i];
    // CU partition for the decode loop (MI355X: 256 CUs, mask bit i -> XCD i%8): the dependent layers of a step get a
    // private slice of 8 CUs in every XCD so the concurrently running history-cone GEMMs (the next 16 CUs per XCD) and the
    // SSRN partition (the last 8 per XCD) cannot delay them.
    {
        hipDeviceProp_t prop;
        uint32_t m_dec[16] = {0};
        uint32_t* m_cone = h->m_cone; uint32_t* m_conep = h->m_conep; uint32_t* m_ssrn = h->m_ssrn;
        int ncu = 0;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) ncu = prop.multiProcessorCount;
        const int words = (ncu + 31) / 32;
        int ndec = ncu / 4, nconep = ncu / 2;               // 64 | 128 | 64 of 256 CUs (sweep in DESIGN.md)
        if (h->opt.cu_dec > 0 && h->opt.cu_cone > 0 && h->opt.cu_dec + h->opt.cu_cone < ncu) { ndec = h->opt.cu_dec; nconep = h->opt.cu_cone; }
        if (ncu >= 64 && words <= 16 && !h->opt.no_cu_mask) {
            for (int i = 0; i < ncu; ++i) {
                const uint32_t bit = 1u << (i % 32);
                if (i < ndec) m_dec[i / 32] |= bit;
                else {
                    m_cone[i / 32] |= bit;
                    (i < ndec + nconep ? m_conep : m_ssrn)[i / 32] |= bit;
                    if (h->opt.cone_all) m_conep[i / 32] |= bit;
                    if (h->opt.ssrn_all) m_ssrn[i / 32] |= bit;
                }
            }
            // CU-masked queues are a scarce resource: with four alive the queues get time-sliced and even sequential batches
            // run 2x slower (measured), and re-creating masked streams after destroying one hung hipStreamSynchronize: exactly
            // three are created here, once: critical chain | cone | SSRN partitions.  All three or none.
            h->mask_words = words;
            if (hipExtStreamCreateWithCUMask(&h->sdec, words, m_dec) != hipSuccess) { h->sdec = nullptr; h->mask_words = 0; }
            if (h->mask_words && hipExtStreamCreateWithCUMask(&h->scone, words, m_conep) != hipSuccess) { h->scone = nullptr; h->mask_words = 0; }
            if (h->mask_words && hipExtStreamCreateWithCUMask(&h->sssrn, words, m_ssrn) != hipSuccess) { h->sssrn = nullptr; h->mask_words = 0; }
            h->ndec_cus = h->mask_words ? ndec : ncu;
        } else h->ndec_cus = ncu;
        (void)hipGetLastError();
    }
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&h->scopy, hipStreamNonBlocking) != hipSuccess ||
        (!h->sdec && hipStreamCreateWithFlags(&h->sdec, hipStreamNonBlocking) != hipSuccess) ||
        (!h->scone && hipStreamCreateWithFlags(&h->scone, hipStreamNonBlocking) != hipSuccess) ||
        (!h->sssrn && hipStreamCreateWithFlags(&h->sssrn, hipStreamNonBlocking) != hipSuccess) ||
        hipEventCreateWithFlags(&h->ev_dec_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_ssrn_done[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_ssrn_done[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_preenc, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_copy, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_chunk, hipEventDisableTiming) != hipSuccess ||
        hipEventCreate(&h->ev_cs) != hipSuccess || hipEventCreate(&h->ev_ce) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_out, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_attn, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_cone, hipEventDisableTiming) != hipSuccess ||
        hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) {
        g_create_error = "stream/event creation failed";
        delete h;
        return OPH_ERR_DEVICE;
    }
    g_cur = h->stream;
    {
        int can = 0;
        // Device words for the cross-stream dependencies.  The whole-decode launch (dec_loop) polls / raises them in-kernel.
        // For the per-step launch paths, stream write/wait-value operations on them are opt-in (OPH_STREAM_VALUE=1): under
        // rocprofv3 --pmc, which serialises dispatches across queues, a wait-value packet never sees the value the other queue
        // would write and the run deadlocks; events are understood by the profiler.
        h->d_sig = h->dalloc<uint32_t>(LOOP_SIG_WORDS);
        if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, device) == hipSuccess && can) {
            h->can_sigval = h->d_sig != nullptr && hipStreamWriteValue32(h->stream, h->d_sig, 0, 0) == hipSuccess &&
                            hipStreamSynchronize(h->stream) == hipSuccess;
            h->use_sigval = h->can_sigval && h->opt.stream_value;
        }
        (void)hipGetLastError();
        void* hp_ = nullptr;
        if (hipHostMalloc(&hp_, 64, hipHostMallocMapped) == hipSuccess) { h->host_prog = (volatile int*)hp_; h->host_prog[0] = -1; h->host_prog[1] = INT_MAX; }
        (void)hipGetLastError();
    }
    h->ssrn_prec = h->opt.ssrn_prec >= 0 ? h->opt.ssrn_prec : 2;
    build_networks(h);
    *out = h;
    return OPH_OK;
}

int oph_destroy(oph_handle* h) {
    if (!h) return OPH_OK;
    hipSetDevice(h->device);
    TRACE("destroy: sync streams");
    for (hipStream_t st : {h->stream, h->sdec, h->scone, h->sssrn, h->scopy}) if (st) hipStreamSynchronize(st);
    for (auto& pc : h->prof)
        for (auto& e : pc.ev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    for (hipEvent_t e : {h->ev0, h->ev1, h->ev_attn, h->ev_cone, h->ev_in, h->ev_out, h->ev_dec_done, h->ev_ssrn_done[0], h->ev_ssrn_done[1], h->ev_preenc, h->ev_copy, h->ev_chunk, h->ev_cs, h->ev_ce})
        if (e) hipEventDestroy(e);
    TRACE("destroy: free");
    h->free_pool(0); h->free_pool(1);
    if (h->host_prog) hipHostFree((void*)h->host_prog);
    TRACE("destroy: streams");
    for (hipStream_t st : {h->scone, h->sssrn, h->sdec, h->scopy, h->stream}) if (st) { TRACE("  destroy stream %p", (void*)st); hipStreamDestroy(st); }
    TRACE("destroy: done");
    delete h;
    return OPH_OK;
}

// Pinned host memory for result buffers: device-to-host copies into it are true DMA (they overlap the running decode and
// reach the link's rate); any other host pointer is accepted everywhere too, at the cost of staged, blocking copies.
int oph_host_alloc(size_t bytes, void** out) {
    if (!out) return OPH_ERR_INVALID;
    *out = nullptr;
    if (hipHostMalloc(out, std::max<size_t>(bytes, 1), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return OPH_ERR_DEVICE; }
    return OPH_OK;
}
int oph_host_free(void* p) {
    if (!p) return OPH_OK;
    return hipHostFree(p) == hipSuccess ? OPH_OK : OPH_ERR_DEVICE;
}

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

int oph_finalize_weights(oph_handle* h) {
    if (!h) return OPH_ERR_INVALID;
    if (h->finalized) return OPH_OK;
    HIPCHK(h, hipSetDevice(h->device));
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
        auto too_big = [&](const std::vector<Layer>& net, size_t from, size_t to) {
            for (size_t i = from; i < to && i < net.size(); ++i) {
                const std::vector<float>* k = getw(h, net[i].scope + (net[i].kind == K_CONVT ? "/conv2d_transpose/kernel" : "/conv1d/kernel"));
                if (!k) continue;
                for (float v : *k) if (!(std::fabs(v) <= 6.0e4f)) return true;
            }
            return false;
        };
        h->guard_ssrn = too_big(h->ssrn, 0, h->ssrn.size());
        h->guard_cone = too_big(h->audiodec, (size_t)h->dec_pre, (size_t)(h->dec_pre + h->n_hc_dec));
        h->guard_text = too_big(h->textenc, 0, h->textenc.size());
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
        if (!split(l.Wt, (size_t)l.Nalloc * l.ntaps * l.kc, true, l.Wh16, l.Wl16)) { h->fail("out of device memory"); return OPH_ERR_DEVICE; }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->cone_head_ok = !h->opt.no_cone_head && !(h->dm.flags & OPH_FLAG_NO_MONOTONIC) && h->audiodec[0].Wkn != nullptr && h->dm.d <= 256 && (h->dm.d % 4) == 0;
    if (h->cone_head_ok) {
        // Wc = the rows of AudioDec C_1's kernel (1, 2d, d) that multiply the attention context (R' = [ctx | Q], networks.py:316-319)
        const Layer& c1 = h->audiodec[0];
        const std::vector<float>& k = *getw(h, c1.scope + "/conv1d/kernel");
        const int d = h->dm.d;
        h->kc_c = round_up(d, 32); h->ldvw = round_up(d, 128);
        std::vector<float> w((size_t)h->ldvw * h->kc_c, 0.f);
        for (int c = 0; c < d; ++c)
            for (int n = 0; n < d; ++n) w[(size_t)n * h->kc_c + c] = k[(size_t)c * d + n];
        h->Wt_c = upload(h, w);
        if (!h->Wt_c) { h->fail("out of device memory"); return OPH_ERR_DEVICE; }
    }
    // cone_loop: every AudioDec highway layer but the last is re-evaluated over history positions by resident workgroups; its
    // kernel in the lanes' fragment order: [column group cg][wave w][k group i][lane][4] with
    //   column = (w >> 1) * C + 32 cg + 16 (w & 1) + (lane & 15)        (wave 0,1: H1 channels of the group, wave 2,3: the same channels of H2)
    //   k      = 192 (lane >> 4) + 4 i + e                               (k over [tap x[t-2r] | tap x[t-r] | x[t]] x 256 channels)
    {
        const int pre = h->dec_pre, nh = h->n_hc_dec, d = h->dm.d;
        bool ok = !h->opt.no_cone_loop && h->cone_head_ok && pre == 1 && d == 256 && !(h->dm.flags & (OPH_FLAG_LCC | OPH_FLAG_NORM_NONE | OPH_FLAG_NO_MONOTONIC)) &&
                  h->dm.attention_win_size <= 4 && nh >= 2 && nh <= CL_MAX_LEVELS;
        for (int k = 0; ok && k + 1 < nh; ++k) {
            const Layer& l = h->audiodec[pre + k];
            ok = l.kind == K_HC && l.ntaps == 3 && l.kc == 256 && l.cout == 256 && l.cin == 256 && l.ln && !l.lcc && l.ccat == 0 && l.causal;
        }
        for (int k = 0; ok && k + 1 < nh; ++k) {
            Layer& l = h->audiodec[pre + k];
            const std::vector<float>& kr = *getw(h, l.scope + "/conv1d/kernel");      // (3, 256, 512)
            std::vector<float> ws((size_t)8 * 4 * CL_NCH * 64 * 4);
            for (int cg = 0; cg < 8; ++cg)
                for (int w = 0; w < 4; ++w)
                    for (int i = 0; i < CL_NCH; ++i)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 4; ++e) {
                                const int col = (w >> 1) * 256 + 32 * cg + 16 * (w & 1) + (lane & 15);
                                const int kk = 192 * (lane >> 4) + 4 * i + e, tap = kk / 256, c = kk % 256;
                                ws[((((size_t)cg * 4 + w) * CL_NCH + i) * 64 + lane) * 4 + e] = kr[((size_t)tap * 256 + c) * 512 + col];
                            }
            l.Wsw_cone = upload(h, ws);
            if (!l.Wsw_cone) { h->fail("out of device memory"); return OPH_ERR_DEVICE; }
        }
        h->cone_loop_ok = ok;
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
            const std::vector<float>& kr = *getw(h, l.scope + "/conv1d/kernel");      // (3, 256, 512)
            const std::vector<float>& bs = *getw(h, l.scope + "/conv1d/bias");
            std::vector<float> wp((size_t)512 * 768), bp(512);       // [column tile jt][K-step ks][64 columns][64 k]
            for (int np = 0; np < 512; ++np) {
                const int jt = np >> 6, q = np & 63, col = q < 32 ? 32 * jt + q : 256 + 32 * jt + (q - 32);
                bp[np] = bs[col];
                for (int tap = 0; tap < 3; ++tap)
                    for (int c = 0; c < 256; ++c) {
                        const int k = tap * 256 + c;
                        wp[(((size_t)jt * 12 + (k >> 6)) * 64 + q) * 64 + (k & 63)] = kr[((size_t)tap * 256 + c) * 512 + col];
                    }
            }
            float* dwp = upload(h, wp);
            l.bias_p = upload(h, bp);
            if (!dwp || !l.bias_p || !split(dwp, wp.size(), true, l.Wph, l.Wpl)) { h->fail("out of device memory"); return OPH_ERR_DEVICE; }
        }
        h->cone_fused_ok = ok;
    }
    h->emb_text = upload(h, h->hostw["Text2Mel/TextEnc/embed_1/lookup_table"]);
    if (h->dm.flags & OPH_FLAG_SPK_AUDIO_DECODER_INPUT) h->emb_spk = upload(h, h->hostw["Text2Mel/AudioDec/embed_2/lookup_table"]);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    h->hostw.clear();
    h->pool = 1;          // everything allocated from here on is per-batch-size state
    h->use_run = run_supported(h);
    // OPH_DECODE = loop (default where possible) | runs (two launches per step) | layers (one launch per layer, round 1).
    // The whole-decode launch needs its own CU partition (all its workgroups resident while the cone runs beside it) and
    // the mapped progress words.
    h->use_loop = h->use_run && h->opt.decode == 0 && h->d_sig && h->host_prog && h->mask_words > 0;
    h->finalized = true;
    return OPH_OK;
}

static int check_ready(oph_handle* h, int B) {
    if (!h) return OPH_ERR_INVALID;
    if (!h->finalized) { h->fail("weights not finalized"); return OPH_ERR_STATE; }
    if (B < 1) { h->fail("batch must be >= 1"); return OPH_ERR_INVALID; }
    if (hipSetDevice(h->device) != hipSuccess) { h->fail("hipSetDevice failed"); return OPH_ERR_DEVICE; }
    return OPH_OK;
}
static bool model_is_multispeaker(const oph_handle* h) {
    return h->dm.flags & (OPH_FLAG_SPK_AUDIO_DECODER_INPUT | OPH_FLAG_SPK_TEXT_ENCODER_INPUT | OPH_FLAG_SPK_TEXT_ENCODER_TOWARDS_END | OPH_FLAG_LCC | OPH_FLAG_SPK_AUDIO_ENCODER_INPUT);
}
// ends = get_text_lengths(L) (synthesize.py:242-247): a key position of the text, 0 ... max_N
static int check_ends(oph_handle* h, const int32_t* ends, int B) {
    for (int b = 0; b < B; ++b)
        if (ends[b] < 0 || ends[b] > h->dm.max_N) { h->fail("text end %d of utterance %d is outside the text (max_N = %d)", ends[b], b, h->dm.max_N); return OPH_ERR_INVALID; }
    return OPH_OK;
}
static int check_text(oph_handle* h, const int32_t* L, const int32_t* ends, const int32_t* spk, int B) {
    if (!L || !ends) { h->fail("null argument"); return OPH_ERR_INVALID; }
    if (int rc = check_ends(h, ends, B)) return rc;
    const bool ms = model_is_multispeaker(h);
    if (ms && !spk) { h->fail("multispeaker model needs speaker ids"); return OPH_ERR_INVALID; }
    const oph_dims& m = h->dm;
    for (long long i = 0; i < (long long)B * m.max_N; ++i)
        if (L[i] < 0 || L[i] >= m.vocab) { h->fail("text id %d out of range at %lld", L[i], i); return OPH_ERR_INVALID; }
    if (ms) for (int b = 0; b < B; ++b) if (spk[b] < 0 || spk[b] >= m.nspeakers) { h->fail("speaker id out of range"); return OPH_ERR_INVALID; }
    return OPH_OK;
}
// text of a batch into text buffer `slot`
static int upload_text(oph_handle* h, int slot, const int32_t* L, const int32_t* ends, const int32_t* spk, int B) {
    const oph_dims& m = h->dm;
    HIPCHK(h, hipMemcpyAsync(h->bL[slot], L, (size_t)B * m.max_N * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->bEnds[slot], ends, (size_t)B * 4, hipMemcpyHostToDevice, h->stream));
    if (model_is_multispeaker(h)) HIPCHK(h, hipMemcpyAsync(h->bSpk[slot], spk, (size_t)B * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return OPH_OK;
}

int oph_stage_text(oph_handle* h, const int32_t* L, const int32_t* ends, const int32_t* spk, int B) {
    int rc = check_ready(h, B);
    if (rc) return rc;
    if ((rc = check_text(h, L, ends, spk, B))) return rc;
    if ((rc = ensure_decode_state(h, B))) return rc;
    if (h->preenc_valid) { HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_preenc, 0)); HIPCHK(h, hipStreamSynchronize(h->stream)); }    // a running pre-encode still reads the other slot
    h->preenc_valid = false; h->next_staged = false;      // a directly staged text supersedes a staged "next" one
    h->kv_resident = false; h->y_resident = false; h->txt_ran = false; h->kv_pre = false;
    if ((rc = upload_text(h, h->txt, L, ends, spk, B))) return rc;
    select_tile(h, 0);
    return OPH_OK;
}

// make the staged "next" text the current one (its K,V may already be there: pre-encoded under the previous decode)
static int advance_text(oph_handle* h) {
    if (!h->next_staged || !h->txt_ran) return OPH_OK;
    h->txt ^= 1; h->next_staged = false; h->txt_ran = false; h->kv_pre = false;
    h->kv_resident = false; h->y_resident = false;
    if (h->preenc_valid) {
        h->kv_cur ^= 1; h->preenc_valid = false; h->kv_pre = true;
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_preenc, 0));
    }
    select_tile(h, 0);
    return OPH_OK;
}

// The text of the batch AFTER the staged one, into the second text slot (same batch size: the tiles and buffers are
// shared).  While the staged batch decodes (oph_run_resident / oph_run_host) its TextEnc runs on the SSRN partition; the
// run after that switches to it and starts decoding at once.  If the staged batch has already run, this call makes the
// previously staged next text current first and then stages the new one behind it, so a caller simply alternates
//     oph_stage_text(t0); oph_stage_text_next(t1);  loop { run(); oph_stage_text_next(t_{i+2}); }
int oph_stage_text_next(oph_handle* h, const int32_t* L, const int32_t* ends, const int32_t* spk, int B) {
    int rc = check_ready(h, B);
    if (rc) return rc;
    if (!h->bKV[0]) { h->fail("stage a first batch with oph_stage_text"); return OPH_ERR_STATE; }
    if (B != h->nB) { h->fail("the next batch must have the staged batch's size (%d)", h->nB); return OPH_ERR_INVALID; }
    if ((rc = check_text(h, L, ends, spk, B))) return rc;
    if ((rc = advance_text(h))) return rc;
    if (h->next_staged) { h->fail("a next batch is already staged and the current one has not run yet"); return OPH_ERR_STATE; }
    if ((rc = upload_text(h, h->txt ^ 1, L, ends, spk, B))) return rc;
    h->next_staged = true; h->next_B = B;
    return OPH_OK;
}

int oph_decode_steps(oph_handle* h, int t_begin, int t_end, int stop_mode, int32_t* steps_run) {
    if (!h || !h->bKV[0]) { if (h) h->fail("no staged batch"); return OPH_ERR_STATE; }
    if (stop_mode != OPH_STOP_REFERENCE && stop_mode != OPH_STOP_NEVER) { h->fail("unknown stop mode %d", stop_mode); return OPH_ERR_INVALID; }
    if (t_begin < 0 || t_end < t_begin || t_end > h->dm.max_T) { h->fail("steps [%d, %d) are outside [0, max_T = %d]", t_begin, t_end, h->dm.max_T); return OPH_ERR_INVALID; }
    HIPCHK(h, hipSetDevice(h->device));
    if (t_begin == 0) { begin_batch(h); return decode_batch(h, t_end, stop_mode, steps_run); }
    // resume (multi-GPU global stop): every tile continues from the step the batch stopped at; clear the local stops
    const int ntiles = (h->nB + TILE - 1) / TILE;
    int last = t_begin;
    int back = 0, ahead = 0;
    ssrn_margins(h, &back, &ahead);
    for (int j = 0; j < ntiles; ++j) {
        select_tile(h, j);
        if (h->tiles[j].steps != t_begin) { h->fail("resume at step %d, but tile %d stands at step %d", t_begin, j, h->tiles[j].steps); return OPH_ERR_STATE; }
        const int ctl1 = INT_MAX;
        HIPCHK(h, hipMemcpyAsync(h->d_ctl + 1, &ctl1, 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        h->tiles[j].ssrn_done = std::min(h->tiles[j].ssrn_done, std::max(0, t_begin - ahead));      // frames >= t_begin change: SSRN rows that saw them are stale
        h->tiles[j].z_copied = std::min(h->tiles[j].z_copied, h->tiles[j].ssrn_done);
        int32_t st = 0;
        const int rc = decode_range(h, t_begin, t_end, stop_mode, &st);
        if (rc) return rc;
        last = std::max(last, (int)st);
    }
    select_tile(h, 0);
    if (steps_run) *steps_run = last;
    return OPH_OK;
}

// Switch between sequential batches (everything joined when a call returns) and pipelined batches (the SSRN tail of a
// batch stays on the SSRN partition and overlaps the next batch).
static int set_pipelined(oph_handle* h, bool pipe) {
    if (pipe == h->pipelined) return OPH_OK;
    for (hipStream_t st : {h->stream, h->sdec, h->scone, h->sssrn, h->scopy}) if (st) HIPCHK(h, hipStreamSynchronize(st));
    if (!pipe) h->buf = 0;
    h->pipelined = pipe;
    h->ssrn_inflight[0] = h->ssrn_inflight[1] = false;
    select_tile(h, h->tile);
    return OPH_OK;
}

int oph_run_ssrn_resident(oph_handle* h) {
    if (!h || !h->bKV[0]) { if (h) h->fail("no staged batch"); return OPH_ERR_STATE; }
    HIPCHK(h, hipSetDevice(h->device));
    g_cur = h->stream;
    return finish_ssrn(h);
}

int oph_run_resident(oph_handle* h, int stop_mode, int run_ssrn, int32_t* steps_run) {
    if (!h || !h->bKV[0]) { if (h) h->fail("no staged batch"); return OPH_ERR_STATE; }
    if (stop_mode != OPH_STOP_REFERENCE && stop_mode != OPH_STOP_NEVER) { h->fail("unknown stop mode %d", stop_mode); return OPH_ERR_INVALID; }
    if (run_ssrn < 0 || run_ssrn > 2) { h->fail("run_ssrn must be 0, 1 or 2"); return OPH_ERR_INVALID; }
    HIPCHK(h, hipSetDevice(h->device));
    // run_ssrn: 0 = Text2Mel only; 1 = SSRN too, joined when the call returns; 2 = pipelined batches -- the SSRN tail of
    // THIS batch (what its streamed chunks have not covered when the decode ends) stays queued on the SSRN partition and
    // overlaps the next call's decode (Y/Z ping-pong); oph_synchronize / oph_fetch_* / oph_timer_stop join it.
    const bool pipe = run_ssrn == 2;
    int rc = set_pipelined(h, pipe);
    if (rc) return rc;
    g_cur = h->stream;
    if ((rc = advance_text(h))) return rc;       // the current text has run: a text staged with oph_stage_text_next takes its place
    begin_batch(h);
    select_tile(h, 0);
    if (h->kv_pre) h->n_preenc_used++;
    else if ((rc = run_encode(h))) return rc;
    h->kv_pre = false; h->txt_ran = true;
    h->kv_resident = true;
    h->want_preenc = !h->opt.no_preencode;
    const bool spec_saved = h->spec_ssrn;
    h->spec_ssrn = run_ssrn != 0;
    rc = decode_batch(h, h->dm.max_T, stop_mode, steps_run);
    h->want_preenc = false; h->spec_ssrn = spec_saved;
    if (rc) return rc;
    h->y_resident = true;
    if (run_ssrn) rc = finish_ssrn(h);
    return rc;
}

int oph_set_ssrn_precision(oph_handle* h, int mode) {
    if (!h || mode < 0 || mode > 4) return OPH_ERR_INVALID;       // 3, 4: measurement only -- split-fp16 with 2 products / 1 product
    if (h->guard_ssrn && mode != 0) { h->fail("an SSRN weight is outside fp16's range: only the fp32-operand MFMA (mode 0) is offered"); return OPH_ERR_UNSUPPORTED; }
    h->ssrn_prec = mode; h->chunk_ms = 0.f;
    return OPH_OK;
}
// which: 0 SSRN (= oph_set_ssrn_precision), 1 the cone's two many-row contractions, 2 TextEnc.  mode: 0 fp32 MFMA, 2 split-fp16 x3
// (fp32-class), 1 split-bf16 x3 (SSRN; the cone only if the handle was created under OPH_CONE_PREC=1)
int oph_set_precision(oph_handle* h, int which, int mode) {
    if (!h || mode < 0 || mode > 2) return OPH_ERR_INVALID;
    if (mode != 0 && ((which == 0 && h->guard_ssrn) || (which == 1 && h->guard_cone) || (which == 2 && h->guard_text))) {
        h->fail("a weight of that net is outside fp16's range: only the fp32-operand MFMA (mode 0) is offered"); return OPH_ERR_UNSUPPORTED;
    }
    if (which == 0) { h->ssrn_prec = mode; h->chunk_ms = 0.f; return OPH_OK; }
    if (which == 1) {
        if (mode == 1 && !(h->n_hc_dec > 1 && h->audiodec[h->dec_pre].Wh)) { h->fail("the cone's bf16 planes were not built (create the handle under OPH_CONE_PREC=1)"); return OPH_ERR_STATE; }
        h->cone_prec = mode; return OPH_OK;
    }
    if (which == 2 && mode != 1) { h->textenc_prec = mode; return OPH_OK; }
    h->fail("bad precision selector");
    return OPH_ERR_INVALID;
}
// what the pipeline actually did since the handle was created (tests and bench.py assert on these):
// [0] TextEnc evaluations  [1] runs that found their K,V pre-encoded  [2] SSRN chunks launched while a decode was running
// [3] whole-decode launches  [4] fallbacks from the whole-decode launch to two launches per step  [5] tiles resumed to the batch's stop step
int oph_get_counters(oph_handle* h, int64_t* out, int n) {
    if (!h || !out) return OPH_ERR_INVALID;
    const long long v[8] = {h->n_textenc, h->n_preenc_used, h->n_chunks_streamed, h->n_loop_decodes, h->n_loop_fallbacks, h->n_tile_resumes, h->n_cone_loops,
                            (long long)((h->guard_ssrn ? 1 : 0) | (h->guard_cone ? 2 : 0) | (h->guard_text ? 4 : 0))};
    for (int i = 0; i < n && i < 8; ++i) out[i] = v[i];
    return OPH_OK;
}
// Where the speculative SSRN of the NEXT oph_text2mel copies its rows while the decoder is still running: a host buffer of
// (B, r*max_T, full_dim) floats (pinned, oph_host_alloc, for the copies to be asynchronous).  oph_ssrn(Y = NULL, ..., Z = that
// pointer) then only has the tail left to compute and copy.  NULL clears it.
int oph_set_mag_destination(oph_handle* h, float* Z) {
    if (!h) return OPH_ERR_INVALID;
    h->z_spec = Z; h->z_spec_gen = 0;
    return OPH_OK;
}
int oph_set_streaming(oph_handle* h, int on) {
    if (!h || on < 0) return OPH_ERR_INVALID;
    h->spec_ssrn = on != 0;
    if (on >= 2) h->opt.ssrn_chunk = on;          // mel frames per streamed chunk (1 keeps the current size)
    return OPH_OK;
}

int oph_synchronize(oph_handle* h) {
    if (!h) return OPH_ERR_INVALID;
    if (h->sssrn) HIPCHK(h, hipStreamSynchronize(h->sssrn));
    if (h->scopy) HIPCHK(h, hipStreamSynchronize(h->scopy));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    return OPH_OK;
}

int oph_fetch_kv(oph_handle* h, float* K, float* V) {
    if (!h || !h->bKV[0]) { if (h) h->fail("no staged batch"); return OPH_ERR_STATE; }
    const oph_dims& m = h->dm;
    const size_t rows = (size_t)h->nB * m.max_N, w = (size_t)m.d * 4;
    const float* kv = h->bKV[h->kv_cur];
    if (K) HIPCHK(h, hipMemcpy2DAsync(K, w, kv, 2 * w, w, rows, hipMemcpyDeviceToHost, h->stream));
    if (V) HIPCHK(h, hipMemcpy2DAsync(V, w, kv + m.d, 2 * w, w, rows, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return OPH_OK;
}

int oph_fetch_mel(oph_handle* h, float* Y, int32_t* t_ends, float* alignments) {
    if (!h || !h->bKV[0]) { if (h) h->fail("no staged batch"); return OPH_ERR_STATE; }
    const oph_dims& m = h->dm;
    if (Y) HIPCHK(h, hipMemcpy2DAsync(Y, (size_t)m.n_mels * 4, h->bYout[h->buf], (size_t)h->ldy * 4, (size_t)m.n_mels * 4,
                                      (size_t)h->nB * m.max_T, hipMemcpyDeviceToHost, h->stream));
    if (t_ends) HIPCHK(h, hipMemcpyAsync(t_ends, h->bTends, (size_t)h->nB * 4, hipMemcpyDeviceToHost, h->stream));
    if (alignments) HIPCHK(h, hipMemcpyAsync(alignments, h->bAlign, (size_t)h->nB * m.max_N * m.max_T * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return OPH_OK;
}

int oph_fetch_mag(oph_handle* h, float* Z) {
    if (!h || !h->bKV[0] || !Z) { if (h) h->fail("no staged batch / null"); return OPH_ERR_STATE; }
    if (h->sssrn) HIPCHK(h, hipStreamSynchronize(h->sssrn));
    const oph_dims& m = h->dm;
    HIPCHK(h, hipMemcpyAsync(Z, h->bZ[h->buf], (size_t)h->nB * m.max_T * m.r * m.full_dim * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return OPH_OK;
}

int oph_device_mag(oph_handle* h, const float** d_mag, int64_t* utt_stride, int32_t* B) {
    if (!h || !h->bKV[0] || !d_mag) { if (h) h->fail("no staged batch / null"); return OPH_ERR_STATE; }
    if (h->sssrn) HIPCHK(h, hipStreamSynchronize(h->sssrn));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const oph_dims& m = h->dm;
    *d_mag = h->bZ[h->buf];
    if (utt_stride) *utt_stride = (int64_t)m.max_T * m.r * m.full_dim;
    if (B) *B = h->nB;
    return OPH_OK;
}

// One whole batch host -> host: the staged text (oph_stage_text, or the one staged with oph_stage_text_next during the
// previous call) through TextEnc, decode and SSRN, with every result copied into the caller's buffers (any of them may be
// NULL) as it becomes final -- SSRN rows chunk by chunk and Y / alignments at the end of the decode, on the copy stream, under
// the work that is still running.  Pinned buffers (oph_host_alloc) make those copies asynchronous DMA.  What the
// reference's two clocks bracket (synthesize.py:553-576).
int oph_run_host(oph_handle* h, int stop_mode, float* K, float* V, float* Y, int32_t* t_ends, float* alignments, float* Z, int32_t* steps_run) {
    if (!h || !h->bKV[0]) { if (h) h->fail("no staged batch"); return OPH_ERR_STATE; }
    if (stop_mode != OPH_STOP_REFERENCE && stop_mode != OPH_STOP_NEVER) { h->fail("unknown stop mode %d", stop_mode); return OPH_ERR_INVALID; }
    HIPCHK(h, hipSetDevice(h->device));
    int rc = set_pipelined(h, false);
    if (rc) return rc;
    const oph_dims& m = h->dm;
    g_cur = h->stream;
    if ((rc = advance_text(h))) return rc;
    begin_batch(h);
    select_tile(h, 0);
    if (h->kv_pre) h->n_preenc_used++;
    else if ((rc = run_encode(h))) return rc;
    h->kv_pre = false; h->txt_ran = true;
    h->kv_resident = true;
    if (K || V) {       // K,V leave on the copy stream while the decode starts
        HIPCHK(h, hipEventRecord(h->ev_copy, h->stream));
        HIPCHK(h, hipStreamWaitEvent(h->scopy, h->ev_copy, 0));
        const size_t rows = (size_t)h->nB * m.max_N, w = (size_t)m.d * 4;
        const float* kv = h->bKV[h->kv_cur];
        if (K) HIPCHK(h, hipMemcpy2DAsync(K, w, kv, 2 * w, w, rows, hipMemcpyDeviceToHost, h->scopy));
        if (V) HIPCHK(h, hipMemcpy2DAsync(V, w, kv + m.d, 2 * w, w, rows, hipMemcpyDeviceToHost, h->scopy));
    }
    h->want_preenc = !h->opt.no_preencode;
    const bool spec_saved = h->spec_ssrn;
    h->spec_ssrn = Z != nullptr;
    h->z_host = Z;
    rc = decode_batch(h, m.max_T, stop_mode, steps_run);
    h->want_preenc = false; h->spec_ssrn = spec_saved;
    if (rc) { h->z_host = nullptr; return rc; }
    h->y_resident = true;
    // Y, t_ends, alignments: final now (the API stream has joined the decode streams)
    HIPCHK(h, hipEventRecord(h->ev_copy, h->stream));
    HIPCHK(h, hipStreamWaitEvent(h->scopy, h->ev_copy, 0));
    if (Y) HIPCHK(h, hipMemcpy2DAsync(Y, (size_t)m.n_mels * 4, h->bYout[h->buf], (size_t)h->ldy * 4, (size_t)m.n_mels * 4, (size_t)h->nB * m.max_T, hipMemcpyDeviceToHost, h->scopy));
    if (t_ends) HIPCHK(h, hipMemcpyAsync(t_ends, h->bTends, (size_t)h->nB * 4, hipMemcpyDeviceToHost, h->scopy));
    if (alignments) HIPCHK(h, hipMemcpyAsync(alignments, h->bAlign, (size_t)h->nB * m.max_N * m.max_T * 4, hipMemcpyDeviceToHost, h->scopy));
    if (Z) rc = finish_ssrn(h);
    h->z_host = nullptr;
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipStreamSynchronize(h->sssrn));
    HIPCHK(h, hipStreamSynchronize(h->scopy));
    return OPH_OK;
}

// ---- host-buffer session calls ---------------------------------------------------------------
int oph_encode_text(oph_handle* h, const int32_t* L, const int32_t* spk, int B, float* K, float* V) {
    int rc = check_ready(h, B);
    if (rc) return rc;
    if (!L || !K || !V) { h->fail("null argument"); return OPH_ERR_INVALID; }
    // ends are not needed by TextEnc; stage zeros (get_text_lengths stays with the caller, synthesize.py:556)
    std::vector<int32_t> ends(B, 0), spk0(B, 0);
    if ((rc = oph_stage_text(h, L, ends.data(), spk ? spk : spk0.data(), B))) return rc;
    g_cur = h->stream;
    if ((rc = run_encode(h))) return rc;
    h->kv_resident = true;          // oph_text2mel(K = NULL, V = NULL) decodes from these
    return oph_fetch_kv(h, K, V);
}

// shared front of the two decode entry points: K,V (or the resident ones), ends / speakers of the batch
static int stage_decode_inputs(oph_handle* h, const float* K, const float* V, bool need_k, const int32_t* ends, const int32_t* spk, int B) {
    int rc;
    const bool ms = h->dm.flags & (OPH_FLAG_SPK_AUDIO_DECODER_INPUT | OPH_FLAG_LCC | OPH_FLAG_SPK_AUDIO_ENCODER_INPUT);
    if (ms && !spk) { h->fail("multispeaker model needs speaker ids"); return OPH_ERR_INVALID; }
    const oph_dims& m = h->dm;
    if (!K && !V) {
        if (!h->kv_resident || B != h->nB) { h->fail("K = V = NULL asks for the K,V oph_encode_text left in HBM, but there are none for a batch of %d", B); return OPH_ERR_STATE; }
    } else {
        if ((need_k && !K) || !V) { h->fail("null argument"); return OPH_ERR_INVALID; }
        if ((rc = ensure_decode_state(h, B))) return rc;
        const size_t rows = (size_t)B * m.max_N, w = (size_t)m.d * 4;
        float* kv = h->bKV[h->kv_cur];
        if (K) HIPCHK(h, hipMemcpy2DAsync(kv, 2 * w, K, w, w, rows, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpy2DAsync(kv + m.d, 2 * w, V, w, w, rows, hipMemcpyHostToDevice, h->stream));
        h->kv_resident = false;
    }
    if (ends) {
        if ((rc = check_ends(h, ends, B))) return rc;
        HIPCHK(h, hipMemcpyAsync(h->bEnds[h->txt], ends, (size_t)B * 4, hipMemcpyHostToDevice, h->stream));
    }
    if (ms) {
        for (int b = 0; b < B; ++b) if (spk[b] < 0 || spk[b] >= m.nspeakers) { h->fail("speaker id out of range"); return OPH_ERR_INVALID; }
        HIPCHK(h, hipMemcpyAsync(h->bSpk[h->txt], spk, (size_t)B * 4, hipMemcpyHostToDevice, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return OPH_OK;
}

int oph_text2mel(oph_handle* h, const float* K, const float* V, const int32_t* ends, const int32_t* spk, int B,
                 int stop_mode, float* Y, int32_t* t_ends, float* alignments, int32_t* steps_run) {
    int rc = check_ready(h, B);
    if (rc) return rc;
    if (!ends) { h->fail("null argument"); return OPH_ERR_INVALID; }
    if (stop_mode != OPH_STOP_REFERENCE && stop_mode != OPH_STOP_NEVER) { h->fail("unknown stop mode %d", stop_mode); return OPH_ERR_INVALID; }
    if ((rc = set_pipelined(h, false))) return rc;
    if ((rc = stage_decode_inputs(h, K, V, true, ends, spk, B))) return rc;
    begin_batch(h);
    h->z_host = h->spec_ssrn ? h->z_spec : nullptr;       // the speculative SSRN's rows leave for the host as they are produced
    h->z_spec_gen = h->z_host ? h->batch_gen : 0;         // this batch's rows are the ones in z_spec
    rc = decode_batch(h, h->dm.max_T, stop_mode, steps_run);
    h->z_host = nullptr;
    if (rc) return rc;
    h->y_resident = true;           // oph_ssrn(Y = NULL) continues from here
    return oph_fetch_mel(h, Y, t_ends, alignments);
}

int oph_text2mel_durations(oph_handle* h, const float* K, const float* V, const float* durations, const int32_t* spk,
                           int B, int n_steps, float* Y, int32_t* t_ends, float* alignments, int32_t* steps_run) {
    int rc = check_ready(h, B);
    if (rc) return rc;
    if (!durations) { h->fail("null argument"); return OPH_ERR_INVALID; }
    if ((rc = set_pipelined(h, false))) return rc;
    if ((rc = stage_decode_inputs(h, K, V, false, nullptr, spk, B))) return rc;
    const oph_dims& m = h->dm;
    const int ntiles = (B + TILE - 1) / TILE;
    // selection matrix -> key index per (t, b); only hard 0/1 rows (what data_load.py:243-251 produces) are supported
    std::vector<std::vector<int>> ptab(ntiles, std::vector<int>((size_t)m.max_T * TILE, -1));
    std::vector<int32_t> tends(B, 0);
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < m.max_T; ++t) {
            const float* row = durations + ((size_t)b * m.max_T + t) * m.max_N;
            int key = -1;
            for (int n = 0; n < m.max_N; ++n) {
                if (row[n] == 0.0f) continue;
                if (row[n] != 1.0f || key >= 0) { h->fail("durations row (b=%d, t=%d) is not a 0/1 selection of at most one key", b, t); return OPH_ERR_UNSUPPORTED; }
                key = n;
            }
            if (key >= 0) { ptab[b / TILE][(size_t)t * TILE + b % TILE] = key; tends[b]++; }
        }
    int steps = n_steps;
    if (steps <= 0) {
        int mx = 0;
        for (int b = 0; b < B; ++b) mx = std::max(mx, (int)tends[b]);
        steps = std::min((int)m.max_T, mx + 1);          // synthesize.py:211-216: the step at which j >= max(t_ends) still runs
    }
    steps = std::min(steps, (int)m.max_T);
    for (int j = 0; j < ntiles; ++j)
        HIPCHK(h, hipMemcpyAsync(h->tiles[j].d_ptab, ptab[j].data(), ptab[j].size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    begin_batch(h);
    h->fixed_att = true;
    rc = decode_batch(h, steps, OPH_STOP_NEVER, nullptr);
    h->fixed_att = false;
    if (rc) return rc;
    h->y_resident = true;
    if ((rc = oph_fetch_mel(h, Y, nullptr, alignments))) return rc;
    if (t_ends) std::copy(tends.begin(), tends.end(), t_ends);
    if (steps_run) *steps_run = steps;
    return OPH_OK;
}

// One evaluation of the synthesis graph at fixed feeds -- what ONE sess.run of the reference's loop computes
// (synthesize.py:181-183 with the feed dict of :172): S = shift(mels) -> AudioEnc over all max_T positions ->
// Attention of every position under the ONE mask prev_max (networks.py:300-319) -> AudioDec -> logits / sigmoid.
// O(max_T) work per call: the debug / fetch surface of architectures.py:188-239 (g.Q, g.R, g.Y_logits, ...), not the
// decode loop (oph_text2mel).  Batched kernels: conv_gemm_f32 + ln_rows per layer, attn_rows.
int oph_text2mel_graph(oph_handle* h, const float* K, const float* V, const float* mels, const int32_t* prev_max,
                       const int32_t* ends, const int32_t* spk, int B,
                       float* Q, float* R, float* Y_logits, float* Y, float* alignments, int32_t* max_attentions) {
    int rc = check_ready(h, B);
    if (rc) return rc;
    if (!K || !V || !mels || !prev_max) { h->fail("null argument"); return OPH_ERR_INVALID; }
    const oph_dims& m = h->dm;
    if ((m.flags & OPH_FLAG_NO_MONOTONIC) && !ends) { h->fail("turn_off_monotonic_for_synthesis needs the text lengths"); return OPH_ERR_INVALID; }
    for (int b = 0; b < B; ++b) if (prev_max[b] < 0 || prev_max[b] >= m.max_N) { h->fail("prev_max_attentions out of range"); return OPH_ERR_INVALID; }
    if ((rc = set_pipelined(h, false))) return rc;
    if ((rc = stage_decode_inputs(h, K, V, true, ends, spk, B))) return rc;
    h->y_resident = false;
    g_cur = h->stream;
    const int T = m.max_T, d = m.d, ldy = h->ldy;
    // the fed prev_max_attentions of all utterances: one int per utterance, batch order (bTends doubles as the feed buffer)
    int* d_pm = h->bTends;
    HIPCHK(h, hipMemcpyAsync(d_pm, prev_max, (size_t)B * 4, hipMemcpyHostToDevice, h->stream));
    // S = concat(zeros, mels[:, :-1])  (architectures.py:191), rows padded to the first layer's K
    std::vector<float> S((size_t)B * T * ldy, 0.f);
    for (int b = 0; b < B; ++b)
        for (int t = 1; t < T; ++t)
            std::copy(mels + ((size_t)b * T + t - 1) * m.n_mels, mels + ((size_t)b * T + t) * m.n_mels, S.begin() + ((size_t)b * T + t) * ldy);
    HIPCHK(h, hipMemcpyAsync(h->actA, S.data(), S.size() * 4, hipMemcpyHostToDevice, h->stream));
    int ldq = 0;
    float* dQ = run_batched(h, h->audioenc, h->actA, ldy, B, T, 0, 0, nullptr, 0, 0, &ldq, nullptr);
    // attention over all positions, one mask per utterance; R' rows go to the workspace the encoder did not end in
    float* dR = dQ == h->actA ? h->actB : h->actA;
    const float* kv = h->bKV[h->kv_cur];
    HIPCHK(h, hipMemsetAsync(h->bAlign, 0, (size_t)h->nBpad * m.max_N * T * 4, h->stream));
    AttnRowsArgs a{};
    a.mode = 1; a.Q = dQ; a.ldq = ldq; a.K = kv; a.V = kv + d; a.ldkv = 2 * d; a.N = m.max_N; a.d = d; a.win = m.attention_win_size;
    a.p = d_pm; a.B = B; a.Bpad = h->nBpad; a.nrows = B * T; a.T = T; a.R = dR; a.ldr = 2 * d; a.align = h->bAlign; a.amax = h->d_amax;
    if (m.flags & OPH_FLAG_NO_MONOTONIC) a.ends = h->bEnds[h->txt];
    launch_attn_rows(a, h->stream);
    if (Q) HIPCHK(h, hipMemcpy2DAsync(Q, (size_t)d * 4, dQ, (size_t)ldq * 4, (size_t)d * 4, (size_t)B * T, hipMemcpyDeviceToHost, h->stream));
    if (R) HIPCHK(h, hipMemcpyAsync(R, dR, (size_t)B * T * 2 * d * 4, hipMemcpyDeviceToHost, h->stream));
    if (alignments) HIPCHK(h, hipMemcpyAsync(alignments, h->bAlign, (size_t)B * m.max_N * T * 4, hipMemcpyDeviceToHost, h->stream));
    std::vector<long long> amax((size_t)B * T);
    HIPCHK(h, hipMemcpyAsync(amax.data(), h->d_amax, amax.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));     // Q may live in the buffer the decoder overwrites next
    // AudioDec; its last layer's LayerNorm rows are stored twice: squashed (g.Y) and as they are (g.Y_logits)
    int ldl = 0;
    BatchedIO io{};
    float* dlogits = h->raw;         // free once the last layer's epilogues have run (they read it row by row: use the far end)
    const int ldn = round_up(m.n_mels, 32);
    dlogits = h->raw + (h->raw_elems - (size_t)B * T * ldn);
    io.final_logits = dlogits;
    std::vector<Layer> dec = h->audiodec;
    dec.back().act = ACT_SIGMOID;                   // squash_output_t2m (networks.py:430-431)
    float* dY = run_batched(h, dec, dR, 2 * d, B, T, 0, 0, nullptr, 0, 0, &ldl, nullptr, io);
    if (Y) HIPCHK(h, hipMemcpy2DAsync(Y, (size_t)m.n_mels * 4, dY, (size_t)ldl * 4, (size_t)m.n_mels * 4, (size_t)B * T, hipMemcpyDeviceToHost, h->stream));
    if (Y_logits) HIPCHK(h, hipMemcpy2DAsync(Y_logits, (size_t)m.n_mels * 4, dlogits, (size_t)ldl * 4, (size_t)m.n_mels * 4, (size_t)B * T, hipMemcpyDeviceToHost, h->stream));
    hipError_t e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess || (e = hipGetLastError()) != hipSuccess) { h->fail("graph evaluation failed: %s", hipGetErrorString(e)); return OPH_ERR_DEVICE; }
    if (max_attentions) for (size_t i = 0; i < amax.size(); ++i) max_attentions[i] = (int32_t)amax[i];
    // bTends was borrowed for the fed prev_max: put the "not ended" state back
    launch_fill_int(h->bTends, m.max_T, h->nBpad, h->stream);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return OPH_OK;
}

// oph_ssrn / oph_ssrn_logits.  Y == NULL: the mel frames the last decode call left in HBM (B and T must be that batch's);
// whatever its streamed SSRN has not covered yet is computed now.
static int ssrn_common(oph_handle* h, const float* Y, int B, int T, float* Z, float* Z_logits) {
    int rc = check_ready(h, B);
    if (rc) return rc;
    if (!Z || T < 1) { h->fail("bad argument"); return OPH_ERR_INVALID; }
    const oph_dims& m = h->dm;
    if (T > m.max_T) { h->fail("T=%d exceeds max_T=%d", T, m.max_T); return OPH_ERR_INVALID; }
    if (!Y && !Z_logits) {
        if (!h->y_resident || B != h->nB || T != m.max_T) { h->fail("Y = NULL asks for the mel frames the last decode left in HBM, but there are none for B=%d, T=%d", B, T); return OPH_ERR_STATE; }
        g_cur = h->stream;
        if (Z == h->z_spec && Z != nullptr && h->z_spec_gen == h->batch_gen) {
            // the chunks streamed during THIS batch's decode are already in Z (oph_set_mag_destination; the batch generation tells a
            // later batch on the same handle apart): compute and copy what is left -- every row past the copied frontier
            h->z_host = Z;
            rc = finish_ssrn(h);
            h->z_host = nullptr;
            if (rc) return rc;
            HIPCHK(h, hipStreamSynchronize(h->stream));
            HIPCHK(h, hipStreamSynchronize(h->sssrn));
            HIPCHK(h, hipStreamSynchronize(h->scopy));
            return OPH_OK;
        }
        if ((rc = finish_ssrn(h))) return rc;
        return oph_fetch_mag(h, Z);
    }
    if (!Y) { h->fail("the logits fetch needs the mel frames as an argument"); return OPH_ERR_INVALID; }
    if ((rc = ensure_batched_capacity(h, B))) return rc;
    g_cur = h->stream;
    const int ldy = round_up(m.n_mels, 32);
    float* dY = nullptr; float* dZ = nullptr; float* dZl = nullptr;
    const size_t zn = (size_t)B * T * m.r * m.full_dim;
    HIPCHK(h, hipMalloc((void**)&dY, (size_t)B * T * m.n_mels * 4));
    if (hipMalloc((void**)&dZ, zn * 4) != hipSuccess || (Z_logits && hipMalloc((void**)&dZl, zn * 4) != hipSuccess)) {
        hipFree(dY); if (dZ) hipFree(dZ); (void)hipGetLastError(); h->fail("out of device memory for the SSRN output"); return OPH_ERR_DEVICE;
    }
    hipError_t e = hipMemcpyAsync(dY, Y, (size_t)B * T * m.n_mels * 4, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) {
        launch_pad_rows(dY, m.n_mels, h->actB, ldy, (long long)B * T, m.n_mels, h->stream);
        rc = run_ssrn_on(h, h->actB, ldy, B, T, dZ, 0, dZl);
        e = hipMemcpyAsync(Z, dZ, zn * 4, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess && Z_logits) e = hipMemcpyAsync(Z_logits, dZl, zn * 4, hipMemcpyDeviceToHost, h->stream);
    }
    const hipError_t es = hipStreamSynchronize(h->stream);
    hipFree(dY); hipFree(dZ); if (dZl) hipFree(dZl);
    if (e != hipSuccess || es != hipSuccess) { h->fail("ssrn failed: %s", hipGetErrorString(e != hipSuccess ? e : es)); return OPH_ERR_DEVICE; }
    return rc;
}
int oph_ssrn(oph_handle* h, const float* Y, int B, int T, float* Z) { return ssrn_common(h, Y, B, T, Z, nullptr); }
int oph_ssrn_logits(oph_handle* h, const float* Y, int B, int T, float* Z, float* Z_logits) {
    if (h && !Z_logits) { h->fail("null argument"); return OPH_ERR_INVALID; }
    return ssrn_common(h, Y, B, T, Z, Z_logits);
}

// ---- measurement ----------------------------------------------------------------------------
int oph_timer_start(oph_handle* h) {
    if (!h) return OPH_ERR_INVALID;
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    return OPH_OK;
}
int oph_timer_stop(oph_handle* h, float* ms) {
    if (!h || !ms) return OPH_ERR_INVALID;
    if (h->pipelined && h->sssrn) {      // join the pipelined SSRN stream first
        HIPCHK(h, hipEventRecord(h->ev_dec_done, h->sssrn));
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_dec_done, 0));
    }
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipEventSynchronize(h->ev1));
    HIPCHK(h, hipEventElapsedTime(ms, h->ev0, h->ev1));
    return OPH_OK;
}
// Device-side witness of the whole-decode launches since the last reset: every workgroup stores the constant 100 MHz clock when it
// enters (minimum kept) and when it leaves (maximum kept); *total_us = sum over launches of (last out - first in).
int oph_loop_clock(oph_handle* h, int64_t* launches, double* total_us, int reset) {
    if (!h) return OPH_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    if (h->sdec) HIPCHK(h, hipStreamSynchronize(h->sdec));
    drain_loop_clock(h);
    if (launches) *launches = h->clk_launches;
    if (total_us) *total_us = h->clk_total_us;
    if (reset) { h->clk_launches = 0; h->clk_total_us = 0; }
    return OPH_OK;
}
int oph_profile_enable(oph_handle* h, int on) { if (!h) return OPH_ERR_INVALID; h->profiling = on == 2 ? 2 : (on != 0); return OPH_OK; }
int oph_profile_reset(oph_handle* h) {
    if (!h) return OPH_ERR_INVALID;
    hipStreamSynchronize(h->stream);
    for (auto& pc : h->prof) { pc.launches = 0; pc.bytes = pc.flops = pc.ms = 0; pc.used = 0; }
    return OPH_OK;
}
int oph_profile_count(const oph_handle* h) { return h ? PC_COUNT : OPH_ERR_INVALID; }
int oph_profile_get(oph_handle* h, int index, char* name, int name_cap, int64_t* launches, double* total_ms,
                    double* alg_bytes, double* alg_flops) {
    if (!h || index < 0 || index >= PC_COUNT) return OPH_ERR_INVALID;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipStreamSynchronize(h->scone));
    HIPCHK(h, hipStreamSynchronize(h->sdec));
    if (h->sssrn) HIPCHK(h, hipStreamSynchronize(h->sssrn));
    ProfClass& pc = h->prof[index];
    double ms = 0;
    for (size_t i = 0; i < pc.used; ++i) {
        float t = 0;
        if (hipEventElapsedTime(&t, pc.ev[i].first, pc.ev[i].second) == hipSuccess) ms += t;
    }
    pc.ms = ms;
    if (name && name_cap > 0) { strncpy(name, pc.name, name_cap - 1); name[name_cap - 1] = 0; }
    if (launches) *launches = pc.launches;
    if (total_ms) *total_ms = ms;
    if (alg_bytes) *alg_bytes = pc.bytes;
    if (alg_flops) *alg_flops = pc.flops;
    return OPH_OK;
}

// ---- per-operator entry points (unit parity) ---------------------------------------------------
const char* oph_op_last_error(void) { return g_op_error.c_str(); }

}  // extern "C"

namespace {
struct OpCtx {
    hipStream_t s = nullptr;
    std::vector<void*> bufs;
    bool ok = true;
    explicit OpCtx(int device) {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n || hipSetDevice(device) != hipSuccess ||
            hipStreamCreate(&s) != hipSuccess) { ok = false; g_op_error = "no usable HIP device (no CPU fallback)"; }
    }
    ~OpCtx() { for (void* p : bufs) hipFree(p); if (s) hipStreamDestroy(s); }
    template <class T> T* alloc(size_t n) {
        void* p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) { ok = false; g_op_error = "hipMalloc failed"; return nullptr; }
        hipMemsetAsync(p, 0, std::max<size_t>(n, 1) * sizeof(T), s);
        bufs.push_back(p);
        return (T*)p;
    }
    template <class T> T* up(const T* src, size_t n) {
        T* p = alloc<T>(n);
        if (p) hipMemcpyAsync(p, src, n * sizeof(T), hipMemcpyHostToDevice, s);
        return p;
    }
    float* up_pad(const float* src, size_t n, size_t padto) {
        std::vector<float> t((n + padto - 1) / padto * padto, 0.f);
        std::copy(src, src + n, t.begin());
        float* p = up(t.data(), t.size());
        hipStreamSynchronize(s);
        return p;
    }
    int finish() {
        hipError_t e = hipStreamSynchronize(s);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) { g_op_error = hipGetErrorString(e); return OPH_ERR_DEVICE; }
        return ok ? OPH_OK : OPH_ERR_DEVICE;
    }
};
}  // namespace

extern "C" {

int oph_op_embed(int device, const int32_t* ids, int64_t n, const float* table, int vocab, int units, float* out) {
    OpCtx c(device);
    if (!c.ok) return OPH_ERR_DEVICE;
    for (int64_t i = 0; i < n; ++i) if (ids[i] < 0 || ids[i] >= vocab) { g_op_error = "id out of range"; return OPH_ERR_INVALID; }
    const int ldo = round_up(units, 4);
    int* dids = c.up(ids, (size_t)n);
    float* dt = c.up(table, (size_t)vocab * units);
    float* dout = c.alloc<float>((size_t)n * ldo);
    if (!c.ok) return OPH_ERR_DEVICE;
    launch_embed(dids, n, dt, units, dout, ldo, c.s);
    hipMemcpy2DAsync(out, (size_t)units * 4, dout, (size_t)ldo * 4, (size_t)units * 4, (size_t)n, hipMemcpyDeviceToHost, c.s);
    return c.finish();
}

int oph_op_layernorm(int device, const float* x, int64_t rows, int C, const float* gamma, const float* beta, float* y) {
    OpCtx c(device);
    if (!c.ok) return OPH_ERR_DEVICE;
    if (C < 1 || C > 1280) { g_op_error = "C out of range (<=1280)"; return OPH_ERR_UNSUPPORTED; }
    const int ld = round_up(C, 128);
    float* dx = c.up(x, (size_t)rows * C);
    float* dh = c.alloc<float>((size_t)rows * ld);
    float* dy = c.alloc<float>((size_t)rows * C);
    float* g = c.up_pad(gamma, C, 256); float* b = c.up_pad(beta, C, 256);
    if (!c.ok) return OPH_ERR_DEVICE;
    launch_pad_rows(dx, C, dh, ld, rows, C, c.s);
    EpiArgs e{};
    e.H = dh; e.ldh = ld; e.M = (int)rows; e.C = C; e.mode = PRE_CONV; e.act = ACT_NONE; e.g1 = g; e.b1 = b; e.Y = dy; e.ldy = C; e.ypad = C;
    launch_epilogue(e, c.s);
    hipMemcpyAsync(y, dy, (size_t)rows * C * 4, hipMemcpyDeviceToHost, c.s);
    return c.finish();
}

static int op_conv_common(int device, const float* x, int B, int T, int Cin, int Cout, int size, int rate, int padding,
                          const float* kernel, const float* bias, const float* g1, const float* b1, const float* g2,
                          const float* b2, int act, bool is_hc, float* y) {
    OpCtx c(device);
    if (!c.ok) return OPH_ERR_DEVICE;
    if (size != 1 && size != 3) { g_op_error = "size must be 1 or 3"; return OPH_ERR_UNSUPPORTED; }
    if (Cout > 1280 || (is_hc && (Cout > 1024 || Cout % 4 || Cin != Cout))) { g_op_error = "channels out of range"; return OPH_ERR_UNSUPPORTED; }
    const int kc = round_up(Cin, 32), N = is_hc ? 2 * Cout : Cout, Nalloc = round_up(N, 128), M = B * T;
    std::vector<float> wt = pack_conv(kernel, size, Cin, N, kc, Nalloc);
    float* dx = c.up(x, (size_t)M * Cin);
    float* dxp = c.alloc<float>((size_t)M * kc);
    float* dw = c.up(wt.data(), wt.size());
    float* dbias = c.up_pad(bias, N, Nalloc);
    float* dh = c.alloc<float>((size_t)M * Nalloc);
    float* dy = c.alloc<float>((size_t)M * Cout);
    float* dg1 = c.up_pad(g1, Cout, 256); float* db1 = c.up_pad(b1, Cout, 256);
    float* dg2 = is_hc ? c.up_pad(g2, Cout, 256) : nullptr; float* db2 = is_hc ? c.up_pad(b2, Cout, 256) : nullptr;
    if (!c.ok) return OPH_ERR_DEVICE;
    launch_pad_rows(dx, Cin, dxp, kc, M, Cin, c.s);
    GemmArgs g{};
    g.X = dxp; g.ldx = kc; g.Wt = dw; g.ldw = size * kc; g.bias = dbias; g.H = dh; g.ldh = Nalloc; g.M = M; g.N = N; g.kc = kc;
    g.ntaps = size; g.mode = 0; g.T = T;
    for (int t = 0; t < size; ++t) g.off[t] = padding == 1 ? -(size - 1 - t) * rate : (t - (size - 1) / 2) * rate;
    launch_conv_gemm(g, c.s);
    EpiArgs e{};
    e.H = dh; e.ldh = Nalloc; e.M = M; e.C = Cout; e.mode = is_hc ? PRE_HC : PRE_CONV; e.act = act;
    e.g1 = dg1; e.b1 = db1; e.g2 = dg2; e.b2 = db2; e.Xres = dxp; e.ldres = kc; e.Y = dy; e.ldy = Cout; e.ypad = Cout;
    launch_epilogue(e, c.s);
    hipMemcpyAsync(y, dy, (size_t)M * Cout * 4, hipMemcpyDeviceToHost, c.s);
    return c.finish();
}

int oph_op_conv1d(int device, const float* x, int B, int T, int Cin, int Cout, int size, int rate, int padding,
                  const float* kernel, const float* bias, const float* gamma, const float* beta, int act, float* y) {
    return op_conv_common(device, x, B, T, Cin, Cout, size, rate, padding, kernel, bias, gamma, beta, nullptr, nullptr, act, false, y);
}
int oph_op_hc(int device, const float* x, int B, int T, int C, int size, int rate, int padding, const float* kernel,
              const float* bias, const float* gamma1, const float* beta1, const float* gamma2, const float* beta2, float* y) {
    return op_conv_common(device, x, B, T, C, C, size, rate, padding, kernel, bias, gamma1, beta1, gamma2, beta2, ACT_NONE, true, y);
}

int oph_op_conv1d_transpose(int device, const float* x, int B, int T, int Cin, int Cout, const float* kernel,
                            const float* bias, const float* gamma, const float* beta, float* y) {
    return oph_op_conv1d_transpose_prec(device, x, B, T, Cin, Cout, kernel, bias, gamma, beta, 0, y);
}
int oph_op_conv1d_transpose_prec(int device, const float* x, int B, int T, int Cin, int Cout, const float* kernel,
                                 const float* bias, const float* gamma, const float* beta, int precision, float* y) {
    if (precision < 0 || precision > 2) { g_op_error = "precision must be 0 (fp32 MFMA), 1 (split-bf16 x3) or 2 (split-fp16 x3)"; return OPH_ERR_INVALID; }
    OpCtx c(device);
    if (!c.ok) return OPH_ERR_DEVICE;
    if (Cout > 1280) { g_op_error = "channels out of range"; return OPH_ERR_UNSUPPORTED; }
    const int kc = round_up(Cin, 32), Nalloc = round_up(Cout, 128), M = B * T;
    std::vector<float> we((size_t)Nalloc * 2 * kc, 0.f), wo((size_t)Nalloc * kc, 0.f);
    for (int n = 0; n < Cout; ++n)
        for (int ci = 0; ci < Cin; ++ci) {
            we[(size_t)n * 2 * kc + ci] = kernel[((size_t)0 * Cout + n) * Cin + ci];
            we[(size_t)n * 2 * kc + kc + ci] = kernel[((size_t)2 * Cout + n) * Cin + ci];
            wo[(size_t)n * kc + ci] = kernel[((size_t)1 * Cout + n) * Cin + ci];
        }
    float* dx = c.up(x, (size_t)M * Cin);
    float* dxp = c.alloc<float>((size_t)M * kc);
    float* dwe = c.up(we.data(), we.size()); float* dwo = c.up(wo.data(), wo.size());
    float* dbias = c.up_pad(bias, Cout, Nalloc);
    float* dh = c.alloc<float>((size_t)2 * M * Nalloc);
    float* dy = c.alloc<float>((size_t)2 * M * Cout);
    float* dg = c.up_pad(gamma, Cout, 256); float* db = c.up_pad(beta, Cout, 256);
    if (!c.ok) return OPH_ERR_DEVICE;
    launch_pad_rows(dx, Cin, dxp, kc, M, Cin, c.s);
    GemmArgs g{};
    g.X = dxp; g.ldx = kc; g.bias = dbias; g.ldh = 2 * Nalloc; g.M = M; g.N = Cout; g.kc = kc; g.mode = 0; g.T = T;
    g.Wt = dwe; g.ldw = 2 * kc; g.ntaps = 2; g.off[0] = 0; g.off[1] = -1; g.H = dh;
    GemmArgs g2 = g;
    g2.Wt = dwo; g2.ldw = kc; g2.ntaps = 1; g2.off[0] = 0; g2.H = dh + Nalloc;
    if (precision == 0) {
        launch_conv_gemm(g, c.s);
        launch_conv_gemm(g2, c.s);
    } else {        // the SSRN path's launch for this layer: both phases in one, on the split 16-bit planes
        unsigned short* dweh = c.alloc<unsigned short>(we.size()); unsigned short* dwel = c.alloc<unsigned short>(we.size());
        unsigned short* dwoh = c.alloc<unsigned short>(wo.size()); unsigned short* dwol = c.alloc<unsigned short>(wo.size());
        if (!c.ok) return OPH_ERR_DEVICE;
        if (precision == 2) { launch_split_f16(dwe, dweh, dwel, we.size(), c.s); launch_split_f16(dwo, dwoh, dwol, wo.size(), c.s); }
        else { launch_split_bf16(dwe, dweh, dwel, we.size(), c.s); launch_split_bf16(dwo, dwoh, dwol, wo.size(), c.s); }
        g.Wh = dweh; g.Wl = dwel; g.f16 = precision == 2; g.nprod = 3;
        g2.Wh = dwoh; g2.Wl = dwol; g2.f16 = g.f16; g2.nprod = 3;
        launch_conv_gemm_pair(g, g2, precision, c.s);
    }
    EpiArgs e{};
    e.H = dh; e.ldh = Nalloc; e.M = 2 * M; e.C = Cout; e.mode = PRE_CONV; e.act = ACT_NONE; e.g1 = dg; e.b1 = db; e.Y = dy; e.ldy = Cout; e.ypad = Cout;
    launch_epilogue(e, c.s);
    hipMemcpyAsync(y, dy, (size_t)2 * M * Cout * 4, hipMemcpyDeviceToHost, c.s);
    return c.finish();
}

// Device-resident timing of modules.conv1d_transpose (SSRN D_4 / D_7, networks.py:483-486) for the roofline report:
// the same launches as oph_op_conv1d_transpose / the SSRN path (even-phase GEMM, odd-phase GEMM, LayerNorm rows), on
// seeded random device data, `iters` repetitions bracketed by HIP events after `warmup` untimed ones.
int oph_bench_conv1d_transpose(int device, int B, int T, int Cin, int Cout, int precision, int warmup, int iters,
                               double* avg_us, double* alg_bytes, double* alg_flops) {
    OpCtx c(device);
    if (!c.ok) return OPH_ERR_DEVICE;
    if (Cout > 1280 || B < 1 || T < 1 || iters < 1 || !avg_us) { g_op_error = "bad argument"; return OPH_ERR_INVALID; }
    const int kc = round_up(Cin, 32), Nalloc = round_up(Cout, 128), M = B * T;
    std::vector<float> we((size_t)Nalloc * 2 * kc, 0.f), wo((size_t)Nalloc * kc, 0.f), xh((size_t)M * kc, 0.f), bh((size_t)Nalloc, 0.f);
    uint32_t st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f - 0.5f; };
    const float ws = sqrtf(2.6f / (3.0f * Cin));
    for (int n = 0; n < Cout; ++n) {
        bh[n] = 0.02f * rnd();
        for (int ci = 0; ci < Cin; ++ci) { we[(size_t)n * 2 * kc + ci] = ws * rnd(); we[(size_t)n * 2 * kc + kc + ci] = ws * rnd(); wo[(size_t)n * kc + ci] = ws * rnd(); }
    }
    for (int m = 0; m < M; ++m) for (int ci = 0; ci < Cin; ++ci) xh[(size_t)m * kc + ci] = rnd();
    std::vector<float> gh((size_t)round_up(Cout, 256), 1.f), zh((size_t)round_up(Cout, 256), 0.f);
    float* dx = c.up(xh.data(), xh.size());
    float* dwe = c.up(we.data(), we.size()); float* dwo = c.up(wo.data(), wo.size());
    float* dbias = c.up(bh.data(), bh.size());
    float* dh = c.alloc<float>((size_t)2 * M * Nalloc);
    float* dy = c.alloc<float>((size_t)2 * M * Cout);
    float* dg = c.up(gh.data(), gh.size()); float* db = c.up(zh.data(), zh.size());
    // the weights' hi / lo bf16 planes, split once as at load time
    unsigned short* dweh = c.alloc<unsigned short>(we.size()); unsigned short* dwel = c.alloc<unsigned short>(we.size());
    unsigned short* dwoh = c.alloc<unsigned short>(wo.size()); unsigned short* dwol = c.alloc<unsigned short>(wo.size());
    if (!c.ok) return OPH_ERR_DEVICE;
    if (precision >= 2) { launch_split_f16(dwe, dweh, dwel, we.size(), c.s); launch_split_f16(dwo, dwoh, dwol, wo.size(), c.s); }
    else { launch_split_bf16(dwe, dweh, dwel, we.size(), c.s); launch_split_bf16(dwo, dwoh, dwol, wo.size(), c.s); }
    hipStreamSynchronize(c.s);
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { g_op_error = "event creation failed"; return OPH_ERR_DEVICE; }
    auto once = [&]() {
        GemmArgs g{};
        g.X = dx; g.ldx = kc; g.bias = dbias; g.ldh = 2 * Nalloc; g.M = M; g.N = Cout; g.kc = kc; g.mode = 0; g.T = T;
        g.Wt = dwe; g.Wh = dweh; g.Wl = dwel; g.f16 = precision >= 2; g.nprod = precision == 3 ? 2 : (precision == 4 ? 1 : 3); g.ldw = 2 * kc; g.ntaps = 2; g.off[0] = 0; g.off[1] = -1; g.H = dh;
        GemmArgs g2 = g;
        g2.Wt = dwo; g2.Wh = dwoh; g2.Wl = dwol; g2.ldw = kc; g2.ntaps = 1; g2.off[0] = 0; g2.H = dh + Nalloc;
        launch_conv_gemm_pair(g, g2, precision < 0 || precision > 4 ? 0 : std::min(precision, 2), c.s);
        EpiArgs e{};
        e.H = dh; e.ldh = Nalloc; e.M = 2 * M; e.C = Cout; e.mode = PRE_CONV; e.act = ACT_NONE; e.g1 = dg; e.b1 = db; e.Y = dy; e.ldy = Cout; e.ypad = Cout;
        launch_epilogue(e, c.s);
    };
    for (int i = 0; i < warmup; ++i) once();
    hipEventRecord(e0, c.s);
    for (int i = 0; i < iters; ++i) once();
    hipEventRecord(e1, c.s);
    float ms = 0.f;
    hipError_t er = hipEventSynchronize(e1);
    if (er == hipSuccess) er = hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    if (er != hipSuccess) { g_op_error = hipGetErrorString(er); return OPH_ERR_DEVICE; }
    *avg_us = (double)ms * 1e3 / iters;
    // per input row Cin*4 B in + 2*Cout*4 B out, + the 3*Cin*Cout weights once per call; 2*3*Cin*Cout flop per input row
    if (alg_bytes) *alg_bytes = ((double)M * Cin + 2.0 * M * Cout + 3.0 * Cin * Cout) * 4.0;
    if (alg_flops) *alg_flops = 2.0 * 3.0 * (double)M * Cin * Cout;
    return c.finish();
}

int oph_op_attention(int device, const float* Q, const float* K, const float* V, const int32_t* prev_max, int B, int T,
                     int N, int d, int win, float* R, float* alignments, int64_t* max_attentions) {
    OpCtx c(device);
    if (!c.ok) return OPH_ERR_DEVICE;
    if (d % 4 || d > 512 || win < 1 || win > 8) { g_op_error = "d/win out of range"; return OPH_ERR_UNSUPPORTED; }
    for (int b = 0; b < B; ++b) if (prev_max[b] < 0 || prev_max[b] >= N) { g_op_error = "prev_max out of range"; return OPH_ERR_INVALID; }
    float* dq = c.up(Q, (size_t)B * T * d); float* dk = c.up(K, (size_t)B * N * d); float* dv = c.up(V, (size_t)B * N * d);
    int* dp = c.up(prev_max, (size_t)B);
    float* dr = c.alloc<float>((size_t)B * T * 2 * d); float* da = c.alloc<float>((size_t)B * N * T);
    long long* dm = c.alloc<long long>((size_t)B * T);
    if (!c.ok) return OPH_ERR_DEVICE;
    AttnRowsArgs a{};
    a.mode = 1; a.Q = dq; a.ldq = d; a.K = dk; a.V = dv; a.ldkv = d; a.N = N; a.d = d; a.win = win; a.p = dp; a.B = B; a.Bpad = B;
    a.nrows = B * T; a.T = T; a.R = dr; a.ldr = 2 * d; a.align = da; a.amax = dm;
    launch_attn_rows(a, c.s);
    hipMemcpyAsync(R, dr, (size_t)B * T * 2 * d * 4, hipMemcpyDeviceToHost, c.s);
    hipMemcpyAsync(alignments, da, (size_t)B * N * T * 4, hipMemcpyDeviceToHost, c.s);
    hipMemcpyAsync(max_attentions, dm, (size_t)B * T * 8, hipMemcpyDeviceToHost, c.s);
    return c.finish();
}

}  // extern "C"
