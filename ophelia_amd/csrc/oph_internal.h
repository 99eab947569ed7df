// Internal declarations shared by the kernel and API translation units of
// libophelia_hip.so.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace oph {

constexpr int ROWT = 16;        // decode row tile = one 16x16x4 f32 MFMA tile of utterances
constexpr float LN_EPS = 1e-12f;            // tf.contrib.layers.layer_norm epsilon (modules.py:65)
constexpr float MASK_VALUE = -4294967296.0f; // -2**32+1 rounded to fp32 (networks.py:312)

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2 };
enum Pre { PRE_COPY = 0, PRE_CONV = 1, PRE_HC = 2 };

// ---- batched conv-as-GEMM (fp32 MFMA), H[m][n] = sum_tap X[src(m,tap)][:] . Wt[n][tap*kc + :] + bias[n]
struct GemmArgs {
    const float* X; int ldx;        // input activations, row stride (floats)
    const float* Wt; int ldw;       // packed weights [Nalloc][ntaps*kc], k contiguous
    const float* bias;              // [Nalloc]
    float* H; int ldh;              // raw conv output rows (bias added, pre-LayerNorm)
    int M;                          // output rows
    int N;                          // output columns actually needed (tiles cover round_up(N,BN))
    int kc;                         // per-tap K (padded Cin, multiple of 32)
    int ntaps;
    int mode;                       // 0 dense (b,t) rows, 1 table-mapped rows (decoder cone)
    int T; int off[3];              // dense: time length per utterance, per-tap time offsets
    int Bpad; int n_out; int j;     // table: rows = i*Bpad + b ; current decoder step j
    const int* tab;                 // table: [ntaps][n_out] source position index in X (rows idx*Bpad+b)
    const int* need;                // table: [ntaps][n_out] source valid iff j >= need
    const int* stop_after; int t;   // early-out when t > *stop_after (decode loop); stop_after may be null
    int ksplit; long long split_stride;  // split-K over K-steps: grid.y = ksplit, partial s written at H + s*split_stride (bias in split 0)
    // conv_gemm_bf16x3 only: Wt split once into hi = bf16(w), lo = bf16(w - hi), same [Nalloc][ntaps*kc] layout (2-byte elements)
    const void* Wh; const void* Wl; int f16;       // f16: the planes (and the activations' split) are fp16 terms instead of bf16
    int nprod;                      // measurement only: 0 / 3 = ah.bh + ah.bl + al.bh; 2 = ah.bh + al.bh (weights hi only); 1 = ah.bh
};
// ---- the same contraction on operands that arrive as fp16 hi / lo planes (oph_planegemm.hip)
constexpr int PLANE_GEMM_HALO = 16;             // activation rows loaded either side of a 128-row tile: the largest tap offset served
struct PlaneGemmArgs {
    const _Float16* Ah; const _Float16* Al;     // activation planes, K-blocked [kc / 32][M][32]
    // weight planes, K-blocked like the activations: [ntaps kc / 32][Nalloc][32], tap-major (launch_kblock_planes; a tile's 128
    // rows of one K block are 8 contiguous KB -- rows of a k-contiguous [Nalloc][ntaps kc] plane lie a power of two apart and
    // land on one L2 channel).  Transposed conv: the even phase (taps x[t], x[t-1]) ...
    const _Float16* Wh; const _Float16* Wl;
    const _Float16* Wh2; const _Float16* Wl2;   // ... and the odd phase's planes [kc / 32][Nalloc][32]
    const float* bias;
    float* H;                                   // raw rows [M][ldh] (transposed conv: ldh = 2 Nalloc, row 2t at H + m ldh, row 2t + 1 at + Nalloc)
    int M, N, kc, T;                            // rows, output channels, padded input channels (multiple of 32), rows per utterance
    int nalloc, ldh;                            // rows of the weight planes; raw row stride
    int ntaps, off[3];
    int convt;
    int waves;                                  // 0: chosen by the launcher; 4 / 8: forced (measurement)
    int dbg;                                    // measurement only: 1 = no MFMAs (operand stream alone), 2 = no operand stream (MFMAs on whatever the LDS holds)
    // ---- conv1d_transpose with its LayerNorm inside the launch (round 6; ln_gamma != null, N a multiple of 64): the column tiles of a
    // row tile exchange per-row (mean, M2) partials of their 64-channel groups, every workgroup normalises its own channels and
    // writes the rows as fp32 (Y, 2 M rows of ldy floats) and, if Yh is set, as the next layer's fp16 hi / lo planes [ldy / 32][2 M][32]
    const float* ln_gamma; const float* ln_beta;
    float* Y; int ldy;
    _Float16* Yh; _Float16* Yl;
    float* ln_stats;                            // exchange region: [row tile][N / 64 groups][256 = 128 rows x 2 phases] 16-byte granules {mean, M2, epoch, -}
    unsigned ln_epoch;                          // tag of this launch's granules (never reused on this region)
    int* ln_err;                                // set to 1 if a partner's granules never arrived (bounded wait: workgroups of the launch not co-resident)
};
size_t plane_gemm_ln_stats_bytes(int M, int N);            // size of the exchange region for a launch of M input rows
void launch_plane_gemm(const PlaneGemmArgs& a, hipStream_t s);
bool plane_gemm_ok(int ntaps, const int* off, int kc, bool convt);
void launch_kblock_planes(const void* src, void* dst, int rows, int ld, hipStream_t s);      // 2-byte plane [rows][ld] -> [ld / 32][rows][32]
void launch_rows_to_planes(const float* x, int ld, int M, int kc, void* ph, void* pl, hipStream_t s);

void launch_conv_gemm_pair(const GemmArgs& a0, const GemmArgs& a1, int prec, hipStream_t s);   // two contractions (same M, N, no split-K) in one launch
void launch_split_bf16(const float* w, void* hi, void* lo, size_t n, hipStream_t s);      // hi/lo planes of n floats
void launch_split_f16(const float* w, void* hi, void* lo, size_t n, hipStream_t s);

// ---- LayerNorm epilogues over raw conv rows (one wavefront per row)
struct EpiArgs {
    // (pointers and 64-bit fields first, then the 32-bit ones: 248 bytes; see HcFusedArgs)
    const float* H;                 // raw rows
    const float *g1, *b1, *g2, *b2; // gamma/beta (conv: g1,b1 ; hc: H1 -> g1,b1, H2 -> g2,b2)
    const float* Xres;              // hc residual rows
    const int* restab;              // if non-null: residual row = restab[m / Bpad]*Bpad + m % Bpad
    float* Y;                       // output rows; columns [Ctot, ypad) are zero-filled
    // optional speaker-embedding append (AudioDec 'audio_decoder_input', networks.py:381-387)
    const float* spk_table; const int* spk_ids;
    const int* stop_after;
    // learned channel contributions (modules.py:78-88): per-speaker channel gate sigmoid(lcc_embed[spk]) stored as a
    // table [nspeakers][C]; conv: y = gate * act(LN(h)) (a final squash sigmoid comes after the gate); hc: H2 *= gate
    const float* lcc; const int* lcc_ids;
    // optional completion signal (the launch that writes a cone level in dec_loop mode): the rows of positions coh0 and
    // coh1 -- the only ones the loop kernel reads as taps -- are stored write-through (8-byte sc1 stores, no fence); each
    // workgroup holding such rows waits for its stores, adds 1 to *done_count, and the one that makes it done_target
    // raises *done_sig to done_val
    // planes != 0 (the batched nets: no completion signal there): the two pointers are instead the fp16 hi / lo planes the rows
    // are ALSO written to, K-blocked [ypad / 32][M][32] -- plane_gemm's operand format for the next layer (oph_planegemm.hip)
    union { unsigned* done_sig; void* Yh; };
    union { unsigned* done_count; void* Yl; };
    long long* done_stamp;          // diagnostics: the raising lane stores the constant clock here (or null)
    long long split_stride;         // raw = sum of nsplit partial buffers (split-K GEMM)
    long long out_bs;
    int ldh, nsplit;
    int M; int C;                   // rows ; channels of the OUTPUT (hc: raw has 2C)
    int mode;                       // PRE_CONV (LN + act) or PRE_HC (2xLN + gate + highway)
    int act;
    int ldres, Bpad;
    int ldy; int ypad;
    int spk_dim; int spk_T;         // utterance of row m: spk_T>0 ? m / spk_T : m % Bpad
    int t;
    int nonorm;                     // hp.norm None: the "LayerNorm" is the identity (mean 0, rstd 1; gamma/beta = 1/0 buffers)
    // optional output row mapping (streamed SSRN chunks): the M rows are [B][out_T]; row (b, u) with keep_lo <= u < keep_hi is
    // stored at output row b * out_bs + out_t0 + u, the others are skipped.  out_T == 0: row m -> output row m
    int out_T; int keep_lo, keep_hi; int out_t0;
    int lcc_T;                      // utterance of row m: lcc_T > 0 ? m / lcc_T : m % Bpad
    unsigned done_val, done_target; int coh0, coh1;
    int planes;
};
static_assert(sizeof(EpiArgs) <= 256, "ln_rows' kernel arguments: four 64-byte lines");

// ---- fused M=16 decode layer (prologue = previous layer's LN/gate, then 16xK . KxN slice)
struct DecArgs {
    int pre;                        // PRE_COPY / PRE_CONV / PRE_HC : how x[t] is produced
    const float* src; int ldsrc;    // PRE_COPY: x rows ; else raw rows of the previous layer
    const float *g1, *b1, *g2, *b2; int act;
    const float* xres; int ldres;   // PRE_HC residual rows (previous layer's input at t)
    int cin;                        // channels produced by the prologue
    const float* cat_table; const int* cat_ids; int ccat;  // optional speaker embedding concat
    float* xstore; int ldstore;     // block x==0 stores x[t] rows here (this layer's input history), may be null
    int ntaps; int kc;              // kc = padded per-tap K (multiple of 16) >= cin + ccat
    const float* tap0; const float* tap1; int ldtap;        // older taps (x[t-2r], x[t-r]) rows or null = zeros
    const float* Wt; int ldw; const float* bias;
    float* H; int ldh;              // raw output rows [Bpad][ldh]
    int B;
    const int* stop_after; int t;
    int nonorm;                     // the PREVIOUS layer (prologue) has no LayerNorm
    const float* lcc; const int* lcc_ids;   // the PREVIOUS layer's LCC gate table [nspeakers][cin] (or null), speaker per row
};

struct AttnRowsArgs {
    int mode;                       // 0: decoder history rows (position-major), 1: batched op (b,t) rows
    const float* Q; int ldq;
    const float* K; const float* V; int ldkv; int N; int d; int win;
    const int* p;                   // per-utterance window start
    const int* ptab;                // non-null (mode 0): FixedAttention -- the query at time t of utterance b selects key
                                    // ptab[t*Bpad+b] with weight 1 (or nothing if < 0); networks.py:327-358
    const int* ends;                // non-null: hp.turn_off_monotonic_for_synthesis -- no window, the unmasked keys of
                                    // utterance b are [0, min(N, ends[b]+1))  (networks.py:307-309, synthesize.py:505-507)
    int B; int Bpad; int nrows;
    const int* off; int j;          // mode 0: row i*Bpad+b reads Qhist[(j-off[i])*Bpad+b]
    int T;                          // mode 1
    float* R; int ldr;
    float* align; long long* amax;  // mode 1 outputs (B,N,T) and (B,T)
    const int* stop_after; int t;
    const unsigned* wait_sig; unsigned wait_val; int* wait_err;   // optional: spin until *wait_sig >= wait_val before anything else
};

// ---- head of the AudioDec history cone in one launch (oph_kernels.hip: cone_head)
// AudioDec's first layer C_1 is a k=1 conv of R' = [ctx | Q] (networks.py:316-319, 376): it is linear, so
//   C_1(R')[t'] = sum_i prob_i(t') * (V[p+i] . Wc)  +  (Q[t'] . Wq + bias)
// with VW = V . Wc computed once per batch and QW[t'] = Q[t'] . Wq + bias once per position (each history position is
// new exactly once: at offset 1; the workgroups of that position compute it).  One wave per (position, utterance) row: attention window, the
// two cached terms, LayerNorm -- instead of attn_rows + a [1344 x 512 x 256] GEMM + ln_rows every step.
struct ConeHeadArgs {
    // (pointers first, then the 32-bit fields, and no blockDim in the kernel -- that would append 256 bytes of implicit arguments:
    //  252 bytes = four 64-byte lines of kernel arguments; see HcFusedArgs)
    const float* Q;                     // Qhist [max_T][Bpad][d]
    const float* KV;                    // K | V rows [B][N][2d]
    const float* VW;                    // [B][N][ldvw] = V . Wc
    float* QW;                          // [max_T][Bpad][d] cache of Q . Wq + bias
    const float* Wq;                    // [d][ldn] n-contiguous rows of the Q half of C_1's kernel
    const float* bias; const float* gamma; const float* beta;
    const int* p; const int* off;       // row i*Bpad+b <-> time j - off[i]
    float* Y;                           // output rows (layer input of the next cone stage)
    void* Yh; void* Yl;                 // optional: the same rows as fp16 hi / lo planes, K-blocked [d / 64][nrows][64] (hc_fused's operand format)
    const float* spk_table; const int* spk_ids;       // optional embedding appended after the d channels
    int* ctl;                           // the tile's control words: [1] stop_after, [2] error (set when a bounded wait gives up)
    const unsigned* wait_sig;
    unsigned* done_sig; unsigned* done_count;          // as EpiArgs: cone level 0 written
    long long* done_stamp;
    int d, N_keys, win, ldvw, ldn, nonorm;
    int B, Bpad, nrows, j;
    int npos; int i_new;                // positions (nrows = npos * Bpad); index of the newest one (smallest offset)
    int ldy, spk_dim, t;
    unsigned wait_val, done_val, done_target; int coh0, coh1;
    int rb;                             // rows per workgroup (= waves per workgroup), set by launch_cone_head
};
static_assert(sizeof(ConeHeadArgs) <= 256, "cone_head's kernel arguments: four 64-byte lines");
void launch_cone_head(const ConeHeadArgs& a, hipStream_t s);

// ---- hc_fused: a level of the AudioDec history cone as ONE launch (oph_hcfused.hip): split-fp16 x3 contraction with both operands
// as fp16 hi / lo planes through global_load_lds, then LayerNorm x 2 + gate + highway mix in the same kernel -- the 8 column tiles of
// a 64-row block exchange per-row (mean, M2) partials as granules.  C = 256 channels, 3 taps, Bpad = 16.
struct HcFusedArgs {
    // (pointers first, then the 32-bit fields: 248 bytes.  With the fields in reading order the struct was 264 bytes -- a fifth 64-byte
    //  line of kernel arguments for every wave's scalar loads -- and every launch measured ~1 us longer, profiles/r04_ab.sh)
    const void* Xh; const void* Xl;                     // level k-1 rows as fp16 planes, K-blocked: [256 / 64][in_rows = positions * Bpad][64]
    const float* Xres; const int* restab;               // ... and as fp32 (highway residual): row restab[ip] * Bpad + b
    const int* tab; const int* need;                    // [3][n_out] source position per tap (oldest first), valid iff j >= need
    const void* Wh; const void* Wl; const float* bias;  // the layer's kernel as planes [8 column tiles][12 K-steps][64 columns][64], a tile's
                                                        // columns = [32 H1 | the same 32 channels of H2]; bias in the same column order
    const float *g1, *b1, *g2, *b2;                     // LayerNorm parameters of H1 / H2 (channel order)
    float* Y; void* Yh; void* Yl;                       // level k rows: fp32 [M][256] and K-blocked planes [4][M][64]
    unsigned long long* stats;                          // exchange granules [row block][2][2][32][8][2]
    int* ctl; const float* zeros;                       // ctl: the tile's control words ([1] stop_after, [2] error)
    unsigned* done_sig; unsigned* done_count;           // as EpiArgs
    long long* done_stamp;
    long long* dbg;                                     // diagnostics: phase stamps of workgroup 0 [8], or null
    int in_rows, n_out, j;
    int Bpad, M;                                        // M = n_out * Bpad output rows
    unsigned epoch;                                     // tag of this launch's granules (never reused)
    int t;
    unsigned done_val, done_target; int coh0, coh1;
};
static_assert(sizeof(HcFusedArgs) <= 256, "hc_fused's kernel arguments: four 64-byte lines");
void launch_hc_fused(const HcFusedArgs& a, hipStream_t s);
size_t hc_fused_lds_bytes();
int hc_fused_grid(int M);
int hc_fused_active(int M);
int hc_fused_holders(int M, int Bpad, int pos, int pos2);
int hc_fused_blocks_per_cu(int M);

// ---- cone_fc16: a SMALL cone level in one launch (oph_kernels.hip).  Highway layer k of the AudioDec cone evaluated at
// its n_out output positions, with the LayerNorm / gate / highway mix of layer k-1 (the launch ln_rows would be) as the
// prologue of the workgroups that need those rows: per output position the three gathered input positions are
// normalised from layer k-1's raw rows, staged, and contracted (16 x 16 slices, K split over 16 waves, as dec_layer16).
// Column slice 0 stores the x rows it produced (level k of the cone); `extra` positions are rows only the loop kernel
// reads (prologue + store, no contraction).
constexpr int CONE_FC_MAXOUT = 16, CONE_FC_MAXEXTRA = 4;
struct ConeFcArgs {
    const float* rawp; int ldrawp; int nsplit; long long split_stride;     // raw rows of layer k-1 [n_k * Bpad][2C] (sum of nsplit partials)
    const float *g1, *b1, *g2, *b2; int nonorm; int C;                     // LayerNorm parameters of layer k-1
    // the small index tables travel in the kernel arguments (scalar loads, no dependent global round trips)
    const float* xres; int ldres;                                          // level k-1 rows
    short tab[3][CONE_FC_MAXOUT]; short need[3][CONE_FC_MAXOUT];           // level-k position per tap (oldest first), valid iff j >= need
    short res[3][CONE_FC_MAXOUT];                                          // level k-1 position of that level-k position (residual row)
    short extra[CONE_FC_MAXEXTRA]; short extra_res[CONE_FC_MAXEXTRA];      // level-k positions that are stored only (+ their residual positions)
    int n_out; int n_extra; int j;
    float* xstore; int ldx;                                                // level k rows [n_k * Bpad][ldx] (ldx = kc of layer k)
    const float* Wt; int ldw; const float* bias; int kc; int N;            // layer k: [Nalloc][3 kc]
    float* H; int ldh;                                                     // raw rows of layer k [n_out * Bpad][ldh]
    int Bpad;
    const int* stop_after; int t;
    unsigned* done_sig; unsigned done_val; unsigned* done_count; unsigned done_target; int coh0, coh1;   // level k completion (see EpiArgs)
    long long* done_stamp;
};
void launch_cone_fc16(const ConeFcArgs& a, hipStream_t s);

// ---- row-parallel fused chain of k=1 layers (LayerNorm is row-local, so a run of k=1 convs needs no
// cross-workgroup exchange): one workgroup per utterance streams each layer's full [K][N] weights.
struct RowLayer {
    const float* W; int ldn;        // weights [kc][ldn], n contiguous (second packed copy of the k=1 layers)
    const float* bias; const float* g; const float* b;
    int kc; int N; int act; int ccat;   // ccat: speaker-embedding channels appended to this layer's INPUT
    const float* lcc;               // LCC gate table [nspeakers][N] of this layer or null
};
enum RowPro { ROW_COPY = 0, ROW_HC = 1, ROW_ATTN = 2 };
struct RowChainArgs {
    int pro;                        // how the chain input is produced
    const float* src; int ldsrc; int cin;       // ROW_COPY: x rows ; else raw rows of the previous highway layer
    const float *g1, *b1, *g2, *b2; const float* xres; int ldres;
    // ROW_ATTN: attention at row t + bookkeeping (networks.py:286-325, synthesize.py:204-228)
    const float* KV; int N_keys; int d; int win; int max_T;
    const int* pcur; int* pnext; const int* ends; int* t_ends; int* n_ended; int* stop_flag; int stop_mode;
    float* Qhist; float* align; int Bpad;
    int nlayers; RowLayer L[4];
    const float* cat_table; const int* cat_ids;
    float* xout; int ldout;         // final activation rows (zero padded to ldout) or null
    int emit; float* Yout; int ldy; float* Ytm; int ldtm;    // emit: final = mel frame t -> Yout[b][t], Ytm[t+1][b]
    int B; const int* stop_after; int t;
    int nonorm;                     // no LayerNorm anywhere in this chain (hp.norm None)
    int nomono;                     // ROW_ATTN without the monotonic window: keys [0, min(N_keys, ends[b]+1))
    const int* ptab;                // ROW_ATTN with FixedAttention: key index per (t, b), time-major [max_T][Bpad], -1 = none
    const float* lcc_pro;           // LCC gate table of the highway layer whose gate the prologue applies (or null)
    int has_lcc;                    // any LCC table in this chain (selects the kernel instantiation); ids = cat_ids
};

// ---- persistent run of decoder layers in ONE launch (oph_decrun.hip): grid = (N/16 column slices, Bpad/R row groups),
// R waves.  Every workgroup runs each layer's prologue (the producing layer's LayerNorm / gate / attention for its R
// rows, one row per wave) and an R x 16 output slice; between layers the raw outputs travel through 8-byte
// {epoch, value} granules (agent-scope relaxed atomics: the data is its own flag), so a layer boundary costs one
// L2/fabric hand-off instead of a kernel boundary.
enum RunPre { RUN_COPY = 0, RUN_CONV = 1, RUN_HC = 2, RUN_ATTN = 3 };
constexpr int RUN_MAX_LAYERS = 16;
constexpr int RUN_GCOLS = 512;          // granule columns per row (raw width of a highway layer, 2d <= 512)
struct RunLayer {
    int pre;                            // how this layer's input row x[t] is produced (RunPre)
    int act;                            // RUN_CONV: activation of the producing conv layer
    int cin;                            // channels the prologue produces (<= 256); RUN_ATTN yields [ctx | q] = 2*cin
    int nonorm;                         // the producing layer has no LayerNorm (hp.norm None)
    const float *g1, *b1, *g2, *b2;     // the producing layer's gamma / beta (hc: H1 -> g1,b1 ; H2 -> g2,b2)
    const float* src; int ldsrc;        // plain rows: RUN_COPY input, or the raw rows an EARLIER launch left; null = granules
                                        // of the previous layer of this run
    const float* cat_table; int ccat;   // speaker embedding appended to x (ids: RunArgs::spk_ids)
    int ntaps, kc, N;                   // this layer's conv; N == 0: none (the prologue result is the mel frame)
    const float* tap0; const float* tap1; int ldtap;   // older taps x[t-2r], x[t-r]: rows [Bpad][ldtap], null = zeros
    const float* Wt; int ldw; const float* bias;
    float* xstore; int ldstore;         // x[t] rows stored by column slice 0 (tap history of later steps), or null
    float* out; int ldout;              // last layer of the launch: plain raw rows [Bpad][ldout]; else null (granules)
};
struct RunArgs {
    int nlayers; int B; int Bpad; int t;
    const int* stop_after;
    const int* spk_ids;
    unsigned long long* gbuf;           // granules [RUN_MAX_LAYERS][Bpad][RUN_GCOLS]
    unsigned epoch0;                    // tag of layer l's output = epoch0 + l + 1 (never reused: the host advances it per launch)
    int* err;                           // set when a hand-off timed out (the launch still terminates)
    long long* stamps;                  // diagnostics (OPH_RUN_STAMPS): [slice][layer][8] 100 MHz clock stamps of wave 0, or null
    // RUN_ATTN layer: attention at row t + bookkeeping (networks.py:286-325, synthesize.py:204-228)
    const float* KV; int N_keys; int win; int max_T;
    const int* pcur; int* pnext; const int* ends; int* t_ends; int* n_ended; int* stop_flag; int stop_mode;
    float* Qhist; float* align;
    // mel frame t (the N == 0 pseudo-layer)
    float* Yout; int ldy; float* Ytm; int ldtm;
    RunLayer L[RUN_MAX_LAYERS];
};
void launch_dec_run(const RunArgs& a, int col_slices, int rows_per_group, int kmax, hipStream_t s);   // rows_per_group 4 or 8

// ---- the WHOLE decode loop in one launch (oph_decrun.hip, dec_loop): the same workgroups run every layer of every
// step; the layer descriptions are static for a decode and live in device memory.  Layer 0's prologue consumes the
// LAST layer of the previous step (LayerNorm + squash sigmoid = the mel frame, synthesize.py:204-209), so a step
// boundary is one more granule hand-off.  The AudioDec history cone of step t runs on a side stream as before; the two
// dependencies are device words: sig[0] (attention of step t done, written here, awaited by hipStreamWaitValue32 on the
// side stream) and sig[16] (cone of step t done, written by hipStreamWriteValue32 there, polled here).
constexpr int LOOP_MAX_LAYERS = 32;
struct LoopLayer {
    int pre; int act; int cin; int nonorm;          // as RunLayer (layer 0: RUN_CONV + sigmoid of the previous step's last layer)
    const float *g1, *b1, *g2, *b2;
    const float* cat_table; int ccat;
    int ntaps, kc, N;
    const float* Wt; int ldw; const float* bias;
    int tapkind;                        // 0 none; 1 history: hist[t - off][Bpad][kc]; 2 cone: cone[t & 1][idx][Bpad][kc]
    int off0, off1, idx0, idx1;         // time offsets (off0 > off1 > 0) of the two older taps, their row-block indices in a cone buffer
    int level1;                         // tapkind 2: 1 + cone level the taps read (its own completion word), else 0
    float* hist;                        // tapkind 1: this layer's input history, x[t] is stored here by column slice 0
    const float* cone0; const float* cone1;
};
// What dec_loop reads: the same layer as 20 packed dwords at a 32-dword stride, fetched one layer ahead with scalar
// loads that are all issued together (a field-by-field walk of LoopLayer cost ~10 dependent scalar-cache round trips
// per layer: profiles/r02 stamps, ~2 us of a 6 us layer).
//   w0-1 Wt   w2-3 bias   w4-5 lnp (g1 | b1 | g2 | b2, each `ls` floats)   w6-7 cat_table   w8-9 hist   w10-11 cone0   w12-13 cone1
//   w14 pre | act << 4 | nonorm << 8 | ntaps << 12 | tapkind << 16 | (pre of the NEXT layer, cyclic) << 20
//   w15 cin | kc << 16     w16 N | ldw << 16     w17 ccat | ls << 16     w18 off0 | off1 << 16     w19 idx0 | idx1 << 16
constexpr int LOOP_DESC_WORDS = 20, LOOP_DESC_STRIDE = 32;
//   w14 also: (1 + cone level of this layer's taps) << 24 | (1 + cone level of the NEXT layer's taps) << 28   (0 = none)
// Cross-stream words (u32, one per 64-byte line): sig[0] attention of step t done; sig[16] whole cone of step t done (per-layer
// modes); sig[LOOP_SIG_LEVEL0 + 16 k] level k of the cone of step t written (dec_loop: the layer whose taps read level k
// waits for that word only -- the cone's later levels are still being computed while the chain's first tap layers run).
constexpr int LOOP_SIG_LEVEL0 = 32, LOOP_SIG_WORDS = 256, LOOP_MAX_LEVELS = 8;
struct LoopArgs {
    int nlayers; int B; int Bpad; int t_begin, t_end; int stop_mode;      // steps [t_begin, t_end)
    int attn_layer;                     // index of the RUN_ATTN layer
    const unsigned* L;                  // device memory, [nlayers][LOOP_DESC_STRIDE] packed descriptors
    float* QW; int attn_slices;         // if non-null: the attention layer also emits QW[t] = Q[t] . Wq + bias (the Q half of its own contraction,
                                        // kept in a second accumulator) for cone_head's cache; attn_slices = its column slices (all of them then
                                        // arrive before the cone of step t+1 is released)
    int* ctl;                           // [0] n_ended  [1] stop_after  [2] error  [3] attention arrivals
    const int* spk_ids;
    unsigned long long* gbuf;           // granules [LOOP_MAX_LAYERS][Bpad][RUN_GCOLS]
    unsigned epoch0;                    // tag of (step t, layer l) = epoch0 + t * LOOP_MAX_LAYERS + l + 1
    float* vbuf;                        // dec_chain's tag-free hand-off slots [2 step parities][LOOP_MAX_LAYERS][Bpad][RUN_GCOLS], every word
                                        // 0xFFFFFFFF when the launch starts (oph_decchain.hip: CH_SENT)
    long long* stamps; int stamp_t;
    const float* KV; int N_keys; int win; int max_T;
    int* p;                             // prev_max double buffer [2][Bpad] (read by the cone kernels)
    const int* ends; int* t_ends;
    float* Qhist; float* align; float* Yout; int ldy; float* Ytm;
    unsigned* sig; unsigned sig_base;
    volatile int* host_progress;        // pinned host words: [0] last step whose attention is done, [1] stop step (or INT_MAX)
    long long* sigdbg;                  // diagnostics: [max_T][8] clock stamps of the two signals, or null
    long long* clk;                     // device-side witness of the launch's duration (or null): [0] = min over workgroups of the 100 MHz
                                        // constant clock at entry, [1] = max at exit (the host sets them to ~0 / 0 before the launch)
    int dbg;                            // ablation switches for timing experiments (OPH_LOOP_DBG; results are wrong when set):
                                        // 1 no weight loads, 2 single-pass sweeps (no waiting), 4 no tap loads, 8 no prologue math,
                                        // 16 idle column slices do not sit layers out (results stay right)
};
void launch_dec_loop(const LoopArgs& a, int col_slices, int rows_per_group, int kmax, hipStream_t s);   // rows_per_group 4 or 8
int dec_loop_blocks_per_cu(int rows_per_group, int kmax);
// dec_chain (oph_decchain.hip): the same launch for the standard geometry (d = 256, 8 rows per workgroup, LayerNorm everywhere,
// attention window <= 4, QW emitted by the attention layer), specialised per layer kind; same LoopArgs, same descriptors
void launch_dec_chain(const LoopArgs& a, int col_slices, hipStream_t s);
int dec_chain_blocks_per_cu();

// weight repacking on the device (oph_pack.hip)
void launch_pack_conv(const float* k, float* Wt, int size, int cin, int cout, int kc, int Nalloc, hipStream_t s);
void launch_pack_convT(const float* kt, float* We, float* Wo, int cin, int cout, int kc, int Nalloc, hipStream_t s);
void launch_pack_wkn(const float* k, float* Wkn, int cin, int N, int kc, int ldn, hipStream_t s);
void launch_pack_wtc(const float* k, float* Wc, int d, int kc_c, int ldvw, hipStream_t s);
void launch_pack_hcf(const float* kr, const float* bs, float* wp, float* bp, hipStream_t s);
void launch_pack_loop(const float* Wt, int ldw, int rows_have, int nch, int slices, int R, int PF, float* dst, hipStream_t s);
void launch_pad_copy(const float* src, float* dst, size_t n, size_t npad, int mode, int row0, hipStream_t s);
void launch_maxabs(const float* x, size_t n, unsigned* out, hipStream_t s);

// launchers (oph_kernels.hip)
void launch_row_chain(const RowChainArgs& a, hipStream_t s);
void launch_conv_gemm(const GemmArgs& a, hipStream_t s);
void launch_conv_gemm_bf16x3(const GemmArgs& a, hipStream_t s);   // split-bf16 contraction (dense rows only)
int  conv_gemm_tile_m(int M, int N);          // tile size chosen for a problem (64 or 128)
void launch_epilogue(const EpiArgs& a, hipStream_t s);
void launch_dec_layer(const DecArgs& a, int Npad16, hipStream_t s);
void launch_attn_rows(const AttnRowsArgs& a, hipStream_t s);
void launch_embed(const int* ids, long long n, const float* table, int units, float* out, int ldo, hipStream_t s);
void launch_pad_rows(const float* src, int lds_, float* dst, int ldd, long long rows, int C, hipStream_t s);
void launch_copy_rows_strided(const float* src, long long src_bs, int ld, float* dst, int B, int T, int C, hipStream_t s);   // dst[b*T + t][:C] = src[b*src_bs + t*ld ..]
void launch_fill_int(int* p, int v, int n, hipStream_t s);
struct ResetTileArgs {
    float* buf[3]; size_t n4[3];        // three buffers to zero, lengths in 16-byte units (Yout, Ytm, alignments)
    int* p; int n_p;                    // prev_max ping-pong
    int* tends; int n_tends; int max_T; // t_ends = max_T ("not ended")
    int* ctl;                           // {0, INT_MAX, 0, 0}: n_ended, stop step, error, attention arrivals
};
void launch_reset_tile(const ResetTileArgs& a, hipStream_t s);
void launch_spk_append_rows(float* out, int ldo, long long rows, int T, int col0, const float* table, const int* ids, int dim, hipStream_t s);

}  // namespace oph
