/*
 * ophelia_hip.h -- C ABI of libophelia_hip.so: MI355X (gfx950) implementation of
 * Ophelia's Text2Mel + SSRN synthesis hot path.
 *
 * The reference (CSTR-Edinburgh/ophelia) has no FFI: its de-facto boundary for this
 * path is the Python API in synthesize.py around an opaque (graph, session) pair.
 * Every entry point below names the reference interface it replaces (file:line in
 * the reference tree).  INTEGRATION.md shows the ctypes binding a reference
 * maintainer would add.
 *
 * Conventions
 *   - plain C types only; all tensors float32, channels-last (B, T, C), C-contiguous,
 *     exactly the layouts of the reference placeholders (architectures.py:69-81).
 *   - caller allocates every host buffer; the library owns all device memory behind
 *     the opaque handle.  One handle = one GPU = one host thread (the reference is
 *     single-threaded around one tf.Session).
 *   - every call returns 0 on success or a negative oph_status; oph_last_error()
 *     gives the text.  No exceptions, no callbacks.  There is NO CPU fallback: if
 *     no gfx950 device is usable, oph_create fails.
 */
#ifndef OPHELIA_HIP_H
#define OPHELIA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OPH_ABI_VERSION 1

typedef enum oph_status {
    OPH_OK = 0,
    OPH_ERR_INVALID = -1,      /* bad argument / shape mismatch          */
    OPH_ERR_STATE = -2,        /* call out of order (e.g. weights missing) */
    OPH_ERR_DEVICE = -3,       /* HIP runtime error                        */
    OPH_ERR_UNSUPPORTED = -4   /* config outside the supported hot path    */
} oph_status;

/* Model dimensions = the hyper-parameters the reference graph reads from `hp`
 * (config/ *.cfg via configuration.py:60-66). */
typedef struct oph_dims {
    int32_t vocab;                  /* len(hp.vocab)            networks.py:136  */
    int32_t e;                      /* hp.e   embedding          lj_tutorial.cfg:71 */
    int32_t d;                      /* hp.d   Text2Mel hidden    :72 */
    int32_t c;                      /* hp.c   SSRN hidden        :73 */
    int32_t n_mels;                 /* hp.n_mels                 :61 */
    int32_t full_dim;               /* hp.full_dim = n_fft/2+1   :60 */
    int32_t r;                      /* hp.r   reduction factor (4 or 8) :69 */
    int32_t max_N;                  /* hp.max_N                  :41 */
    int32_t max_T;                  /* hp.max_T                  :42 */
    int32_t attention_win_size;     /* hp.attention_win_size     :74 */
    int32_t nspeakers;              /* hp.nspeakers (0 if single speaker) vctk_01.cfg:62 */
    int32_t speaker_embedding_size; /* hp.speaker_embedding_size vctk_01.cfg:63 */
    int32_t flags;                  /* OPH_FLAG_* */
} oph_dims;

#define OPH_FLAG_SPK_AUDIO_DECODER_INPUT 1  /* 'audio_decoder_input' in hp.multispeaker (networks.py:381-389) */
#define OPH_FLAG_NORM_NONE 2               /* hp.norm is None (modules.py:62-74): Text2Mel has no LayerNorm variables.  SSRN is
                                              unaffected: synthesize() builds SSRNGraph under hp.norm = 'layer'
                                              (synthesize.py:513-534), its gamma / beta variables are always expected */
#define OPH_FLAG_NO_MONOTONIC 4             /* hp.turn_off_monotonic_for_synthesis (networks.py:304-309): no attention window;
                                              keys n >= text_length+1 are masked (hp.text_lengths, synthesize.py:505-507) */
#define OPH_FLAG_SPK_TEXT_ENCODER_INPUT 8  /* 'text_encoder_input' in hp.multispeaker (networks.py:138-144)        */
#define OPH_FLAG_SPK_TEXT_ENCODER_TOWARDS_END 16 /* 'text_encoder_towards_end' (networks.py:184-199)               */
#define OPH_FLAG_SPK_AUDIO_ENCODER_INPUT 64 /* 'audio_encoder_input' in hp.multispeaker (networks.py:237-245)         */
#define OPH_FLAG_NO_CONCAT_QUERY 128       /* hp.concatenate_query False (networks.py:317-321): AudioDec reads the attention context alone;
                                              Text2Mel/AudioDec/C_1/conv1d/kernel is (1, d, d) instead of (1, 2d, d)                  */
#define OPH_FLAG_NO_SQUASH_T2M 256         /* hp.squash_output_t2m False (networks.py:430-433): Y = Y_logits, no sigmoid              */
#define OPH_FLAG_NO_SQUASH_SSRN 512        /* hp.squash_output_ssrn False (networks.py:533-536): Z = Z_logits                         */
#define OPH_FLAG_SPK_SSRN_INPUT 1024       /* 'ssrn_input' in hp.multispeaker (networks.py:457-465): SSRN/embed_2 + SSRN/C_3 behind C_1.
                                              The speaker codes reach SSRN through oph_ssrn_speakers, or are those staged with the text
                                              when SSRN runs on resident frames; plain oph_ssrn(Y != NULL) is an error then -- as the
                                              reference's sess.run(g.Z, {g.mels: Y}) is without g.speakers (synthesize.py:257)        */
#define OPH_FLAG_LCC 32                    /* 'learn_channel_contributions' (modules.py:78-88): per-speaker sigmoid
                                              channel gates on every Text2Mel layer that the reference passes lcc= to */

/* stop_mode of the decode loop */
#define OPH_STOP_REFERENCE 0   /* synthesize.py:218-228: break after the step at which all utterances ended */
#define OPH_STOP_NEVER     1   /* always run max_T steps (fixed-length timed configuration)                 */

typedef struct oph_handle oph_handle;

/* ---- lifetime -------------------------------------------------------------
 * replaces: Text2MelGraph(hp, mode="synthesize") + SSRNGraph(hp, mode="synthesize")
 *           + tf.Session()                    (synthesize.py:511,525,536)        */
int oph_abi_version(void);
int oph_create(const oph_dims* dims, int device, oph_handle** out);
/* oph_create with launch-path / arithmetic options: "NAME=value NAME ..." (NULL or "" = oph_create).  The library reads NO
 * environment variable for these (only OPH_TRACE, diagnostics on stderr): what a handle computes and how fast cannot depend on an
 * inherited environment.  Every option selects a launch path the library also takes by itself (geometries outside the default
 * kernels', the recovery ladder of oph_get_counters[9]) or an arithmetic flavour of oph_set_precision; results stay within
 * 1e-4 of the default's with identical attention traces (tests/test_gpu_decode_modes.py).  Names:
 *   DECODE=loop|runs|layers  RUN_ROWS=4|8  NO_CHAIN  NO_CONE_HEAD  NO_FUSED_CONE  NO_LOOP_QW  CONE_FC_ROWS=n  CONE_FC_INSPLIT=n
 *   CONE_KSPLIT=a,b  LOOP_LOOKAHEAD=n  STREAM_VALUE=1  NO_STREAM_SSRN  SSRN_CHUNK=n  NO_PREENCODE  NO_PLANE_GEMM  PG_WAVES=4|8
 *   CU_SPLIT=chain,cone  NO_CU_MASK  SSRN_PREC / CONE_PREC / TEXTENC_PREC = 0|1|2 (oph_set_precision's codes)  RUN_STAMPS
 * An unknown name is OPH_ERR_INVALID (oph_last_error(NULL) names it).  The ablation switches of the measurement scripts under
 * profiles/ (SKIP_CONE, LOOP_ALONE, LOOP_DBG: wrong or unused results) exist only in -DOPH_ABLATE builds. */
int oph_create_opts(const oph_dims* dims, int device, const char* options, oph_handle** out);
int oph_destroy(oph_handle* h);
const char* oph_last_error(const oph_handle* h);   /* h may be NULL: last create error */

/* ---- weights ----------------------------------------------------------------
 * replaces: tf.train.Saver(var_list=TRAINABLE_VARIABLES in scope).restore
 *           (synthesize.py:302-330).  Variables are addressed by the TF variable
 *           names the reference creates, e.g. "Text2Mel/TextEnc/HC_4/conv1d/kernel"
 *           (3,512,1024), "SSRN/D_4/conv2d_transpose/kernel" (1,3,Cout,Cin).      */
int oph_num_weights(const oph_handle* h);
int oph_weight_info(const oph_handle* h, int index, char* name, int name_cap,
                    int64_t* shape /*[4]*/, int* rank);
int oph_set_weight(oph_handle* h, const char* tf_var_name, const float* data,
                   const int64_t* shape, int rank);
/* oph_set_weights_device: ALL variables at once from a buffer already on this handle's GPU -- float32, back to back in
 * oph_weight_info order, n_floats in total -- instead of one oph_set_weight call per variable.  replaces: the same restore, when
 * the weights arrive over the wire: in a multi-GPU run rank 0 loads the checkpoint and broadcasts one flat tensor (RCCL over
 * xGMI); every rank hands its receive buffer over here, oph_finalize_weights repacks it with device kernels and
 * nothing passes through the host.  The buffer must stay valid until oph_finalize_weights returns. */
int oph_set_weights_device(oph_handle* h, const float* d_flat, int64_t n_floats);
int oph_finalize_weights(oph_handle* h);   /* repack to kernel layout (device kernels) */

/* ---- the three session calls of the hot path (host buffers in / out) ----------
 * oph_encode_text   replaces encode_text()          synthesize.py:232-240
 *     L (B,max_N) int32, spk (B) int32 or NULL  ->  K,V (B,max_N,d)
 * oph_text2mel      replaces synth_codedtext2mel()  synthesize.py:150-230
 *     K,V as returned by oph_encode_text (or any (B,max_N,d) arrays), ends (B) int32
 *     = get_text_lengths(L) (synthesize.py:242-247), spk (B) int32 or NULL
 *     -> Y (B,max_T,n_mels), t_ends (B) int32, alignments (B,max_N,max_T),
 *        *steps_run = number of decoder steps executed.  Frames after the break
 *        step stay 0, as in the reference.
 * oph_ssrn          replaces one sess.run(g.Z) of synth_mel2mag()  synthesize.py:250-260
 *     Y (B,T,n_mels) -> Z (B, r*T, full_dim)      (chunking stays in the caller)
 *
 * Residency between the three calls.  The reference hands K,V and Y back to Python and feeds them in again
 * (synthesize.py:172, 256); the results of each call also stay in HBM, and a NULL input means "what the previous call on
 * this handle left there":
 *     oph_text2mel(K = NULL, V = NULL, ...)  decodes from the K,V of the last oph_encode_text (same B);
 *     oph_ssrn(Y = NULL, B, T = max_T, Z)    continues from the mel frames of the last oph_text2mel /
 *                                            oph_text2mel_durations (same B).  While that decode ran, SSRN was already
 *                                            evaluated over the frames that were final (oph_set_streaming), so this call
 *                                            computes the tail and copies Z out.
 * Non-NULL inputs are uploaded and used as they are; results are identical either way.
 * Batches larger than 16 utterances are decoded in tiles of 16, the tiles that stop early resumed to the batch's stop step
 * (the reference's break couples the whole batch, synthesize.py:225-228).                                           */
int oph_encode_text(oph_handle* h, const int32_t* L, const int32_t* spk, int B,
                    float* K, float* V);
int oph_text2mel(oph_handle* h, const float* K, const float* V, const int32_t* ends,
                 const int32_t* spk, int B, int stop_mode,
                 float* Y, int32_t* t_ends, float* alignments, int32_t* steps_run);
int oph_ssrn(oph_handle* h, const float* Y, int B, int T, float* Z);
/* oph_ssrn_logits   the same with the fetch surface's second tensor: Z = g.Z and Z_logits = g.Z_logits
 *     (networks.py:527-534: the last conv1d's LayerNorm rows before squash_output_ssrn's sigmoid), both (B, r*T, full_dim). */
int oph_ssrn_logits(oph_handle* h, const float* Y, int B, int T, float* Z, float* Z_logits);
/* oph_ssrn_speakers  sess.run([g.Z, g.Z_logits], {g.mels: Y, g.speakers: spk}) on the SSRN graph of a configuration with
 *     'ssrn_input' in hp.multispeaker (architectures.py:142, networks.py:457-465): spk (B) speaker codes; Z_logits may be NULL. */
int oph_ssrn_speakers(oph_handle* h, const float* Y, const int32_t* spk, int B, int T, float* Z, float* Z_logits);
/* Speculative SSRN during oph_text2mel (default on): chunks of mel frames go through SSRN on their own CU partition as
 * soon as the decoder has produced them (SSRN's receptive field is +-9 mel frames), for oph_ssrn(Y = NULL) to pick up.
 * on: 0 off, 1 on, n >= 2 on with n mel frames per chunk (default 40). */
int oph_set_streaming(oph_handle* h, int on);
/* Host buffer (B, r*max_T, full_dim) the speculative SSRN of the NEXT oph_text2mel copies its rows to while the decoder is
 * still running (pinned memory from oph_host_alloc makes the copies asynchronous); oph_ssrn(Y = NULL, ..., Z = that pointer)
 * then only computes and copies the tail.  NULL clears it. */
int oph_set_mag_destination(oph_handle* h, float* Z);
/* What the pipeline did since oph_create, out[0..n): [0] TextEnc evaluations, [1] runs whose K,V had been pre-encoded under
 * the previous decode, [2] SSRN chunks launched while a decode was running, [3] whole-decode launches, [4] fall-backs from the
 * whole-decode launch to two launches per step, [5] tiles resumed to their batch's stop step, [6] 0 (reserved),
 * [7] fp16 range guard: bit 0 SSRN, bit 1 cone, bit 2 TextEnc -- a weight of that net exceeds fp16's range (|w| > 6e4), so its
 *     split-fp16 contractions are pinned to the fp32-operand MFMA,
 * [8] 1 if this handle runs on the device's CU-masked streams (the three partitions chain | cone | SSRN are created once per process
 *     and device and shared by the handles of that device, whose calls take turns under the device's lock), 0 if it runs on ordinary
 *     streams (no whole-decode launch: another partition than the process's first was asked for, or masking is unavailable),
 * [9] decodes in which an in-kernel wait timed out (workgroups of a launch not co-resident) and the affected steps were redone on
 *     the per-step launch path,
 * [10] decodes this handle will still run on the reduced launch paths after such a recovery (0 = the default launches are armed):
 *     results on the reduced paths are within the cross-flavour bar (1e-4, identical attention traces), not bit-equal to the default's. */
int oph_get_counters(oph_handle* h, int64_t* out, int n);
/* oph_text2mel_graph  replaces ONE sess.run([g.Y, g.max_attentions, g.alignments], feed) of the reference's loop
 *     (synthesize.py:172,181-183) and serves as the fetch surface for the graph tensors of architectures.py:188-239:
 *     K,V (B,max_N,d), mels (B,max_T,n_mels) = the frames generated so far, prev_max (B) = the fed
 *     prev_max_attentions, ends (B) = text lengths (only read with OPH_FLAG_NO_MONOTONIC, else may be NULL),
 *     spk (B) or NULL  ->  any of (NULL = not wanted)  Q (B,max_T,d), R (B,max_T,2d), Y_logits / Y (B,max_T,n_mels),
 *     alignments (B,max_N,max_T), max_attentions (B,max_T) int32.  All max_T positions are evaluated under the ONE
 *     mask of prev_max, as the reference graph does (networks.py:311): O(max_T) per call, for validation and
 *     feed/fetch-style callers -- the decode loop proper is oph_text2mel.  External durations are not offered here. */
int oph_text2mel_graph(oph_handle* h, const float* K, const float* V, const float* mels, const int32_t* prev_max,
                       const int32_t* ends, const int32_t* spk, int B,
                       float* Q, float* R, float* Y_logits, float* Y, float* alignments, int32_t* max_attentions);
/* oph_text2mel_durations  replaces synth_codedtext2mel() when hp.use_external_durations is set (synthesize.py:168-169,
 *     175-176, 211-216): the attention is the externally supplied selection matrix (networks.FixedAttention 327-358),
 *     K is ignored by the graph and may be NULL.
 *     durations (B,max_T,max_N) float32: rows are one-hot (weight exactly 1) or all zero, as data_load.py:243-251
 *     builds them from the transcript's duration field.  t_ends[b] = number of selected frames of utterance b;
 *     steps executed = n_steps if > 0, else min(max_T, max_b t_ends + 1)  (the reference's break rule; n_steps lets a
 *     sharded batch use the maximum over ALL ranks).  Frames / alignment columns after the last step stay 0.       */
int oph_text2mel_durations(oph_handle* h, const float* K, const float* V, const float* durations,
                           const int32_t* spk, int B, int n_steps,
                           float* Y, int32_t* t_ends, float* alignments, int32_t* steps_run);

/* ---- device-resident pipeline (what bench.py times; no PCIe in the timed region)
 * oph_stage_text copies L / ends / spk into HBM.  oph_run_resident runs
 * encode_text -> decode loop -> (optionally) SSRN entirely on the device, leaving
 * K,V,Y,alignments,Z in HBM; oph_fetch_* copy results out afterwards.
 *   run_ssrn: 0 Text2Mel only; 1 SSRN too (streamed under the decode, tail joined); 2 pipelined batches: the SSRN tail
 *   of this batch overlaps the next call (oph_synchronize / oph_fetch_* / oph_timer_stop join it).
 * oph_stage_text_next stages the text of the FOLLOWING batch (second text slot, same B).  While the staged batch decodes,
 *   the TextEnc of that next text runs on the SSRN partition; the run after it switches over and starts decoding at
 *   once.  Called after the staged batch has run, it first makes the previously staged next text current, so a caller
 *   with a different text per batch (synthesize.py:477-482) alternates
 *       oph_stage_text(t0); oph_stage_text_next(t1);  loop { run; oph_stage_text_next(t[i+2]); }
 * oph_run_host is the whole path host -> host in one call (the reference's timed region, synthesize.py:553-576, for the
 *   staged text): every result is copied into the caller's buffer (NULL = not wanted) as it becomes final, Z chunk by chunk
 *   while the decode runs.  Buffers from oph_host_alloc (pinned) make those copies asynchronous.
 * oph_decode_steps continues the decode loop of the staged batch over steps
 * [t_begin, t_end) (multi-GPU global-stop fix-up).                  */
int oph_stage_text(oph_handle* h, const int32_t* L, const int32_t* ends,
                   const int32_t* spk, int B);
int oph_stage_text_next(oph_handle* h, const int32_t* L, const int32_t* ends,
                        const int32_t* spk, int B);
int oph_run_resident(oph_handle* h, int stop_mode, int run_ssrn, int32_t* steps_run);
int oph_run_host(oph_handle* h, int stop_mode, float* K, float* V, float* Y, int32_t* t_ends,
                 float* alignments, float* Z, int32_t* steps_run);
int oph_host_alloc(size_t bytes, void** out);     /* pinned host memory for result buffers */
int oph_host_free(void* p);
int oph_decode_steps(oph_handle* h, int t_begin, int t_end, int stop_mode, int32_t* steps_run);
int oph_run_ssrn_resident(oph_handle* h);
int oph_fetch_kv(oph_handle* h, float* K, float* V);
int oph_fetch_mel(oph_handle* h, float* Y, int32_t* t_ends, float* alignments);
int oph_fetch_mag(oph_handle* h, float* Z);
int oph_synchronize(oph_handle* h);
/* Device address of the SSRN output of the last resident run: (B, r*max_T, full_dim) fp32, utterance b at
 * *d_mag + b * *utt_stride floats.  Synchronises first; valid until the next run on this handle.  Lets the vocoder
 * library (ophelia_vocoder.h) consume Z without the host round trip of synthesize.py:585-617. */
int oph_device_mag(oph_handle* h, const float** d_mag, int64_t* utt_stride, int32_t* B);
/* Contraction arithmetic of the large batched contractions.  Every multiply-accumulate is fp32 x fp32 -> fp32; the choice is
 * how the products are formed:
 *   0  v_mfma_f32_32x32x2_f32 (fp32 operands);
 *   2  (default) each fp32 operand split into hi + lo fp16 terms (22 of fp32's 24 significant bits), a.b ~ ah.bh + ah.bl +
 *      al.bh on the fp16 MFMA with fp32 accumulation: 2.4e-7 relative per product, below the rounding of the fp32
 *      accumulation itself -- measured in the same accuracy class as mode 0 (SSRN mag vs the CPU oracle 5.0e-6 vs 4.4e-6; mel
 *      differs from the mode-0 decode by 6.7e-6, two mode-0 decode flavours among themselves by 5.6e-6; identical attention
 *      traces), 2-3x faster;
 *   1  the same with bf16 terms (16 bits): ~1e-5 relative per product (mag 2.9e-5); SSRN only -- Text2Mel feeds an argmax
 *      back into itself and is only offered fp32-class arithmetic.
 * oph_set_ssrn_precision(h, mode) = oph_set_precision(h, 0, mode).  which: 0 SSRN, 1 the two many-row levels of the
 * AudioDec history cone, 2 TextEnc.  The decoder chain itself (dec_loop) is always mode 0. */
int oph_set_ssrn_precision(oph_handle* h, int mode);
int oph_set_precision(oph_handle* h, int which, int mode);

/* ---- measurement helpers (HIP events on the handle's own stream) ---------------- */
int oph_timer_start(oph_handle* h);
int oph_timer_stop(oph_handle* h, float* elapsed_ms);      /* synchronises */
/* Per-kernel-class accounting: when enabled every launch of each kernel class is
 * bracketed by HIP events on the launch stream.  oph_profile_get returns, for
 * class index i, its name, number of launches, summed device time, and the summed
 * ALGORITHMIC bytes and flops of those launches (DESIGN.md gives the formulas).
 * on = 2 brackets only the whole-decode launch (dec_loop: one event pair per decode,
 * cheap enough to stay on inside a timed region); 1 = every class; 0 = off.       */
/* Device-side witness of the whole-decode launches (dec_chain / dec_loop) since the last reset: every workgroup stores the
 * 100 MHz constant clock (s_memrealtime) when it enters and when it leaves; per launch the library keeps the minimum of the
 * former and the maximum of the latter.  *launches launches, *total_us = sum of (last out - first in).  Independent of HIP
 * events and of any profiler: bench.py reports it beside the event-based average (roofline.device_clock_us). */
int oph_loop_clock(oph_handle* h, int64_t* launches, double* total_us, int reset);
int oph_profile_enable(oph_handle* h, int on);
int oph_profile_reset(oph_handle* h);
int oph_profile_count(const oph_handle* h);
int oph_profile_get(oph_handle* h, int index, char* name, int name_cap, int64_t* launches,
                    double* total_ms, double* alg_bytes, double* alg_flops);

/* ---- per-operator entry points (unit parity; host buffers, (B,T,C) fp32) ---------
 * Same kernels as the model path.  `padding`: 0 = SAME, 1 = CAUSAL.
 * `act`: 0 none, 1 relu, 2 sigmoid.
 * oph_op_embed            modules.py:15-44      table (vocab,units), row 0 zeroed at lookup
 * oph_op_layernorm        modules.py:47-75      eps 1e-12, biased variance
 * oph_op_conv1d           modules.py:91-146     kernel (size,Cin,Cout)
 * oph_op_hc               modules.py:148-207    kernel (size,C,2C)
 * oph_op_conv1d_transpose modules.py:209-258    kernel (1,3,Cout,Cin), stride 2, 'same'
 * oph_op_attention        networks.py:286-325   monotonic branch; one mask per utterance
 */
int oph_op_embed(int device, const int32_t* ids, int64_t n_ids, const float* table,
                 int vocab, int units, float* out);
int oph_op_layernorm(int device, const float* x, int64_t rows, int C, const float* gamma,
                     const float* beta, float* y);
int oph_op_conv1d(int device, const float* x, int B, int T, int Cin, int Cout, int size, int rate,
                  int padding, const float* kernel, const float* bias, const float* gamma,
                  const float* beta, int act, float* y);
int oph_op_hc(int device, const float* x, int B, int T, int C, int size, int rate, int padding,
              const float* kernel, const float* bias, const float* gamma1, const float* beta1,
              const float* gamma2, const float* beta2, float* y);
int oph_op_conv1d_transpose(int device, const float* x, int B, int T, int Cin, int Cout,
                            const float* kernel, const float* bias, const float* gamma,
                            const float* beta, float* y /* (B,2T,Cout) */);
/* The same layer through the launches the SSRN path makes at one of oph_set_precision's arithmetic codes: 0 = fp32-operand
 * MFMA (= oph_op_conv1d_transpose), 1 = split-bf16 x3 (both phases in one launch, then the LayerNorm rows), 2 = split-fp16 x3
 * (the input split into fp16 hi / lo planes as the previous layer's launch writes them in SSRN, both phases as one plane_gemm
 * problem with the layer's LayerNorm inside the launch -- round 6; channel counts that are not a multiple of 64 or exceed 1024:
 * plane_gemm, then the LayerNorm rows). */
int oph_op_conv1d_transpose_prec(int device, const float* x, int B, int T, int Cin, int Cout,
                                 const float* kernel, const float* bias, const float* gamma,
                                 const float* beta, int precision, float* y /* (B,2T,Cout) */);
int oph_op_attention(int device, const float* Q, const float* K, const float* V,
                     const int32_t* prev_max, int B, int T, int N, int d, int win,
                     float* R /*(B,T,2d)*/, float* alignments /*(B,N,T)*/,
                     int64_t* max_attentions /*(B,T)*/);
/* oph_bench_conv1d_transpose: device-resident timing of the conv1d_transpose launches (SSRN D_4 / D_7) on seeded random
 * data already in HBM -- the measurement behind bench.py's kernel_rooflines.  precision: 0 exact fp32 MFMA, 1 split-bf16 x3,
 * 2 split-fp16 x3 as SSRN runs it (operands arrive as fp16 planes; ONE launch: plane_gemm with the LayerNorm inside, fp32 rows and
 * planes out), 10 the same as round 5 ran it (plane_gemm + LayerNorm rows: two launches),
 * 5 split-fp16 x3 as round 3 ran it (fp32 rows split inside the paired contraction); 3 / 4: precision 5 with two / one of the
 * three products; 6..9: ablation builds of plane_gemm (no MFMAs / no operand stream / no stores / three K blocks), 11 / 12: the fused
 * launch without its wait for the partners / without its plane stores -- measurement only, present only in a library built with
 * -DOPH_ABLATE (OPH_ERR_UNSUPPORTED otherwise).
 * Returns the average time of one layer evaluation and its ALGORITHMIC bytes / flops. */
int oph_bench_conv1d_transpose(int device, int B, int T, int Cin, int Cout, int precision, int warmup, int iters,
                               double* avg_us, double* alg_bytes, double* alg_flops);
const char* oph_op_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* OPHELIA_HIP_H */
