# -*- coding: utf-8 -*-
"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

NumPy fp32 CPU restatement of Ophelia's Text2Mel + SSRN *synthesis* path
(reference: CSTR-Edinburgh/ophelia, files cited per function as file:line
relative to the reference tree).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; nothing under
ophelia_amd/ does.

PARITY STATUS: "parity unpinned" at the TensorFlow-primitive level.
The reference's arithmetic lives in tensorflow-gpu==1.12.0 (reference
requirements.txt:44), which is not vendored and not installable here, and the
reference holds no tests / golden vectors for this path.
What IS pinned:
  * network wiring (layer order, dilations, scopes, which layers have ReLU,
    attention mask logic, shift-by-one, concat order) -- by executing the
    reference's own architectures.py/networks.py/modules.py over the small eager
    stand-in in tests/golden/tf_standin.py and comparing with this file
    (tests/golden/make_golden.py -> tests/golden/*.npz, tests/test_oracle_golden.py);
  * each primitive (conv, LayerNorm, transposed conv, softmax) -- against the
    independent torch CPU implementations (tests/test_oracle_primitives.py).
The TF-1.12 operator semantics themselves ([TF-sem] notes below) are stated from
knowledge of TF 1.12 and cannot be executed here.

All tensors are channels-last (B, T, C) float32, as at every reference interface
(modules.py:106).  Weights live in a dict keyed by the TF variable names the
reference creates, e.g.
  'Text2Mel/TextEnc/HC_4/conv1d/kernel' (3, 512, 1024)
  'SSRN/D_4/conv2d_transpose/kernel'     (1, 3, Cout, Cin)
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
LN_EPS = F32(1e-12)        # [TF-sem] tf.contrib.layers.layer_norm variance_epsilon
MASK_VALUE = F32(-2.0 ** 32 + 1)  # networks.py:312  (rounds to -4294967296.0 in fp32)


# --------------------------------------------------------------------------
# primitives  (modules.py)
# --------------------------------------------------------------------------
def sigmoid(x):
    x = np.asarray(x, F32)
    return (F32(1) / (F32(1) + np.exp(-x, dtype=F32))).astype(F32)


def relu(x):
    return np.maximum(x, F32(0))


def embed(ids, table, zero_pad=True):
    """modules.py:15-44.  Row 0 of the table is replaced by zeros at lookup time
    (modules.py:38-40), not in the stored variable."""
    table = np.asarray(table, F32)
    if zero_pad:
        table = np.concatenate((np.zeros((1, table.shape[1]), F32), table[1:]), 0)
    return table[np.asarray(ids)]


def normalize(x, gamma, beta):
    """modules.py:47-75 -> tf.contrib.layers.layer_norm(begin_norm_axis=-1).
    [TF-sem] mean and *biased* variance over the last axis (nn.moments),
    y = (x - mean) * rsqrt(var + 1e-12) * gamma + beta."""
    x = np.asarray(x, F32)
    mean = x.mean(axis=-1, keepdims=True, dtype=F32)
    xc = x - mean
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=F32)
    inv = (F32(1) / np.sqrt(var + LN_EPS)).astype(F32)
    return (xc * inv * gamma + beta).astype(F32)


def _conv_taps(x, kernel, bias, rate, padding):
    """tf.layers.conv1d on (B,T,Cin) with kernel (size,Cin,Cout).
    CAUSAL: modules.py:123-127 left-pads (size-1)*rate zeros then VALID, so tap k
    reads x[t-(size-1-k)*rate].  [TF-sem] SAME with dilation pads (size-1)*rate in
    total, left = total//2, so tap k reads x[t+(k-(size-1)/2)*rate] for odd size."""
    x = np.asarray(x, F32)
    B, T, Cin = x.shape
    size, Cin2, Cout = kernel.shape
    assert Cin == Cin2, (Cin, Cin2)
    pad_total = (size - 1) * rate
    if padding.lower() == "causal":
        left = pad_total
    elif padding.lower() == "same":
        left = pad_total // 2
    else:
        raise ValueError(padding)
    xp = np.zeros((B, T + pad_total, Cin), F32)
    xp[:, left:left + T] = x
    out = np.zeros((B, T, Cout), F32)
    for k in range(size):
        seg = xp[:, k * rate:k * rate + T]          # x[t + k*rate - left]
        out += (seg.reshape(B * T, Cin) @ kernel[k]).reshape(B, T, Cout)
    if bias is not None:
        out += bias
    return out.astype(F32)


def _norm(h, W, prefix):
    """modules.normalize with hp.norm in ('layer', None): with norm=None the graph creates no gamma/beta variables
    (modules.py:62-74), so the weight dictionary itself says which one applies."""
    if prefix + "/gamma" in W:
        return normalize(h, W[prefix + "/gamma"], W[prefix + "/beta"])
    return np.asarray(h, F32)


def _lcc_gate(W, scope, speakers, like):
    """modules.learn_channel_contributions (78-88): sigmoid(embed(codes)) broadcast over time; the table's row 0 reads as
    zeros (embed zero_pad) -> gate 0.5.  Returns None when the layer has no LCC table or no speakers are given."""
    key = scope + "/lcc_embed/lookup_table"
    if key not in W or speakers is None:
        return None
    B = like.shape[0]
    g = sigmoid(embed(np.asarray(speakers).reshape(B, 1).astype(np.int64), W[key]))      # (B,1,C)
    return g if like.ndim == 3 else g[:, 0]


def conv1d(x, W, scope, rate=1, padding="SAME", activation_fn=None, speakers=None):
    """modules.py:91-146: [causal pad] -> conv -> LayerNorm (unless hp.norm is None) -> activation.
    Dropout is identity at synthesis (training=False)."""
    h = _conv_taps(x, W[scope + "/conv1d/kernel"], W[scope + "/conv1d/bias"], rate, padding)
    h = _norm(h, W, scope + "/normalize")
    if activation_fn is not None:
        h = activation_fn(h)
    g = _lcc_gate(W, scope, speakers, h)              # modules.py:143-144: after activation (and dropout)
    if g is not None:
        h = (g * h).astype(F32)
    return h


def hc(x, W, scope, rate=1, padding="SAME", speakers=None):
    """modules.py:148-207 highway conv: conv to 2C -> split H1,H2 -> separate LN on
    each (scopes H1, H2) -> out = sigmoid(H1)*H2 + (1-sigmoid(H1))*x."""
    h = _conv_taps(x, W[scope + "/conv1d/kernel"], W[scope + "/conv1d/bias"], rate, padding)
    C = h.shape[-1] // 2
    H1 = _norm(h[..., :C], W, scope + "/H1")
    H2 = _norm(h[..., C:], W, scope + "/H2")
    lg = _lcc_gate(W, scope, speakers, H2)            # modules.py:200-201: on the transformation branch only
    if lg is not None:
        H2 = (lg * H2).astype(F32)
    g = sigmoid(H1)
    return (g * H2 + (F32(1) - g) * x).astype(F32)


def conv1d_transpose(x, W, scope):
    """modules.py:209-258 -> tf.layers.conv2d_transpose(kernel (1,3), strides (1,2),
    'same'), kernel variable laid out (1, 3, Cout, Cin), then LayerNorm.
    [TF-sem] = gradient of a stride-2 SAME conv (pad 0 left / 1 right):
      o[2t] = x[t]·Kt[0,0]^T + x[t-1]·Kt[0,2]^T + b ;  o[2t+1] = x[t]·Kt[0,1]^T + b."""
    x = np.asarray(x, F32)
    Kt = W[scope + "/conv2d_transpose/kernel"]
    b = W[scope + "/conv2d_transpose/bias"]
    B, T, Cin = x.shape
    assert Kt.shape[0] == 1 and Kt.shape[1] == 3 and Kt.shape[3] == Cin
    Cout = Kt.shape[2]
    xf = x.reshape(B * T, Cin)
    xprev = np.zeros_like(x)
    xprev[:, 1:] = x[:, :-1]
    even = xf @ Kt[0, 0].T + xprev.reshape(B * T, Cin) @ Kt[0, 2].T
    odd = xf @ Kt[0, 1].T
    out = np.empty((B, 2 * T, Cout), F32)
    out[:, 0::2] = even.reshape(B, T, Cout)
    out[:, 1::2] = odd.reshape(B, T, Cout)
    out += b
    return _norm(out, W, scope + "/normalize")


# --------------------------------------------------------------------------
# networks  (networks.py)
# --------------------------------------------------------------------------
def _speaker_reps(speakers, T, table):
    """tf.tile(speaker_codes, [1, T]) -> embed (zero_pad: speaker 0 -> zeros)  (networks.py:139-143)"""
    B = len(speakers)
    codes = np.tile(np.asarray(speakers).reshape(B, 1).astype(np.int64), (1, T))
    return embed(codes, table)


def text_enc(hp, L, W, scope="Text2Mel/TextEnc", speakers=None):
    """networks.py:121-212 with the 'text_encoder_input' (138-144) and 'text_encoder_towards_end' (184-199) speaker
    hooks; 'learn_channel_contributions' is not covered."""
    i = 1
    t = embed(L, W["%s/embed_%d/lookup_table" % (scope, i)]); i += 1
    if "text_encoder_input" in hp.multispeaker:
        reps = _speaker_reps(speakers, t.shape[1], W["%s/embed_%d/lookup_table" % (scope, i)]); i += 1
        t = np.concatenate((t, reps), -1)
    t = conv1d(t, W, "%s/C_%d" % (scope, i), activation_fn=relu, speakers=speakers); i += 1
    t = conv1d(t, W, "%s/C_%d" % (scope, i), speakers=speakers); i += 1
    for _ in range(2):
        for j in range(4):
            t = hc(t, W, "%s/HC_%d" % (scope, i), rate=3 ** j, speakers=speakers); i += 1
    for _ in range(2):
        t = hc(t, W, "%s/HC_%d" % (scope, i), rate=1, speakers=speakers); i += 1
    if "text_encoder_towards_end" in hp.multispeaker:
        reps = _speaker_reps(speakers, t.shape[1], W["%s/embed_%d/lookup_table" % (scope, i)]); i += 1
        t = np.concatenate((t, reps), -1)
        t = conv1d(t, W, "%s/C_%d" % (scope, i), activation_fn=relu); i += 1   # squash hidden+embedding back to 2d
    for _ in range(2):                       # size-1 highway convs, networks.py:200-208
        t = hc(t, W, "%s/HC_%d" % (scope, i), rate=1, speakers=speakers); i += 1
    d = t.shape[-1] // 2
    return t[..., :d].copy(), t[..., d:].copy()


def audio_enc(hp, S, W, scope="Text2Mel/AudioEnc", speakers=None):
    """networks.py:214-284 incl. the 'audio_encoder_input' speaker hook (237-245) and the LCC gates."""
    i = 1
    t = conv1d(S, W, "%s/C_%d" % (scope, i), padding="CAUSAL", activation_fn=relu, speakers=speakers); i += 1
    if "audio_encoder_input" in hp.multispeaker:          # networks.py:237-245: embed, concat, 1x1 conv (no LCC, no act)
        reps = _speaker_reps(speakers, t.shape[1], W["%s/embed_%d/lookup_table" % (scope, i)]); i += 1
        t = conv1d(np.concatenate((t, reps), -1), W, "%s/C_%d" % (scope, i)); i += 1
    t = conv1d(t, W, "%s/C_%d" % (scope, i), padding="CAUSAL", activation_fn=relu, speakers=speakers); i += 1
    t = conv1d(t, W, "%s/C_%d" % (scope, i), padding="CAUSAL", speakers=speakers); i += 1
    for _ in range(2):
        for j in range(4):
            t = hc(t, W, "%s/HC_%d" % (scope, i), rate=3 ** j, padding="CAUSAL", speakers=speakers); i += 1
    for _ in range(2):
        t = hc(t, W, "%s/HC_%d" % (scope, i), rate=3, padding="CAUSAL", speakers=speakers); i += 1
    return t


def attention(hp, Q, K, V, prev_max_attentions):
    """networks.py:286-325, monotonic (synthesis) branch with the FIA mask on.
    The same mask (a function of prev_max only) is tiled over every query position
    (networks.py:311).  Returns R (B,T,2d), alignments (B,N,T), max_attentions (B,T)."""
    Q = np.asarray(Q, F32); K = np.asarray(K, F32); V = np.asarray(V, F32)
    B, T, d = Q.shape
    N = K.shape[1]
    scale = F32(1.0) / np.sqrt(F32(hp.d))            # tf.rsqrt(tf.to_float(hp.d))
    A = np.einsum("btd,bnd->btn", Q, K).astype(F32) * scale
    p = np.asarray(prev_max_attentions).astype(np.int64).reshape(B, 1)
    n = np.arange(N).reshape(1, N)
    key_masks = n < p                                # sequence_mask(prev_max, max_N)
    rev_len = hp.max_N - hp.attention_win_size - p   # sequence_mask(len)[:, ::-1]
    reverse_masks = (N - 1 - n) < rev_len            # [TF-sem] len<=0 -> all False
    masks = np.logical_or(key_masks, reverse_masks)  # (B,N)
    if getattr(hp, "turn_off_monotonic_for_synthesis", False):
        # networks.py:307-309: no forcibly-incremental window; only keys past the text are masked.
        # hp.text_lengths = get_text_lengths(L) + 1 is set by the host (synthesize.py:505-507)
        tl = np.asarray(hp.text_lengths).astype(np.int64).reshape(B, 1)
        masks = (N - 1 - n) < (hp.max_N - tl)
    A = np.where(masks[:, None, :], MASK_VALUE, A).astype(F32)
    A = A - A.max(axis=-1, keepdims=True)
    E = np.exp(A, dtype=F32)
    A = (E / E.sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)
    max_attentions = A.argmax(-1)                    # first max on ties [TF-sem]
    R = np.einsum("btn,bnd->btd", A, V).astype(F32)
    if getattr(hp, "concatenate_query", True):       # networks.py:317-319
        R = np.concatenate((R, Q), -1)
    alignments = np.transpose(A, (0, 2, 1))
    return R, alignments, max_attentions


def fixed_attention(hp, D, Q, V):
    """networks.FixedAttention (327-358): an externally supplied (B,T,N) selection matrix replaces softmax(QK^T);
    K is never used.  Returns R (B,T,2d), alignments (B,N,T), max_attentions (B,T)."""
    D = np.asarray(D, F32)
    Q = np.asarray(Q, F32); V = np.asarray(V, F32)
    max_attentions = D.argmax(-1)
    R = np.einsum("btn,bnd->btd", D, V).astype(F32)
    if getattr(hp, "concatenate_query", True):
        R = np.concatenate((R, Q), -1)
    return R, np.transpose(D, (0, 2, 1)), max_attentions


def audio_dec(hp, R, W, speakers=None, scope="Text2Mel/AudioDec"):
    """networks.py:360-435.  `speakers` (B,1) int only when
    'audio_decoder_input' in hp.multispeaker (vctk_01.cfg)."""
    i = 1
    t = conv1d(R, W, "%s/C_%d" % (scope, i), padding="CAUSAL"); i += 1
    if "audio_decoder_input" in hp.multispeaker:
        B, T, _ = t.shape
        codes = np.tile(np.asarray(speakers).reshape(B, 1), (1, T))
        reps = embed(codes, W["%s/embed_%d/lookup_table" % (scope, i)]); i += 1
        t = np.concatenate((t, reps), -1)
        t = conv1d(t, W, "%s/C_%d" % (scope, i)); i += 1     # default SAME, size 1
    for j in range(4):
        t = hc(t, W, "%s/HC_%d" % (scope, i), rate=3 ** j, padding="CAUSAL", speakers=speakers); i += 1
    for _ in range(2):
        t = hc(t, W, "%s/HC_%d" % (scope, i), rate=1, padding="CAUSAL", speakers=speakers); i += 1
    for _ in range(3):
        t = conv1d(t, W, "%s/C_%d" % (scope, i), padding="CAUSAL", activation_fn=relu, speakers=speakers); i += 1
    logits = conv1d(t, W, "%s/C_%d" % (scope, i), padding="CAUSAL", speakers=speakers); i += 1
    Y = sigmoid(logits) if getattr(hp, "squash_output_t2m", True) else logits
    return logits, Y


def ssrn(hp, Y, W, scope="SSRN", speakers=None):
    """networks.py:437-537.  `speakers` (B,1) int only when 'ssrn_input' in hp.multispeaker (networks.py:457-465: no shipped
    config sets it, and the reference's own synth_mel2mag does not feed g.speakers -- synthesize.py:250-260 -- so it is reachable
    through the graph surface only)."""
    i = 1
    t = conv1d(Y, W, "%s/C_%d" % (scope, i)); i += 1
    if "ssrn_input" in hp.multispeaker:
        reps = _speaker_reps(speakers, t.shape[1], W["%s/embed_%d/lookup_table" % (scope, i)]); i += 1
        t = np.concatenate((t, reps), -1)
        t = conv1d(t, W, "%s/C_%d" % (scope, i)); i += 1
    for j in range(2):
        t = hc(t, W, "%s/HC_%d" % (scope, i), rate=3 ** j); i += 1
    n_transposes = {4: 2, 8: 3}[hp.r]
    for _ in range(n_transposes):
        t = conv1d_transpose(t, W, "%s/D_%d" % (scope, i)); i += 1
        for j in range(2):
            t = hc(t, W, "%s/HC_%d" % (scope, i), rate=3 ** j); i += 1
    t = conv1d(t, W, "%s/C_%d" % (scope, i)); i += 1
    for _ in range(2):
        t = hc(t, W, "%s/HC_%d" % (scope, i), rate=1); i += 1
    t = conv1d(t, W, "%s/C_%d" % (scope, i)); i += 1
    for _ in range(2):
        t = conv1d(t, W, "%s/C_%d" % (scope, i), activation_fn=relu); i += 1
    logits = conv1d(t, W, "%s/C_%d" % (scope, i))
    Z = sigmoid(logits) if getattr(hp, "squash_output_ssrn", True) else logits
    return logits, Z


# --------------------------------------------------------------------------
# graph + host loop  (architectures.py, synthesize.py)
# --------------------------------------------------------------------------
def text2mel_graph(hp, W, K, V, mels, prev_max_attentions, speakers=None, durations=None):
    """architectures.py:188-239, mode 'synthesize', with K and V fed."""
    S = np.concatenate((np.zeros_like(mels[:, :1]), mels[:, :-1]), 1)   # :191
    Q = audio_enc(hp, S, W, speakers=speakers)
    if getattr(hp, "use_external_durations", False):                    # :222-223
        R, alignments, max_attentions = fixed_attention(hp, durations, Q, V)
    else:
        R, alignments, max_attentions = attention(hp, Q, K, V, prev_max_attentions)
    _, Y = audio_dec(hp, R, W, speakers)
    return Y, max_attentions, alignments


def encode_text(hp, W, L, speakers=None):
    """synthesize.py:232-240."""
    return text_enc(hp, np.asarray(L), W, speakers=speakers)


def get_text_lengths(L):
    """synthesize.py:242-247 (IndexError if a row has no padding)."""
    return np.array([np.where(L[i, :] == 0)[0][0] for i in range(len(L))])


def synth_codedtext2mel(hp, W, K, V, ends, speakers=None, stop=True, trace=None, durations=None):
    """synthesize.py:150-230, faithful: the whole (B,max_T) graph is recomputed at
    every step and only column j is kept.  stop=False runs all max_T steps
    (the fixed-length timed configuration)."""
    B = len(K)
    Y = np.zeros((B, hp.max_T, hp.n_mels), F32)
    alignments = np.zeros((B, hp.max_N, hp.max_T), F32)
    prev_max = np.zeros((B,), np.int32)
    ends = np.asarray(ends)
    endcounts = np.zeros(ends.shape, dtype=int)
    t_ends = np.ones(ends.shape, dtype=int) * hp.max_T
    fixed = getattr(hp, "use_external_durations", False)
    if fixed:
        t_ends = np.asarray(durations).sum(axis=(1, 2)).astype(int)      # synthesize.py:168-169
    for j in range(hp.max_T):
        _Y, _max, _al = text2mel_graph(hp, W, K, V, Y, prev_max, speakers, durations)
        Y[:, j, :] = _Y[:, j, :]
        alignments[:, :, j] = _al[:, :, j]
        prev_max = _max[:, j].astype(np.int32)
        if trace is not None:
            trace.append(prev_max.copy())
        if fixed:                                   # synthesize.py:211-216: stop once the longest utterance is through
            if j >= t_ends.max():
                break
            continue
        reached_end = (_max[:, j] >= ends)
        endcounts += reached_end
        for i in range(B):
            if t_ends[i] == hp.max_T and endcounts[i] >= 1:
                t_ends[i] = j
        if stop and (t_ends < hp.max_T).all():
            break
    return Y, t_ends.tolist(), alignments


# ---- exact incremental (O(T)) restatement ------------------------------------
# NOTE (found while pinning this oracle against the reference-wiring goldens): the
# reference graph applies the CURRENT step's monotonic mask to EVERY query position
# (networks.py:311 tiles one mask over max_T), so at step j the AudioDec input R[t]
# for past t<j is re-evaluated under prev_max(j), not under the mask that was in
# force at step t.  AudioDec's hidden history is therefore a function of the current
# prev_max and cannot be cached across steps; only AudioEnc (causal, mask-free) can.
# Exact O(T) form: cache AudioEnc per-layer history and Q[t]; at each step recompute
# AudioDec over its causal receptive field (2*(1+3+9+27+1+1) = 84 frames back).
AUDIODEC_LOOKBACK = 84


def _hist_conv(hist, j, kernel, bias, rate):
    """causal conv output at time j from a (B, >=j+1, Cin) history buffer."""
    size = kernel.shape[0]
    B = hist.shape[0]
    out = np.zeros((B, kernel.shape[2]), F32)
    for k in range(size):
        tt = j - (size - 1 - k) * rate
        if tt >= 0:
            out += hist[:, tt] @ kernel[k]
    return out + bias


def _push(st, name, x, j):
    buf = st.get(name)
    if buf is None:
        buf = st[name] = np.zeros((x.shape[0], st["__T__"], x.shape[1]), F32)
    buf[:, j] = x
    return buf


def _inc_conv1d(st, name, x, j, W, act=None, speakers=None):
    h = _hist_conv(_push(st, name, x, j), j, W[name + "/conv1d/kernel"], W[name + "/conv1d/bias"], 1)
    h = _norm(h, W, name + "/normalize")
    h = act(h) if act is not None else h
    lg = _lcc_gate(W, name, speakers, h)
    return h if lg is None else (lg * h).astype(F32)


def _inc_hc(st, name, x, j, W, rate, speakers=None):
    h = _hist_conv(_push(st, name, x, j), j, W[name + "/conv1d/kernel"], W[name + "/conv1d/bias"], rate)
    C = h.shape[-1] // 2
    g = sigmoid(_norm(h[:, :C], W, name + "/H1"))
    u = _norm(h[:, C:], W, name + "/H2")
    lg = _lcc_gate(W, name, speakers, u)
    if lg is not None:
        u = (lg * u).astype(F32)
    return (g * u + (F32(1) - g) * x).astype(F32)


def synth_codedtext2mel_incremental(hp, W, K, V, ends, speakers=None, stop=True, trace=None,
                                    forced_prev_max=None, margins=None, max_steps=None, step_times=None, durations=None):
    """Same contract and (up to fp reassociation) same outputs as synth_codedtext2mel.
    forced_prev_max: optional (steps,B) int array -- teacher-forced attention
    positions (separates numerics from argmax flips in parity tests).
    margins: optional list; receives per step the (B,) gap between the two largest
    attention probabilities of row j (how close the argmax is to flipping)."""
    B = len(K)
    T = hp.max_T
    st = {"__T__": T}
    Y = np.zeros((B, T, hp.n_mels), F32)
    Qh = np.zeros((B, T, hp.d), F32)
    alignments = np.zeros((B, hp.max_N, T), F32)
    prev_max = np.zeros((B,), np.int32)
    ends = np.asarray(ends)
    t_ends = np.ones(ends.shape, dtype=int) * T
    fixed = getattr(hp, "use_external_durations", False)
    if fixed:
        t_ends = np.asarray(durations).sum(axis=(1, 2)).astype(int)
    ae = "Text2Mel/AudioEnc"
    import time as _time
    for j in range(T if max_steps is None else min(T, max_steps)):
        _t0 = _time.perf_counter()
        x = Y[:, j - 1] if j > 0 else np.zeros((B, hp.n_mels), F32)
        i = 1
        x = _inc_conv1d(st, "%s/C_%d" % (ae, i), x, j, W, relu, speakers); i += 1
        if "audio_encoder_input" in hp.multispeaker:
            tab = W["%s/embed_%d/lookup_table" % (ae, i)]; i += 1
            reps = embed(np.asarray(speakers).reshape(B).astype(np.int64), tab)
            x = _inc_conv1d(st, "%s/C_%d" % (ae, i), np.concatenate((x, reps), -1), j, W); i += 1
        x = _inc_conv1d(st, "%s/C_%d" % (ae, i), x, j, W, relu, speakers); i += 1
        x = _inc_conv1d(st, "%s/C_%d" % (ae, i), x, j, W, None, speakers); i += 1
        for _ in range(2):
            for jj in range(4):
                x = _inc_hc(st, "%s/HC_%d" % (ae, i), x, j, W, 3 ** jj, speakers); i += 1
        for _ in range(2):
            x = _inc_hc(st, "%s/HC_%d" % (ae, i), x, j, W, 3, speakers); i += 1
        Qh[:, j] = x
        lo = max(0, j - AUDIODEC_LOOKBACK)
        if fixed:
            R, al, mx = fixed_attention(hp, np.asarray(durations)[:, lo:j + 1], Qh[:, lo:j + 1], V)
        else:
            R, al, mx = attention(hp, Qh[:, lo:j + 1], K, V, prev_max)   # current mask, all rows
        alignments[:, :, j] = al[:, :, -1]
        m = mx[:, -1].astype(np.int32)
        if margins is not None:
            srt = np.sort(al[:, :, -1], axis=1)
            margins.append((srt[:, -1] - srt[:, -2]).copy())
        _, Yw = audio_dec(hp, R, W, speakers)                       # stateless cone
        Y[:, j] = Yw[:, -1]
        prev_max = m if forced_prev_max is None else np.asarray(forced_prev_max[j], np.int32)
        if trace is not None:
            trace.append(m.copy())
        if fixed:
            if step_times is not None:
                step_times.append(_time.perf_counter() - _t0)
            if j >= t_ends.max():
                break
            continue
        reached = m >= ends
        for b in range(B):
            if t_ends[b] == T and reached[b]:
                t_ends[b] = j
        if step_times is not None:
            step_times.append(_time.perf_counter() - _t0)
        if stop and (t_ends < T).all():
            break
    return Y, t_ends.tolist(), alignments


def synth_mel2mag(hp, W, Y, batchsize=128, speakers=None):
    """synthesize.py:250-260.  nbatches = max(1, len(Y) / batchsize) is Python-2
    integer division.  (speakers: only for 'ssrn_input', which the reference's function cannot feed -- see ssrn.)"""
    if batchsize > 0:
        nbatches = max(1, len(Y) // batchsize)
        batches = np.array_split(Y, nbatches)
        spk = [None] * nbatches if speakers is None else np.array_split(np.asarray(speakers), nbatches)
    else:
        batches, spk = [Y], [speakers]
    return np.concatenate([ssrn(hp, Yb, W, speakers=sb)[1] for Yb, sb in zip(batches, spk)])


# --------------------------------------------------------------------------
# variable inventory + seeded synthetic weights
# --------------------------------------------------------------------------
def variable_shapes(hp):
    """Ordered {tf_variable_name: shape} for the synth-mode graphs."""
    out = {}

    ln = getattr(hp, "norm", "layer") == "layer"     # hp.norm None: no gamma/beta variables at all
    lcc_on = "learn_channel_contributions" in hp.multispeaker

    def conv(scope, cin, cout, size=1, lcc=True):
        out[scope + "/conv1d/kernel"] = (size, cin, cout)
        out[scope + "/conv1d/bias"] = (cout,)
        if ln:
            out[scope + "/normalize/beta"] = (cout,)
            out[scope + "/normalize/gamma"] = (cout,)
        if lcc_on and lcc:
            out[scope + "/lcc_embed/lookup_table"] = (hp.nspeakers, cout)

    def hcl(scope, c, size=3, lcc=True):
        out[scope + "/conv1d/kernel"] = (size, c, 2 * c)
        out[scope + "/conv1d/bias"] = (2 * c,)
        for h in ("H1", "H2"):
            if ln:
                out["%s/%s/beta" % (scope, h)] = (c,)
                out["%s/%s/gamma" % (scope, h)] = (c,)
        if lcc_on and lcc:
            out[scope + "/lcc_embed/lookup_table"] = (hp.nspeakers, c)

    d, e, c = hp.d, hp.e, hp.c
    s = "Text2Mel/TextEnc"; i = 1
    out["%s/embed_%d/lookup_table" % (s, i)] = (len(hp.vocab), e); i += 1
    e_in = e
    if "text_encoder_input" in hp.multispeaker:
        out["%s/embed_%d/lookup_table" % (s, i)] = (hp.nspeakers, hp.speaker_embedding_size); i += 1
        e_in = e + hp.speaker_embedding_size
    conv("%s/C_%d" % (s, i), e_in, 2 * d); i += 1
    conv("%s/C_%d" % (s, i), 2 * d, 2 * d); i += 1
    for _ in range(10):
        hcl("%s/HC_%d" % (s, i), 2 * d, 3); i += 1
    if "text_encoder_towards_end" in hp.multispeaker:
        out["%s/embed_%d/lookup_table" % (s, i)] = (hp.nspeakers, hp.speaker_embedding_size); i += 1
        conv("%s/C_%d" % (s, i), 2 * d + hp.speaker_embedding_size, 2 * d, lcc=False); i += 1
    for _ in range(2):
        hcl("%s/HC_%d" % (s, i), 2 * d, 1); i += 1
    s = "Text2Mel/AudioEnc"; i = 1
    conv("%s/C_%d" % (s, i), hp.n_mels, d); i += 1
    if "audio_encoder_input" in hp.multispeaker:
        out["%s/embed_%d/lookup_table" % (s, i)] = (hp.nspeakers, hp.speaker_embedding_size); i += 1
        conv("%s/C_%d" % (s, i), d + hp.speaker_embedding_size, d, lcc=False); i += 1
    conv("%s/C_%d" % (s, i), d, d); i += 1
    conv("%s/C_%d" % (s, i), d, d); i += 1
    for _ in range(10):
        hcl("%s/HC_%d" % (s, i), d, 3); i += 1
    s = "Text2Mel/AudioDec"; i = 1
    conv("%s/C_%d" % (s, i), 2 * d if getattr(hp, "concatenate_query", True) else d, d, lcc=False); i += 1      # networks.py:317-321: R = [ctx | Q] or ctx alone
    if "audio_decoder_input" in hp.multispeaker:
        out["%s/embed_%d/lookup_table" % (s, i)] = (hp.nspeakers, hp.speaker_embedding_size); i += 1
        conv("%s/C_%d" % (s, i), d + hp.speaker_embedding_size, d, lcc=False); i += 1
    for _ in range(6):
        hcl("%s/HC_%d" % (s, i), d, 3); i += 1
    for _ in range(3):
        conv("%s/C_%d" % (s, i), d, d); i += 1
    conv("%s/C_%d" % (s, i), d, hp.n_mels); i += 1
    lcc_on = False                                   # SSRN passes no lcc / codes to its layers (networks.py:437-537)
    # synthesize() switches hp.norm to 'layer' before it builds SSRNGraph and back afterwards (synthesize.py:513-534):
    # at synthesis the SSRN graph ALWAYS has its LayerNorm variables, whatever hp.norm says for Text2Mel
    ln = True
    s = "SSRN"; i = 1
    conv("%s/C_%d" % (s, i), hp.n_mels, c); i += 1
    if "ssrn_input" in hp.multispeaker:              # networks.py:457-465
        out["%s/embed_%d/lookup_table" % (s, i)] = (hp.nspeakers, hp.speaker_embedding_size); i += 1
        conv("%s/C_%d" % (s, i), c + hp.speaker_embedding_size, c); i += 1
    for _ in range(2):
        hcl("%s/HC_%d" % (s, i), c, 3); i += 1
    for _ in range({4: 2, 8: 3}[hp.r]):
        sc = "%s/D_%d" % (s, i); i += 1
        out[sc + "/conv2d_transpose/kernel"] = (1, 3, c, c)
        out[sc + "/conv2d_transpose/bias"] = (c,)
        # (networks.py:483-486 does not even pass normtype to the transposed convs)
        out[sc + "/normalize/beta"] = (c,)
        out[sc + "/normalize/gamma"] = (c,)
        for _ in range(2):
            hcl("%s/HC_%d" % (s, i), c, 3); i += 1
    conv("%s/C_%d" % (s, i), c, 2 * c); i += 1
    for _ in range(2):
        hcl("%s/HC_%d" % (s, i), 2 * c, 3); i += 1
    conv("%s/C_%d" % (s, i), 2 * c, hp.full_dim); i += 1
    for _ in range(3):
        conv("%s/C_%d" % (s, i), hp.full_dim, hp.full_dim); i += 1
    return out


def _trunc_normal(rng, shape, std):
    x = rng.standard_normal(shape)
    bad = np.abs(x) > 2.0
    while bad.any():
        x[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(x) > 2.0
    return (x * std).astype(F32)


def random_weights(hp, seed, scopes=("Text2Mel", "SSRN"), attention_gain=1.0):
    """Seeded synthetic weights: conv kernels truncated-normal
    with std sqrt(1.3*2/fan_in) (variance_scaling_initializer defaults,
    modules.py:134,191,249), embeddings TN std 0.1 (modules.py:37); bias ~N(0,0.02),
    gamma ~1+N(0,0.05), beta ~N(0,0.05) so the bias/LN paths are exercised.
    One PCG64 stream per variable, keyed by (seed, crc32(name)), so any subset of
    variables is reproducible independently."""
    import zlib
    W = {}
    for name, shape in variable_shapes(hp).items():
        if not name.startswith(tuple(scopes)):
            continue
        rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))
        if name.endswith("lookup_table"):
            W[name] = _trunc_normal(rng, shape, 0.1)
        elif name.endswith("conv1d/kernel"):
            fan_in = shape[0] * shape[1]
            W[name] = _trunc_normal(rng, shape, np.sqrt(1.3 * 2.0 / fan_in))
        elif name.endswith("conv2d_transpose/kernel"):
            # variance_scaling on (1,3,Cout,Cin): fan_in = 1*3*Cout  [TF-sem shape[:-1] product / ...]
            fan_in = shape[1] * shape[2]
            W[name] = _trunc_normal(rng, shape, np.sqrt(1.3 * 2.0 / fan_in))
        elif name.endswith("bias"):
            W[name] = (rng.standard_normal(shape) * 0.02).astype(F32)
        elif name.endswith("gamma"):
            W[name] = (1.0 + rng.standard_normal(shape) * 0.05).astype(F32)
        elif name.endswith("beta"):
            W[name] = (rng.standard_normal(shape) * 0.05).astype(F32)
        else:
            raise KeyError(name)
    return W


def random_text(hp, B, seed, min_len=None, max_len=None):
    """L (B,max_N) int32: length ~U{min..max}, ids ~U{1..V-1}, id 0 padding."""
    rng = np.random.Generator(np.random.PCG64(seed))
    V = len(hp.vocab)
    max_len = hp.max_N - 1 if max_len is None else max_len
    min_len = max(1, hp.max_N // 2) if min_len is None else min_len
    L = np.zeros((B, hp.max_N), np.int32)
    for b in range(B):
        n = int(rng.integers(min_len, max_len + 1))
        L[b, :n] = rng.integers(1, V, size=n)
    return L
