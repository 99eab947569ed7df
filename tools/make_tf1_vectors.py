#!/usr/bin/env python
"""Pins for the five TensorFlow-1.x primitives the synthesis path stands on, produced BY TensorFlow -- for a maintainer who has the
reference's environment (tensorflow-gpu==1.12.0, requirements.txt:44; any TF 1.x with tf.contrib works).  This repository's oracle
restates those primitives from their documented behaviour and checks them against torch (tests/test_oracle_primitives.py); no
TensorFlow exists in the build container, so the oracle's reading of TF has never been executed against TF itself (DESIGN.md
section 2: "parity unpinned").  This script closes that gap in one command:

    python tools/make_tf1_vectors.py            # writes tests/golden/tf1_vectors.npz and tests/golden/tf1_ckpt/

and `pytest tests/test_tf1_vectors.py` then compares the oracle (CPU) and, with -m gpu, the HIP operators with what TF computed.
Until the files exist those tests skip with exactly that reason.

What is recorded (seeded inputs, the TF outputs; call sites in the reference in brackets):
  layer_norm        tf.contrib.layers.layer_norm(x, begin_norm_axis=-1)                       [modules.py:65]     eps, biased variance
  conv1d_same_*     tf.layers.conv1d(padding="same", dilation_rate=r), r in 1, 3, 9, 27        [modules.py:132-136] SAME split of dilated padding
  conv1d_causal_*   tf.pad(x, [[0,0],[(k-1)*r,0],[0,0]]) + tf.layers.conv1d("valid")           [modules.py:123-127]
  conv2d_transpose  tf.layers.conv2d_transpose(kernel (1,3), strides (1,2), padding="same")    [modules.py:243-250] output alignment, length 2T
  attention         masked softmax + argmax exactly as networks.Attention builds them           [networks.py:300-316] -2**32+1 fill, argmax ties,
                    for prev_max values including the ones where sequence_mask gets a length <= 0                    sequence_mask(<= 0)
  checkpoint        a tf.train.Saver checkpoint of a few Text2Mel/SSRN-scoped variables with Adam slots and global_step,
                    written the way train.py:296-305 writes them                               [synthesize.py:302-330] tensor-bundle reader
Nothing of the reference's source is imported or copied: the graph snippets below are written from its call signatures.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    try:
        import tensorflow as tf
    except ImportError:
        sys.exit("this script needs TensorFlow 1.x (the reference pins tensorflow-gpu==1.12.0)")
    if not tf.__version__.startswith("1."):
        sys.exit("TensorFlow %s found; the reference's primitives are TF 1.x (tf.contrib.layers.layer_norm, tf.layers.*)" % tf.__version__)
    rng = np.random.RandomState(20250930)
    out = {"tf_version": np.array(tf.__version__)}

    def run(build, feeds, var_values=None):
        """build(placeholders...) -> tensor(s) in a fresh graph; variables are set to var_values[name] (recorded too)."""
        g = tf.Graph()
        with g.as_default():
            phs = [tf.placeholder(tf.as_dtype(v.dtype), v.shape) for v in feeds]
            fetch = build(*phs)
            with tf.Session(graph=g) as sess:
                sess.run(tf.global_variables_initializer())
                for v in tf.global_variables():
                    if var_values is not None and v.op.name in var_values:
                        v.load(var_values[v.op.name], sess)
                vals = {v.op.name: sess.run(v) for v in tf.global_variables()}
                res = sess.run(fetch, dict(zip(phs, feeds)))
        return res, vals

    # ---- layer_norm
    x = (rng.randn(3, 7, 80) * 2).astype(np.float32)
    gam = (1 + 0.1 * rng.randn(80)).astype(np.float32); bet = (0.1 * rng.randn(80)).astype(np.float32)
    y, _ = run(lambda p: tf.contrib.layers.layer_norm(p, begin_norm_axis=-1, scope="normalize"), [x],
               {"normalize/gamma": gam, "normalize/beta": bet})
    out.update(ln_x=x, ln_gamma=gam, ln_beta=bet, ln_y=y)
    # a row with (nearly) zero variance: where eps matters
    xc = np.full((1, 2, 80), 0.37, np.float32); xc[0, 1, 3] += 1e-6
    yc, _ = run(lambda p: tf.contrib.layers.layer_norm(p, begin_norm_axis=-1, scope="normalize"), [xc],
                {"normalize/gamma": gam, "normalize/beta": bet})
    out.update(ln_const_x=xc, ln_const_y=yc)

    # ---- conv1d, SAME and CAUSAL, dilated
    B, T, Cin, Cout = 2, 61, 24, 40
    xin = rng.randn(B, T, Cin).astype(np.float32)
    for size, rate in [(1, 1), (3, 1), (3, 3), (3, 9), (3, 27)]:
        k = (0.1 * rng.randn(size, Cin, Cout)).astype(np.float32); b = (0.1 * rng.randn(Cout)).astype(np.float32)
        ys, _ = run(lambda p: tf.layers.conv1d(inputs=p, filters=Cout, kernel_size=size, dilation_rate=rate, padding="same", use_bias=True, name="conv1d"),
                    [xin], {"conv1d/kernel": k, "conv1d/bias": b})
        yc_, _ = run(lambda p: tf.layers.conv1d(inputs=tf.pad(p, [[0, 0], [(size - 1) * rate, 0], [0, 0]]), filters=Cout, kernel_size=size,
                                               dilation_rate=rate, padding="valid", use_bias=True, name="conv1d"),
                     [xin], {"conv1d/kernel": k, "conv1d/bias": b})
        tag = "k%d_r%d" % (size, rate)
        out.update({"conv_%s_kernel" % tag: k, "conv_%s_bias" % tag: b, "conv_same_%s_y" % tag: ys, "conv_causal_%s_y" % tag: yc_})
    out["conv_x"] = xin

    # ---- conv2d_transpose as modules.conv1d_transpose calls it: (B,T,C) -> expand_dims(1) -> kernel (1,3), strides (1,2), SAME -> squeeze
    for T2 in (1, 2, 11):
        C = 16
        xt = rng.randn(2, T2, C).astype(np.float32)
        kt = (0.1 * rng.randn(1, 3, C, C)).astype(np.float32); bt = (0.1 * rng.randn(C)).astype(np.float32)
        yt, _ = run(lambda p: tf.squeeze(tf.layers.conv2d_transpose(tf.expand_dims(p, 1), filters=C, kernel_size=(1, 3), strides=(1, 2), padding="same",
                                                                     activation=None, use_bias=True, name="conv2d_transpose"), 1),
                    [xt], {"conv2d_transpose/kernel": kt, "conv2d_transpose/bias": bt})
        out.update({"convt_T%d_x" % T2: xt, "convt_T%d_kernel" % T2: kt, "convt_T%d_bias" % T2: bt, "convt_T%d_y" % T2: yt})

    # ---- the attention's mask / softmax / argmax, for prev_max values across the range (networks.py:300-316)
    d, N, Tq, win = 16, 10, 4, 3
    Q = rng.randn(5, Tq, d).astype(np.float32); K = rng.randn(5, N, d).astype(np.float32); V = rng.randn(5, N, d).astype(np.float32)
    K[4, 3] = K[4, 2]                      # two equal logits inside a window: which one does argmax take?
    prev = np.array([0, 2, 7, 8, 2], np.int32)       # N - win - p > 0 for p < 7; = 0 at 7; < 0 at 8 (sequence_mask of a length <= 0)

    def attn(q, k, v, p):
        A = tf.matmul(q, k, transpose_b=True) * tf.rsqrt(tf.to_float(d))
        key_masks = tf.sequence_mask(p, N)
        reverse_masks = tf.sequence_mask(N - win - p, N)[:, ::-1]
        masks = tf.logical_or(key_masks, reverse_masks)
        masks = tf.tile(tf.expand_dims(masks, 1), [1, Tq, 1])
        paddings = tf.ones_like(A) * (-2 ** 32 + 1)
        A = tf.where(tf.equal(masks, False), A, paddings)
        A = tf.nn.softmax(A)
        mx = tf.argmax(A, -1)
        R = tf.concat((tf.matmul(A, v), q), -1)
        return A, mx, R
    (A, mx, R), _ = run(attn, [Q, K, V, prev])
    out.update(att_Q=Q, att_K=K, att_V=V, att_prev=prev, att_A=A, att_argmax=mx.astype(np.int64), att_R=R,
               att_dims=np.array([d, N, Tq, win], np.int32))

    np.savez_compressed(os.path.join(GOLD, "tf1_vectors.npz"), **out)

    # ---- a Saver checkpoint with Adam slots, named like the reference's variables (train.py:296-305 saves with tf.train.Saver)
    ck = os.path.join(GOLD, "tf1_ckpt")
    os.makedirs(ck, exist_ok=True)
    g = tf.Graph()
    expect = {}
    with g.as_default():
        tf.set_random_seed(7)
        with tf.variable_scope("Text2Mel"):
            with tf.variable_scope("TextEnc"):
                e = tf.get_variable("embed_1/lookup_table", [9, 8], tf.float32, tf.truncated_normal_initializer(stddev=0.1))
                w = tf.get_variable("C_2/conv1d/kernel", [1, 8, 16], tf.float32, tf.truncated_normal_initializer(stddev=0.1))
                b = tf.get_variable("C_2/conv1d/bias", [16], tf.float32, tf.zeros_initializer())
        with tf.variable_scope("SSRN"):
            wt = tf.get_variable("D_4/conv2d_transpose/kernel", [1, 3, 6, 6], tf.float32, tf.truncated_normal_initializer(stddev=0.1))
        gs = tf.Variable(0, name="global_step", trainable=False)
        loss = tf.reduce_sum(tf.square(e)) + tf.reduce_sum(tf.square(w)) + tf.reduce_sum(b) + tf.reduce_sum(tf.square(wt))
        train = tf.train.AdamOptimizer(1e-3).minimize(loss, global_step=gs)
        saver = tf.train.Saver(max_to_keep=2)
        with tf.Session(graph=g) as sess:
            sess.run(tf.global_variables_initializer())
            sess.run(train)
            saver.save(sess, os.path.join(ck, "model_epoch_3"))
            sess.run(train)
            saver.save(sess, os.path.join(ck, "model_epoch_7"))
            for v in tf.trainable_variables():
                expect[v.op.name] = sess.run(v)
    np.savez_compressed(os.path.join(GOLD, "tf1_ckpt_expected.npz"), **expect)
    json.dump({"latest": "model_epoch_7", "tf_version": tf.__version__}, open(os.path.join(ck, "made_by.json"), "w"))
    print("wrote", os.path.join(GOLD, "tf1_vectors.npz"), "and", ck)


if __name__ == "__main__":
    main()
