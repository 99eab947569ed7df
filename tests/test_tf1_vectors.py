"""The oracle's (and, with -m gpu, the HIP operators') reading of the five TensorFlow-1.x primitives against vectors computed BY
TensorFlow -- tests/golden/tf1_vectors.npz and tests/golden/tf1_ckpt/, written by `python tools/make_tf1_vectors.py` on a machine
with the reference's environment (tensorflow-gpu==1.12.0).  No TensorFlow exists where this repository is built, so until someone
runs that script and commits its output these tests SKIP, and parity at the TF-primitive level stays "unpinned" (DESIGN.md
section 2).  When the files are there, this is the check that turns it green: modules.py:65, 132-136, 189-193, 243-250,
networks.py:300-316, synthesize.py:302-330."""
import os

import numpy as np
import pytest

from oracle import ophelia_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
VEC = os.path.join(GOLD, "tf1_vectors.npz")
CKPT = os.path.join(GOLD, "tf1_ckpt")
need_vec = pytest.mark.skipif(not os.path.isfile(VEC), reason="tests/golden/tf1_vectors.npz not present: run tools/make_tf1_vectors.py under TensorFlow 1.x")
need_ckpt = pytest.mark.skipif(not os.path.isfile(os.path.join(CKPT, "checkpoint")), reason="tests/golden/tf1_ckpt/ not present: run tools/make_tf1_vectors.py under TensorFlow 1.x")
TOL = 2e-5
CONV_CASES = [(1, 1), (3, 1), (3, 3), (3, 9), (3, 27)]


def _v():
    return np.load(VEC, allow_pickle=False)


def _attn_hp(v):
    class hp: pass
    hp.d, hp.max_N, _, hp.attention_win_size = [int(x) for x in v["att_dims"]]
    hp.concatenate_query = True
    return hp


@need_vec
def test_oracle_layer_norm_matches_tensorflow():
    v = _v()
    assert np.abs(O.normalize(v["ln_x"], v["ln_gamma"], v["ln_beta"]) - v["ln_y"]).max() < TOL
    # the nearly constant rows: the epsilon (1e-12 inside the square root) decides these values
    assert np.abs(O.normalize(v["ln_const_x"], v["ln_gamma"], v["ln_beta"]) - v["ln_const_y"]).max() < 1e-3 * max(1.0, float(np.abs(v["ln_const_y"]).max()))


@need_vec
@pytest.mark.parametrize("size,rate", CONV_CASES)
def test_oracle_conv1d_same_and_causal_match_tensorflow(size, rate):
    v = _v()
    tag = "k%d_r%d" % (size, rate)
    k, b = v["conv_%s_kernel" % tag], v["conv_%s_bias" % tag]
    assert np.abs(O._conv_taps(v["conv_x"], k, b, rate, "SAME") - v["conv_same_%s_y" % tag]).max() < TOL
    assert np.abs(O._conv_taps(v["conv_x"], k, b, rate, "CAUSAL") - v["conv_causal_%s_y" % tag]).max() < TOL


@need_vec
@pytest.mark.parametrize("T", [1, 2, 11])
def test_oracle_conv2d_transpose_matches_tensorflow(T):
    v = _v()
    W = {"d/conv2d_transpose/kernel": v["convt_T%d_kernel" % T], "d/conv2d_transpose/bias": v["convt_T%d_bias" % T]}      # no gamma / beta: the raw layer
    got = O.conv1d_transpose(v["convt_T%d_x" % T], W, "d")
    assert got.shape == v["convt_T%d_y" % T].shape
    assert np.abs(got - v["convt_T%d_y" % T]).max() < TOL


@need_vec
def test_oracle_attention_mask_softmax_argmax_match_tensorflow():
    v = _v()
    R, al, mx = O.attention(_attn_hp(v), v["att_Q"], v["att_K"], v["att_V"], v["att_prev"])
    assert np.array_equal(mx, v["att_argmax"])                        # first maximum on ties included (utterance 4)
    assert np.abs(np.transpose(al, (0, 2, 1)) - v["att_A"]).max() < 1e-6
    assert np.abs(R - v["att_R"]).max() < 1e-5


@need_ckpt
def test_checkpoint_reader_reads_a_tensorflow_written_checkpoint():
    from ophelia_amd import tf_checkpoint as TC
    exp = np.load(os.path.join(GOLD, "tf1_ckpt_expected.npz"))
    prefix = TC.latest_checkpoint(CKPT)
    assert prefix is not None and os.path.basename(prefix) == "model_epoch_7"
    for scope in ("Text2Mel/", "SSRN/"):
        got = TC.read_checkpoint(prefix, scope=scope, verify_data=True)
        want = {k: exp[k] for k in exp.files if k.startswith(scope)}
        assert sorted(got) == sorted(want)                            # Adam slots, global_step, beta powers left out
        for k in want:
            assert got[k].dtype == want[k].dtype and np.array_equal(got[k], want[k]), k
    older = TC.read_checkpoint(os.path.join(CKPT, "model_epoch_3"), scope="Text2Mel/")
    assert any(not np.array_equal(older[k], exp[k]) for k in older)   # a different training step: really a second file


# ---- the HIP operators against the same vectors (through the C ABI)
@need_vec
@pytest.mark.gpu
def test_hip_operators_match_tensorflow():
    from ophelia_amd import modules as M
    v = _v()
    W = {"n/gamma": v["ln_gamma"], "n/beta": v["ln_beta"]}
    assert np.abs(M.normalize(v["ln_x"], W, "n") - v["ln_y"]).max() < TOL
    ident = lambda C: {"gamma": np.ones(C, np.float32), "beta": np.zeros(C, np.float32)}
    # conv1d / conv1d_transpose include their LayerNorm in the operator: compare after applying the oracle's LayerNorm to TF's raw output
    for size, rate in CONV_CASES:
        tag = "k%d_r%d" % (size, rate)
        k, b = v["conv_%s_kernel" % tag], v["conv_%s_bias" % tag]
        C = k.shape[2]
        Wc = {"c/conv1d/kernel": k, "c/conv1d/bias": b, "c/normalize/gamma": ident(C)["gamma"], "c/normalize/beta": ident(C)["beta"]}
        for padding, key in (("SAME", "conv_same_%s_y"), ("CAUSAL", "conv_causal_%s_y")):
            ref = O.normalize(v[key % tag], ident(C)["gamma"], ident(C)["beta"])
            got = M.conv1d(v["conv_x"], Wc, "c", size=size, rate=rate, padding=padding)
            assert np.abs(got - ref).max() < TOL, (tag, padding)
    for T in (1, 2, 11):
        k, b = v["convt_T%d_kernel" % T], v["convt_T%d_bias" % T]
        C = k.shape[2]
        Wt = {"d/conv2d_transpose/kernel": k, "d/conv2d_transpose/bias": b, "d/normalize/gamma": ident(C)["gamma"], "d/normalize/beta": ident(C)["beta"]}
        ref = O.normalize(v["convt_T%d_y" % T], ident(C)["gamma"], ident(C)["beta"])
        assert np.abs(M.conv1d_transpose(v["convt_T%d_x" % T], Wt, "d") - ref).max() < TOL
    R, al, mx = M.attention(_attn_hp(v), v["att_Q"], v["att_K"], v["att_V"], v["att_prev"])
    assert np.array_equal(mx, v["att_argmax"])
    assert np.abs(np.transpose(al, (0, 2, 1)) - v["att_A"]).max() < 1e-6 and np.abs(R - v["att_R"]).max() < 1e-5
