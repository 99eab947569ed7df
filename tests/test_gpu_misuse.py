"""Calls in the wrong order and with bad arguments, straight at the C ABI: each must come back with an error code and a message
(never a crash, a hang or a silent success), and the handle must serve a normal batch afterwards exactly as before -- the
reference's session would raise a Python exception at the same places (a feed without its placeholder, a fetch before a run)."""
import ctypes as C

import numpy as np
import pytest

from conftest import hp_from_snapshot

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model():
    from oracle import ophelia_oracle as O
    from ophelia_amd.engine import Engine
    hp = hp_from_snapshot("lj_tutorial.cfg", max_T=64)
    W = O.random_weights(hp, 9)
    eng = Engine(hp, device=0)
    eng.load_weights(W)
    yield hp, eng, O
    eng.close()


def _batch(eng, O, hp, B=5, seed=4):
    L = O.random_text(hp, B, seed, min_len=30, max_len=90)
    ends = O.get_text_lengths(L).astype(np.int32)
    K, V = eng.encode_text(L)
    Y, t_ends, al, steps = eng.text2mel(np.array(K), np.array(V), ends, stop_mode=1)
    return np.array(Y), np.array(al), np.array(eng.ssrn(np.array(Y)))


def test_wrong_order_and_bad_arguments_are_errors_not_crashes(model):
    from ophelia_amd import _lib
    hp, eng, O = model
    lib, h = eng.lib, eng._h
    d = eng.dims
    before = _batch(eng, O, hp)

    def err(rc, what):
        assert rc != 0, "%s: returned success" % what
        msg = lib.oph_last_error(h)
        assert msg and len(msg) > 0, "%s: no message" % what

    B = 4
    L = np.ascontiguousarray(O.random_text(hp, B, 1, min_len=20, max_len=60), dtype=np.int32)
    ends = np.ascontiguousarray(O.get_text_lengths(L), dtype=np.int32)
    Y = np.zeros((B, d.max_T, d.n_mels), np.float32)
    Z = np.zeros((B, d.max_T * d.r, d.full_dim), np.float32)
    K = np.zeros((B, d.max_N, d.d), np.float32)
    te = np.zeros((B,), np.int32)
    al = np.zeros((B, d.max_N, d.max_T), np.float32)
    steps = C.c_int32()
    fp, ip = _lib.fptr, _lib.iptr

    # a freshly staged text: no oph_encode_text result and no decoded frames in HBM that NULL operands could refer to
    eng.stage_text(L, ends)
    err(lib.oph_text2mel(h, None, None, ip(ends), None, B, 1, fp(Y), ip(te), fp(al), C.byref(steps)), "text2mel(NULL K,V) with no resident K,V")
    err(lib.oph_ssrn(h, None, B, d.max_T, fp(Z)), "ssrn(NULL Y) with no resident frames")
    eng.encode_text(L)
    err(lib.oph_text2mel(h, None, None, ip(ends), None, B + 1, 1, fp(Y), ip(te), fp(al), C.byref(steps)), "text2mel(NULL K,V) for another batch size")
    err(lib.oph_ssrn(h, None, B, d.max_T, fp(Z)), "ssrn(NULL Y) after an encode only")
    # batch sizes and lengths out of range
    err(lib.oph_encode_text(h, ip(L), None, 0, fp(K), fp(K)), "encode_text B = 0")
    err(lib.oph_encode_text(h, ip(L), None, -3, fp(K), fp(K)), "encode_text B < 0")
    err(lib.oph_ssrn(h, fp(Y), B, d.max_T + 1, fp(Z)), "ssrn T > max_T")
    err(lib.oph_ssrn(h, fp(Y), 0, d.max_T, fp(Z)), "ssrn B = 0")
    err(lib.oph_stage_text(h, ip(L), ip(ends), None, 0), "stage_text B = 0")
    # NULL where data is required
    err(lib.oph_encode_text(h, None, None, B, fp(K), fp(K)), "encode_text L = NULL")
    err(lib.oph_stage_text(h, None, ip(ends), None, B), "stage_text L = NULL")
    # token ids outside the vocabulary, ends outside the text
    bad = L.copy(); bad[1, 3] = 10 ** 6
    err(lib.oph_encode_text(h, ip(bad), None, B, fp(K), fp(K)), "token id out of range")
    bad_ends = ends.copy(); bad_ends[0] = d.max_N + 5
    err(lib.oph_stage_text(h, ip(L), ip(bad_ends), None, B), "end position beyond max_N")
    # the staged / resident entry points before anything is staged on this geometry, and bad step ranges
    eng.stage_text(L, ends)
    err(lib.oph_decode_steps(h, 5, 3, 1, C.byref(steps)), "decode_steps with t_end < t_begin")
    err(lib.oph_decode_steps(h, -1, 3, 1, C.byref(steps)), "decode_steps with t_begin < 0")
    err(lib.oph_stage_text_next(h, ip(L), ip(ends), None, B + 1), "stage_text_next with another batch size")
    err(lib.oph_run_resident(h, 7, 1, C.byref(steps)), "run_resident with an unknown stop mode")
    # unknown switches
    err(lib.oph_set_precision(h, 9, 0), "set_precision: unknown network")
    err(lib.oph_set_precision(h, 0, 17), "set_precision: unknown arithmetic")
    err(lib.oph_set_ssrn_precision(h, -2), "set_ssrn_precision(-2)")
    # weights cannot change under a finalised handle
    w = np.zeros((hp.c,), np.float32)
    err(lib.oph_set_weight(h, b"SSRN/C_1/conv1d/bias", fp(w), (C.c_int64 * 1)(hp.c), 1), "set_weight after finalize")

    # a NULL handle is an error everywhere, not a crash
    for rc in (lib.oph_encode_text(None, ip(L), None, B, fp(K), fp(K)), lib.oph_ssrn(None, fp(Y), B, d.max_T, fp(Z)),
               lib.oph_stage_text(None, ip(L), ip(ends), None, B), lib.oph_run_resident(None, 1, 1, C.byref(steps)),
               lib.oph_decode_steps(None, 0, 3, 1, C.byref(steps)), lib.oph_fetch_mag(None, fp(Z)), lib.oph_synchronize(None),
               lib.oph_set_streaming(None, 1), lib.oph_set_precision(None, 0, 0), lib.oph_timer_start(None), lib.oph_finalize_weights(None)):
        assert rc != 0
    # run_ssrn codes, step ranges beyond max_T
    err(lib.oph_run_resident(h, 1, 5, C.byref(steps)), "run_resident with an unknown SSRN mode")
    err(lib.oph_decode_steps(h, 0, d.max_T + 1, 1, C.byref(steps)), "decode_steps beyond max_T")

    after = _batch(eng, O, hp)
    assert all(np.array_equal(a, b) for a, b in zip(before, after)), "the handle does not serve the same batch the same way after the errors"


def test_inherited_environment_cannot_change_a_result_or_a_launch_path(model, monkeypatch):
    """OPH_SKIP_CONE ("results are wrong") and OPH_LOOP_ALONE ("mel unused") used to be read by every production
    build.  Now the launch paths are options of oph_create_opts and the ablation names exist only in -DOPH_ABLATE builds: with the
    old variables in the environment a fresh handle computes the same bits through the same launches, and asking for an ablation
    option explicitly is an error."""
    from ophelia_amd.engine import Engine
    from ophelia_amd import _lib
    hp, eng, O = model
    W = O.random_weights(hp, 9)
    want = _batch(eng, O, hp, B=16, seed=11)
    for k, v in (("OPH_SKIP_CONE", "1"), ("OPH_LOOP_ALONE", "1"), ("OPH_LOOP_DBG", "32"), ("OPH_DECODE", "layers"), ("OPH_NO_CHAIN", "1"),
                 ("OPH_SSRN_PREC", "1"), ("OPH_CONE_PREC", "1"), ("OPH_NO_CU_MASK", "1"), ("OPH_SSRN_CHUNK", "3")):
        monkeypatch.setenv(k, v)
    e2 = Engine(hp, device=0)
    try:
        e2.load_weights(W)
        c0 = e2.counters()
        got = _batch(e2, O, hp, B=16, seed=11)
        c1 = e2.counters()
        for a, b, what in zip(got, want, ("Y", "alignments", "Z")):
            assert np.array_equal(a, b), "%s differs under an inherited OPH_* environment" % what
        assert c1["loop_decodes"] - c0["loop_decodes"] == 1 and c1["masked_streams"] == 1      # the default launch path ran
    finally:
        e2.close()
    for bad in ("SKIP_CONE=1", "LOOP_ALONE", "LOOP_DBG=32", "NO_SUCH_OPTION=1"):
        with pytest.raises(_lib.OpheliaHipError) as ei:
            Engine(hp, device=0, options=bad)
        assert bad.split("=")[0] in str(ei.value)
