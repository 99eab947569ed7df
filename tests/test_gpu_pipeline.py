"""The pipeline around the kernels, on the GPU, through the C ABI: streamed SSRN (chunks of mel frames evaluated while the
decoder is still running) against the one-piece evaluation, a different text per batch with the next text pre-encoded under
the running decode, batches of more than 16 utterances decoded in tiles (with the reference's batch-coupled break,
synthesize.py:225-228), residency between the three session calls, oph_run_host, and a soak of the whole-decode launch's
signalling protocol.  Bit-exact comparisons: every variant runs the same kernels on the same rows."""
import numpy as np
import pytest

from conftest import hp_from_snapshot

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model():
    from oracle import ophelia_oracle as O
    from ophelia_amd.engine import Engine
    hp = hp_from_snapshot("lj_tutorial.cfg", max_T=120)
    W = O.random_weights(hp, 2)
    eng = Engine(hp, device=0)
    eng.load_weights(W)
    yield hp, W, eng, O
    eng.close()


def _texts(O, hp, B, seed, lo, hi):
    L = O.random_text(hp, B, seed, min_len=lo, max_len=hi)
    return L, O.get_text_lengths(L).astype(np.int32)


@pytest.mark.parametrize("prec", [0, 1, 2])
def test_streamed_ssrn_is_the_one_piece_ssrn(model, prec):
    """Chunks of frames go through SSRN as soon as the frames within their receptive field exist; every row must equal the
    row of the evaluation over the whole utterance (same dot products, same order), in every arithmetic flavour."""
    hp, W, eng, O = model
    eng.set_ssrn_precision(prec)
    L, ends = _texts(O, hp, 16, 11, 75, 149)
    c0 = eng.counters()
    eng.stage_text(L, ends)
    assert eng.run_resident(stop_mode=1, run_ssrn=True, pipelined=False) == hp.max_T
    c1 = eng.counters()
    assert c1["chunks_streamed"] > c0["chunks_streamed"], "no SSRN chunk was launched while the decode ran"
    assert c1["loop_decodes"] == c0["loop_decodes"] + 1 and c1["loop_fallbacks"] == c0["loop_fallbacks"]
    Y, _, _ = eng.fetch_mel()
    Z = eng.fetch_mag()
    Z1 = eng.ssrn(np.array(Y))                     # a copy: uploaded, evaluated in one piece
    assert np.array_equal(Z, Z1)
    eng.set_ssrn_precision(2)


def test_streamed_ssrn_with_an_early_stop(model):
    """The reference's break leaves frames after the stop step at zero; SSRN still covers all max_T frames."""
    hp, W, eng, O = model
    eng.set_ssrn_precision(0)
    L, ends = _texts(O, hp, 7, 12, 6, 20)
    K, V = eng.encode_text(L)
    Y, t_ends, al, steps = eng.text2mel(K, V, ends)         # K, V resident; SSRN streams speculatively
    assert steps < hp.max_T and not Y[:, steps:].any()
    Z = eng.ssrn(Y)                                         # resident frames: only the tail is left to do
    Z1 = eng.ssrn(np.array(Y))
    assert np.array_equal(Z, Z1)
    K0, V0 = O.encode_text(hp, W, L)
    Y0, t0, al0 = O.synth_codedtext2mel(hp, W, K0, V0, ends)
    assert t_ends.tolist() == list(t0) and np.abs(Y - Y0).max() < 1e-4 and np.abs(al - al0).max() < 1e-4
    assert np.abs(Z - O.synth_mel2mag(hp, W, Y0)).max() < 1e-3
    eng.set_ssrn_precision(2)


def test_residency_between_the_session_calls(model):
    """K,V / Y handed back unchanged are used from HBM; copies are uploaded: same results, bit for bit."""
    hp, W, eng, O = model
    L, ends = _texts(O, hp, 16, 13, 40, 100)
    K, V = eng.encode_text(L)
    assert not K.flags.writeable and not V.flags.writeable
    a = eng.text2mel(K, V, ends, stop_mode=1)
    b = eng.text2mel(np.array(K), np.array(V), ends, stop_mode=1)
    assert a[3] == b[3] and all(np.array_equal(x, y) for x, y in zip(a[:3], b[:3]))
    Za = eng.ssrn(b[0])
    Zb = eng.ssrn(np.array(b[0]))
    assert np.array_equal(Za, Zb)
    with pytest.raises(ValueError):
        K[0, 0, 0] = 1.0                                    # the resident arrays are read-only ...
    K2 = np.array(K); K2[0, :, :] = 0.0                     # ... a modified copy is simply uploaded
    c = eng.text2mel(K2, V, ends, stop_mode=1)
    assert not np.array_equal(c[0], a[0])


def test_two_texts_pipelined_with_preencode_equal_sequential(model):
    """A different text every batch: the text of batch i+1 is staged while batch i runs and pre-encoded under its decode.
    Results must equal the strictly sequential ones bit for bit, and the pre-encode must actually have been used."""
    hp, W, eng, O = model
    texts = [_texts(O, hp, 16, 21 + k, 75, 149) for k in range(2)]
    ref = []
    for L, ends in texts:                                   # sequential reference, each text staged and run on its own
        eng.stage_text(L, ends)
        eng.run_resident(stop_mode=1, run_ssrn=True, pipelined=False)
        ref.append(eng.fetch_mel() + (eng.fetch_mag(), eng.fetch_kv()))
    assert not np.array_equal(ref[0][0], ref[1][0])
    eng.synchronize()
    c0 = eng.counters()
    eng.stage_text(*texts[0])
    eng.stage_text_next(*texts[1])
    for i in range(6):
        k = i & 1
        assert eng.run_resident(stop_mode=1, run_ssrn=True, pipelined=True) == hp.max_T
        Yk, tk, alk = eng.fetch_mel()
        Zk = eng.fetch_mag()
        Kk, Vk = eng.fetch_kv()
        assert np.array_equal(Yk, ref[k][0]) and np.array_equal(alk, ref[k][2]) and np.array_equal(Zk, ref[k][3]), "batch %d" % i
        assert np.array_equal(Kk, ref[k][4][0]) and np.array_equal(Vk, ref[k][4][1])
        eng.stage_text_next(*texts[k])
    c1 = eng.counters()
    assert c1["preenc_used"] - c0["preenc_used"] == 5, c1          # every batch after the first found its K,V ready
    assert c1["textenc"] - c0["textenc"] == 1 + 6                   # first batch encoded in place, then one pre-encode per run
    eng.run_resident(stop_mode=1, run_ssrn=True, pipelined=False)


def test_run_host_equals_the_resident_run(model):
    hp, W, eng, O = model
    L, ends = _texts(O, hp, 16, 31, 6, 30)
    eng.stage_text(L, ends)
    steps = eng.run_resident(stop_mode=0, run_ssrn=True, pipelined=False)
    Y, t_ends, al = eng.fetch_mel(); Z = eng.fetch_mag(); K, V = eng.fetch_kv()
    eng.stage_text(L, ends)
    out = eng.run_host(stop_mode=0, want_kv=True)
    assert out["steps"] == steps and np.array_equal(out["t_ends"], t_ends)
    for name, want in (("K", K), ("V", V), ("Y", Y), ("alignments", al), ("Z", Z)):
        assert np.array_equal(out[name], want), name


@pytest.mark.parametrize("B", [20, 40])
def test_batches_beyond_16_decode_in_tiles_with_the_batch_coupled_break(model, B):
    """B > 16: tiles of 16 utterances, each on the whole-decode launch; the reference breaks when the LAST utterance of the
    whole batch has ended (synthesize.py:225-228), so tiles that finish early are resumed to the batch's stop step."""
    hp, W, eng, O = model
    eng.set_ssrn_precision(0)
    L, ends = _texts(O, hp, B, 41, 6, 36)
    c0 = eng.counters()
    K, V = eng.encode_text(L)
    Y, t_ends, al, steps = eng.text2mel(K, V, ends)
    Z = eng.ssrn(Y)
    c1 = eng.counters()
    ntiles = (B + 15) // 16
    # every tile, and every resumed tile, on the whole-decode launch
    assert c1["loop_decodes"] - c0["loop_decodes"] == ntiles + (c1["tile_resumes"] - c0["tile_resumes"]) and c1["loop_fallbacks"] == c0["loop_fallbacks"]
    K0, V0 = O.encode_text(hp, W, L)
    Y0, t0, al0 = O.synth_codedtext2mel(hp, W, K0, V0, ends)          # the whole batch in one loop, as the reference runs it
    assert t_ends.tolist() == list(t0)
    assert steps == max(t0) + 1 < hp.max_T
    assert c1["tile_resumes"] > c0["tile_resumes"], "this text should make at least one tile stop before the batch does"
    assert np.abs(Y - Y0).max() < 1e-4 and np.abs(al - al0).max() < 1e-4
    assert not Y[:, steps:].any() and not al[:, :, steps:].any()
    assert np.abs(Z - O.synth_mel2mag(hp, W, Y0)).max() < 1e-3
    # fixed length: every tile equals the same 16 utterances decoded as their own batch
    Yf, _, alf, _ = eng.text2mel(K, V, ends, stop_mode=1)
    for j in range(ntiles):
        sl = slice(16 * j, min(B, 16 * j + 16))
        Yj, _, alj, _ = eng.text2mel(np.array(K[sl]), np.array(V[sl]), ends[sl], stop_mode=1)
        assert np.array_equal(Yf[sl], Yj) and np.array_equal(alf[sl], alj)
    eng.set_ssrn_precision(2)


def test_soak_of_the_decode_protocol(model):
    """>= 500 decodes on one handle -- the reference's early stop at varying steps, fixed length, ragged batches, pipelined
    resident batches with a different text each -- every one bitwise equal to the first decode of the same inputs.  A lost
    signal or a stale cone row shows up as a mismatch, a protocol dead-lock as the bounded-spin error."""
    hp, W, eng, O = model
    cases = []
    for i, (B, lo, hi) in enumerate([(16, 6, 30), (16, 75, 149), (5, 10, 40), (11, 4, 12), (16, 20, 60)]):
        L, ends = _texts(O, hp, B, 100 + i, lo, hi)
        K, V = eng.encode_text(L)
        K, V = np.array(K), np.array(V)
        ref = {sm: eng.text2mel(K, V, ends, stop_mode=sm) for sm in (0, 1)}
        cases.append((L, ends, K, V, ref))
    assert len({c[4][0][3] for c in cases}) >= 3, "the early-stop cases should stop at different steps"
    n = 0
    c0 = eng.counters()
    for r in range(100):
        for ci, (L, ends, K, V, ref) in enumerate(cases):
            sm = (r + ci) & 1
            Y, t_ends, al, steps = eng.text2mel(K, V, ends, stop_mode=sm)
            Y0, t0_, al0, s0 = ref[sm]
            assert steps == s0 and np.array_equal(t_ends, t0_), (r, ci, sm, steps, s0)
            assert np.array_equal(Y, Y0) and np.array_equal(al, al0), "decode %d of case %d differs from its first run" % (r, ci)
            n += 1
        if r % 10 == 9:                                  # pipelined resident batches in between (streamed SSRN + pre-encode)
            eng.stage_text(cases[1][0], cases[1][1])
            eng.stage_text_next(cases[4][0], cases[4][1])
            for k in range(4):
                assert eng.run_resident(stop_mode=1, run_ssrn=True, pipelined=True) == hp.max_T
                Yp, _, alp = eng.fetch_mel()
                want = cases[1 if k % 2 == 0 else 4][4][1]
                assert np.array_equal(Yp, want[0]) and np.array_equal(alp, want[2]), "pipelined batch %d of round %d" % (k, r)
                eng.stage_text_next(*(cases[1][:2] if k % 2 == 0 else cases[4][:2]))
                n += 1
            eng.synchronize()
    c1 = eng.counters()
    assert n >= 500 and c1["loop_fallbacks"] == c0["loop_fallbacks"]
    print("soak: %d decodes identical to their first run" % n)


def test_resumed_tiles_redo_the_ssrn_rows_that_saw_their_tail(model):
    """A tile that stopped early and is resumed to the batch's stop step gets new frames after its stop: the SSRN chunks that
    were streamed while it first ran (and saw zeros there) must be redone.  Forced here with a tiny chunk size so that
    chunks do stream before the early stops (child process: a fresh set of streams)."""
    import os, subprocess, sys
    code = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from conftest import hp_from_snapshot
from oracle import ophelia_oracle as O
from ophelia_amd.engine import Engine
hp = hp_from_snapshot("lj_tutorial.cfg", max_T=120)
W = O.random_weights(hp, 2)
L = O.random_text(hp, 24, 41, min_len=6, max_len=36); ends = O.get_text_lengths(L).astype(np.int32)
eng = Engine(hp, device=0, options={"SSRN_CHUNK": 4}); eng.load_weights(W); eng.set_ssrn_precision(0)
K, V = eng.encode_text(L)
Y, t_ends, al, steps = eng.text2mel(K, V, ends)
c = eng.counters()
assert c["tile_resumes"] >= 1 and c["chunks_streamed"] >= 1, c
Z = eng.ssrn(Y)
Z1 = eng.ssrn(np.array(Y))
assert np.array_equal(Z, Z1), float(np.abs(Z - Z1).max())
print("ok", steps, c)
"""
    env = dict(os.environ)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code, root], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-3000:]


def test_a_small_batch_first_then_a_full_one_on_the_same_handle(model):
    """The batched nets' workspaces are sized when the decode state is built; a 5-utterance batch and a 16-utterance one
    share a tile count, so the state is not rebuilt in between -- the workspaces must fit the larger one all the same
    (regression: they were sized for the first batch, and the full batch's SSRN overran them)."""
    from ophelia_amd.engine import Engine
    hp, W, eng, O = model
    L5, e5 = _texts(O, hp, 5, 61, 6, 20)
    L16, e16 = _texts(O, hp, 16, 62, 75, 149)
    eng.stage_text(L16, e16)
    eng.run_resident(stop_mode=1, run_ssrn=True, pipelined=False)
    want = eng.fetch_mel() + (eng.fetch_mag(),)
    fresh = Engine(hp, device=0)
    try:
        fresh.load_weights(W)
        K, V = fresh.encode_text(L5)
        Y, _, _, _ = fresh.text2mel(K, V, e5)
        fresh.ssrn(Y)
        fresh.stage_text(L16, e16)
        fresh.run_resident(stop_mode=1, run_ssrn=True, pipelined=False)
        got = fresh.fetch_mel() + (fresh.fetch_mag(),)
    finally:
        fresh.close()
    assert all(np.array_equal(a, b) for a, b in zip(got, want))


@pytest.mark.parametrize("cfg,over,B,chunk", [("lj_tutorial.cfg", dict(max_T=97), 9, 7), ("lj_tutorial.cfg", dict(max_T=64, r=8), 16, 16),
                                              ("vctk_01.cfg", dict(max_T=100), 8, 23)])
def test_streamed_ssrn_for_other_geometries_and_chunk_sizes(cfg, over, B, chunk):
    """SSRN's margins are derived from the layer list (three up-sampling stages for r = 8, odd lengths, chunk sizes that do
    not divide them, a multispeaker model): the streamed rows equal the one-piece evaluation bit for bit."""
    from oracle import ophelia_oracle as O
    from ophelia_amd.engine import Engine
    hp = hp_from_snapshot(cfg, **over)
    W = O.random_weights(hp, 5)
    eng = Engine(hp, device=0)
    try:
        eng.load_weights(W)
        eng.set_streaming(chunk)
        L = O.random_text(hp, B, 71, min_len=min(40, hp.max_N - 2), max_len=hp.max_N - 1)
        ends = O.get_text_lengths(L).astype(np.int32)
        spk = np.random.default_rng(5).integers(1, 100, size=(B, 1)) if hp.multispeaker else None
        for prec in (0, 2):
            eng.set_ssrn_precision(prec)
            c0 = eng.counters()
            K, V = eng.encode_text(L, spk)
            Y, t_ends, al, steps = eng.text2mel(K, V, ends, spk, stop_mode=1)
            assert eng.counters()["chunks_streamed"] - c0["chunks_streamed"] >= 1        # (one chunk in flight at a time: not every boundary is used)
            Z = eng.ssrn(Y)
            Z1 = eng.ssrn(np.array(Y))
            assert Z.shape == (B, hp.max_T * hp.r, hp.full_dim) and np.array_equal(Z, Z1), prec
    finally:
        eng.close()


def test_resumed_decode_then_ssrn_returns_every_row(model):
    """text2mel stops early and streams its SSRN chunks to the host; the decode is then resumed to a later
    (global, multi-GPU) stop step -- chunks streamed during the resume have no host destination; ssrn(Y) on the fetched frames
    must still return every magnitude row (copied frontier tracked apart from the computed one)."""
    hp, W, eng, O = model
    eng.set_ssrn_precision(0)
    eng.set_streaming(8)                                   # small chunks: several are streamed during the short resume
    try:
        L, ends = _texts(O, hp, 9, 21, 6, 14)
        K, V = eng.encode_text(L)
        Y, t_ends, al, steps = eng.text2mel(K, V, ends)
        assert steps < hp.max_T - 60
        later = steps + 56
        assert eng.decode_steps(steps, later, 1) == later          # the shard resumes to the batch's stop step
        Y2, _, _ = eng.fetch_mel()
        assert np.array_equal(Y2[:, :steps], Y[:, :steps]) and Y2[:, steps:later].any() and not Y2[:, later:].any()
        Z = eng.ssrn(Y2)                                        # resident: picks up what was streamed, computes / copies the rest
        Z1 = eng.ssrn(np.array(Y2))                             # one piece from the uploaded copy
        assert np.array_equal(Z, Z1)
    finally:
        eng.set_streaming(40)
        eng.set_ssrn_precision(2)


def test_unfetched_magnitudes_of_an_earlier_batch_are_never_adopted(model):
    """text2mel() whose magnitudes nobody fetched, then another batch through the resident pipeline, then
    ssrn() on ITS fetched frames: the result must be that batch's magnitudes, not the stale streaming buffer of the first."""
    hp, W, eng, O = model
    La, ea = _texts(O, hp, 16, 31, 40, 100)
    Lb, eb = _texts(O, hp, 16, 32, 40, 100)
    K, V = eng.encode_text(La)
    eng.text2mel(K, V, ea, stop_mode=1)                        # streams into its own pinned buffer; Z never asked for
    eng.stage_text(Lb, eb)
    assert eng.run_resident(stop_mode=1, run_ssrn=True, pipelined=False) == hp.max_T
    Zb = eng.fetch_mag()
    Y, _, _ = eng.fetch_mel()
    Z = eng.ssrn(Y)
    assert np.array_equal(Z, Zb)
    assert np.array_equal(Z, eng.ssrn(np.array(Y)))


def test_writable_results_when_residency_is_off():
    """Engine(resident_results=False): plain writable arrays out, everything uploaded -- the reference's data flow; same numbers."""
    from oracle import ophelia_oracle as O
    from ophelia_amd.engine import Engine
    hp = hp_from_snapshot("lj_tutorial.cfg", max_N=30, max_T=24)
    W = O.random_weights(hp, 2)
    L, ends = _texts(O, hp, 5, 41, 8, 20)
    res = []
    for flag in (True, False):
        eng = Engine(hp, device=0, resident_results=flag)
        eng.load_weights(W)
        K, V = eng.encode_text(L)
        Y, t_ends, al, steps = eng.text2mel(K, V, ends, stop_mode=1)
        assert K.flags.writeable == (not flag) and Y.flags.writeable == (not flag)
        if not flag:
            Y[0, -1] += 0.0                                    # in-place post-processing works on the writable arrays
        res.append((np.array(K), np.array(Y), np.array(eng.ssrn(Y))))
        eng.close()
    assert all(np.array_equal(a, b) for a, b in zip(res[0], res[1]))
