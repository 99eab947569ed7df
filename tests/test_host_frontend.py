"""Host-side mirror of the reference API (config / transcript / helpers).  CPU only.
Front-end expectations come from the reference's own load_config + load_data run by
tests/golden/make_golden.py (frontend.npz, config_snapshot.json)."""
import os
import pickle
import re

import numpy as np
import pytest

from conftest import GOLDEN, hp_from_snapshot


def test_load_config_executes_python_and_fills_defaults():
    from ophelia_amd.configuration import load_config, CONFIG_DEFAULTS
    hp = load_config(os.path.join(GOLDEN, "cfg_unit.cfg"))
    assert hp.config_name == "cfg_unit"
    assert hp.hop_length == int(22050 * 0.0125) and hp.full_dim == 1025       # computed values
    assert hp.logdir.endswith(os.path.join("work", "cfg_unit", "train"))       # derived from __file__
    assert not hasattr(hp, "os")                                               # imported modules dropped
    for k, v in CONFIG_DEFAULTS.items():                                       # late-added options get defaults
        assert getattr(hp, k) == v
    assert hp.concatenate_query is True and hp.store_synth_features is False
    pickle.loads(pickle.dumps(hp))                                              # picklable (used with process pools)
    with pytest.raises(AssertionError):
        load_config("/nonexistent.cfg")


def test_config_defaults_match_reference_snapshot():
    """Every default the reference's load_config filled into lj_tutorial.cfg equals ours."""
    from ophelia_amd.configuration import CONFIG_DEFAULTS
    hp = hp_from_snapshot("lj_tutorial.cfg")
    for k, v in CONFIG_DEFAULTS.items():
        if k == "num_threads" or "dir" in k:      # set explicitly by that config / path-like keys are not in the snapshot
            continue
        assert getattr(hp, k) == v, k


@pytest.mark.parametrize("cfg", ["lj_tutorial", "lj_test"])
def test_load_data_synthesis_matches_reference(cfg):
    from ophelia_amd.data_load import load_data
    from ophelia_amd.libutil import basename
    g = np.load(os.path.join(GOLDEN, "frontend.npz"))
    hp = hp_from_snapshot(cfg + ".cfg")
    hp.test_transcript = os.path.join(GOLDEN, "test_transcript_%s.csv" % cfg)
    hp.waveforms = "/data/wav"
    ds = load_data(hp, mode="synthesis")
    assert np.array_equal(ds["texts"], g[cfg + "_L"]) and ds["texts"].dtype == np.int32
    assert [basename(p) for p in ds["fpaths"]] == g[cfg + "_bases"].tolist()
    assert ds["text_lengths"] == g[cfg + "_text_lengths"].tolist()


def test_duplicate_vocab_entries_later_index_wins():
    from ophelia_amd.data_load import load_vocab
    hp = hp_from_snapshot("lj_tutorial.cfg")
    c2i, i2c = load_vocab(hp)
    assert c2i["<_START_>"] == 16 and c2i["<_END_>"] == 15 and i2c[1] == "<_END_>"


def test_too_long_utterances_dropped_and_blank_lines_skipped(tmp_path):
    from ophelia_amd.data_load import load_data
    hp = hp_from_snapshot("lj_tutorial.cfg", max_N=5)
    p = tmp_path / "t.csv"
    p.write_text("a||x|aa ae\n\n\nb||y|aa ae ah ao aw ax\nc||z|b|\n", encoding="utf-8")
    hp.test_transcript, hp.waveforms = str(p), "w"
    ds = load_data(hp, mode="synthesis")
    assert [os.path.basename(f) for f in ds["fpaths"]] == ["a.wav", "c.wav"]
    assert ds["texts"].shape == (2, 5) and ds["texts"][0, 2:].tolist() == [0, 0, 0]
    with pytest.raises(NotImplementedError):
        load_data(hp, mode="train")


def test_unknown_phone_exits_like_reference(tmp_path):
    from ophelia_amd.data_load import load_data
    hp = hp_from_snapshot("lj_tutorial.cfg")
    p = tmp_path / "t.csv"
    p.write_text("a||x|aa QQ\n", encoding="utf-8")
    hp.test_transcript, hp.waveforms = str(p), "w"
    with pytest.raises(SystemExit) as e:
        load_data(hp, mode="synthesis")
    assert "Phone QQ not listed in phone set" in str(e.value)


def test_letters_mode_appends_eos():
    from ophelia_amd.data_load import text_normalize
    class hp: vocab = "PE abcdefghijklmnopqrstuvwxyz'.?"
    assert text_normalize(u"Héllo,  W0rld", hp) == "hello w rld"


def test_basename_and_text_lengths():
    from ophelia_amd.libutil import basename
    from ophelia_amd.synthesize import get_text_lengths, split_batch, list2batch
    assert basename("/a/b/LJ001-0001.wav") == "LJ001-0001" and basename("x.tar.gz") == "x.tar" and basename("noext") == "noext"
    L = np.array([[3, 4, 0, 0], [1, 0, 5, 0]])
    assert get_text_lengths(L).tolist() == [2, 1]
    with pytest.raises(IndexError):                       # a row without padding (synthesize.py:245)
        get_text_lengths(np.array([[1, 2, 3]]))
    out = split_batch(np.zeros((2, 10, 3)), [4, 7])
    assert [o.shape for o in out] == [(4, 3), (7, 3)]
    b = list2batch([np.ones((2, 3), np.float32), np.ones((4, 3), np.float32)], 0)
    assert b.shape == (2, 4, 3) and b[0, 2:].sum() == 0


def test_mel2mag_chunking_py2_division():
    from ophelia_amd import synthesize as S
    class FakeEng:
        def __init__(self): self.calls = []
        def ssrn(self, Y): self.calls.append(len(Y)); return np.zeros((len(Y), 4, 2), np.float32)
    class FakeSess:
        def __init__(self): self.e = FakeEng()
        def ensure_ready(self): return self.e
    s = FakeSess()
    Z = S.synth_mel2mag(None, np.zeros((300, 1, 1), np.float32), None, s)
    assert s.e.calls == [150, 150] and Z.shape == (300, 4, 2)
    s = FakeSess(); S.synth_mel2mag(None, np.zeros((16, 1, 1), np.float32), None, s)
    assert s.e.calls == [16]


def test_missing_checkpoint_exits_like_reference(tmp_path):
    from ophelia_amd import architectures as A
    class hp: logdir = str(tmp_path / "train")
    with pytest.raises(SystemExit) as e:
        A.restore_latest_model_parameters(None, hp, "t2m")
    assert re.match(r"No t2m at .*train-t2m\?", str(e.value))
    with pytest.raises(SystemExit) as e:
        A.restore_archived_model_parameters(None, hp, "ssrn", 7)
    assert "No ssrn at" in str(e.value) and "archive/model_epoch_7" in str(e.value)


def test_cli_requires_speaker_for_multispeaker(tmp_path, monkeypatch):
    from ophelia_amd import synthesize as S
    cfg = tmp_path / "ms.cfg"
    cfg.write_text(open(os.path.join(GOLDEN, "cfg_unit.cfg")).read() +
                   "\nmultispeaker=['audio_decoder_input']\nspeaker_list=['<PADDING>','p1']\nnspeakers=2\nspeaker_embedding_size=128\n")
    monkeypatch.setattr("sys.argv", ["synthesize", "-c", str(cfg)])
    with pytest.raises(AssertionError, match="Please specify a speaker"):
        S.main_work()
    monkeypatch.setattr("sys.argv", ["synthesize", "-c", str(cfg), "-speaker", "nobody"])
    with pytest.raises(AssertionError):
        S.main_work()


def test_external_durations_front_end_matches_reference():
    """6th transcript field -> (n, max_T, max_N) hard attention matrices: golden from the reference's own load_data"""
    from ophelia_amd.data_load import load_data, durations_to_hard_attention_matrix
    g = np.load(os.path.join(GOLDEN, "frontend_durations.npz"))
    hp = hp_from_snapshot("ssw10/G1AB_03.cfg")
    hp.test_transcript, hp.waveforms = os.path.join(GOLDEN, "test_transcript_durations.csv"), "w"
    assert hp.use_external_durations
    ds = load_data(hp, mode="synthesis")
    assert np.array_equal(ds["texts"], g["L"]) and ds["text_lengths"] == g["text_lengths"].tolist()
    assert ds["durations"].dtype == np.int32 and np.array_equal(ds["durations"], g["durations"])
    # docstring example of the reference (utils.py:202-207)
    A = durations_to_hard_attention_matrix(np.array([3, 0, 1, 2]))
    assert np.array_equal(A.T, [[1, 1, 1, 0, 0, 0], [0, 0, 0, 0, 0, 0], [0, 0, 0, 1, 0, 0], [0, 0, 0, 0, 1, 1]])


def test_out_of_scope_configs_fail_loudly():
    from ophelia_amd.engine import dims_from_hp
    # what SURVEY.md section 2 marks out of scope -- and nothing else
    for attr, val in (("multispeaker", ["no_such_position"]),
                      ("norm", "batch"), ("merlin_label_dir", "/some/labels"), ("text_encoder_type", "minimal_feedforward"),
                      ("history_type", "fractional_position_in_phone")):
        hp = hp_from_snapshot("lj_tutorial.cfg")
        setattr(hp, attr, val)
        with pytest.raises(NotImplementedError):
            dims_from_hp(hp)


def test_f4_leftovers_are_configurations_not_refusals():
    """hp.concatenate_query / squash_output_* False and the 'ssrn_input' / 'speaker_dependent_phones' positions build a handle
    description (the GPU legs are tests/test_gpu_model.py::test_golden_cases[lj_noconcat | lj_nosquash | vctk_spk_ssrn])"""
    from ophelia_amd import _lib
    from ophelia_amd.engine import dims_from_hp
    assert dims_from_hp(hp_from_snapshot("lj_tutorial.cfg", concatenate_query=False)).flags == _lib.FLAG_NO_CONCAT_QUERY
    assert dims_from_hp(hp_from_snapshot("lj_tutorial.cfg", squash_output_t2m=False)).flags == _lib.FLAG_NO_SQUASH_T2M
    assert dims_from_hp(hp_from_snapshot("lj_tutorial.cfg", squash_output_ssrn=False)).flags == _lib.FLAG_NO_SQUASH_SSRN
    d = dims_from_hp(hp_from_snapshot("vctk_01.cfg", multispeaker=["audio_decoder_input", "ssrn_input"]))
    assert d.flags == _lib.FLAG_SPK_AUDIO_DECODER_INPUT | _lib.FLAG_SPK_SSRN_INPUT and d.nspeakers > 0
    d = dims_from_hp(hp_from_snapshot("nancyplusnick_01.cfg", multispeaker=["learn_channel_contributions", "speaker_dependent_phones"]))
    assert d.flags == _lib.FLAG_LCC               # speaker-dependent phones live in the transcript front-end alone


def test_speaker_dependent_phones_front_end_is_the_reference_one(tmp_path):
    """data_load.py:42-46: the vocabulary becomes [padding] + phone_speaker for every speaker and phone; and the reference's
    load_data(mode='synthesis') on such a configuration raises UnboundLocalError (`speaker` is only bound when speakers are read
    from the transcript, which synthesis never does) -- tests/golden/frontend_sdp.json holds what the reference itself did."""
    import json
    from ophelia_amd.data_load import load_data, load_vocab, phones_normalize
    ref = json.load(open(os.path.join(GOLDEN, "frontend_sdp.json")))
    hp = hp_from_snapshot("lj_tutorial.cfg", multispeaker=["speaker_dependent_phones"])
    hp.speaker_list = ref["speaker_list"]
    c2i, i2c = load_vocab(hp)
    assert c2i == ref["char2idx"] and len(i2c) == ref["n_vocab"]
    assert phones_normalize("aa ae", c2i, speaker_code="spkB") == ["aa_spkB", "ae_spkB"]
    hp.test_transcript, hp.waveforms = os.path.join(GOLDEN, "test_transcript_lj_tutorial.csv"), "w"
    kind, msg = ref["synthesis_load_data_raises"]
    assert kind == "UnboundLocalError"
    with pytest.raises(UnboundLocalError) as ei:
        load_data(hp, mode="synthesis")
    assert str(ei.value) == msg


def test_option_variants_map_to_abi_flags():
    """hp.norm None, the non-monotonic synthesis switch and the speaker-embedding positions are carried as oph_dims.flags"""
    from ophelia_amd import _lib
    from ophelia_amd.engine import dims_from_hp
    assert dims_from_hp(hp_from_snapshot("lj_tutorial.cfg")).flags == 0
    d = dims_from_hp(hp_from_snapshot("project/baseline.cfg"))
    assert d.flags == _lib.FLAG_NORM_NONE | _lib.FLAG_NO_MONOTONIC and d.nspeakers == 0
    d = dims_from_hp(hp_from_snapshot("nancyplusnick_01.cfg"))
    assert d.flags == _lib.FLAG_SPK_TEXT_ENCODER_INPUT | _lib.FLAG_SPK_AUDIO_DECODER_INPUT
    assert d.nspeakers == hp_from_snapshot("nancyplusnick_01.cfg").nspeakers and d.speaker_embedding_size == 128
    d = dims_from_hp(hp_from_snapshot("vctk_02.cfg"))
    assert d.flags == _lib.FLAG_SPK_TEXT_ENCODER_TOWARDS_END | _lib.FLAG_SPK_AUDIO_DECODER_INPUT
    hp = hp_from_snapshot("project/baseline.cfg", max_N=300)
    with pytest.raises(NotImplementedError):
        dims_from_hp(hp)                      # full-key attention keeps one key per lane slot: max_N <= 256


def test_degenerate_utterances_do_not_break_the_wav_batch(tmp_path, monkeypatch):
    """An utterance whose attention reaches the end of the text at step 0 has t_end = 0 -> no spectrogram frames (the
    reference's own report loop crashes on it, see tests/golden/wiring_lj_stop.json).  It must get an empty wav and stay
    out of the Griffin-Lim batch, which needs >= 2 frames per utterance; the others are vocoded as usual."""
    from types import SimpleNamespace
    from ophelia_amd import synthesize as S
    from ophelia_amd import vocoder as V
    calls = []

    class FakeVoc(object):
        def spectrogram2wav_batch(self, mags):
            calls.append([len(m) for m in mags])
            assert all(len(m) >= 2 for m in mags)
            return [np.full(10 * (len(m) - 1), 0.25, np.float32) for m in mags]
    monkeypatch.setattr(V, "_vocoder_for", lambda hp, device: FakeVoc())
    hp = SimpleNamespace(vocoder="griffin_lim", store_synth_features=False, sr=22050)
    mags = [np.zeros((8, 5), np.float32), np.zeros((0, 5), np.float32), np.zeros((1, 5), np.float32), np.zeros((3, 5), np.float32)]
    files = [str(tmp_path / ("u%d.wav" % i)) for i in range(4)]
    S.synth_waves(hp, mags, files, device=0)
    assert calls == [[8, 3]]
    import wave
    n = [wave.open(f).getnframes() for f in files]
    assert n == [70, 0, 0, 20]


def test_more_ranks_than_utterances_is_refused_before_any_collective(monkeypatch, tmp_path):
    """A rank with an empty shard cannot stage a batch; skipping the others' collectives would hang them."""
    from ophelia_amd import synthesize as S
    from ophelia_amd import parallel

    class FakeDist(object):
        def get_rank(self): return 1
        def get_world_size(self): return 4
    monkeypatch.setattr(parallel, "_dist", lambda: FakeDist())
    hp = hp_from_snapshot("lj_tutorial.cfg")
    hp.test_transcript = os.path.join(GOLDEN, "test_transcript_lj_tutorial.csv")
    hp.vocoder = "griffin_lim"
    hp.waveforms = str(tmp_path)                     # (the snapshot fixture leaves the site-specific paths out)
    with pytest.raises(SystemExit, match="4 ranks for 2 utterance"):
        S.synthesize(hp, num_sentences=2, topoutdir=str(tmp_path), weights={})
