# -*- coding: utf-8 -*-
"""
Generates the committed golden fixtures under tests/golden/ by executing the
REFERENCE's own code (read from /root/reference at run time, build container only):

  * wiring + host-loop goldens: the reference's own synthesize.synthesize() -- encode_text,
    synth_codedtext2mel (the max_T-step loop with its early stop / t_ends rules),
    synth_mel2mag (py2 integer-division chunking), get_text_lengths, the per-utterance
    trimming, the output directory / file naming and the CDP / Ain report of
    calculate_CDP_Ain_Aout.getCDP / getAP -- driving the reference's
    architectures.Text2MelGraph / SSRNGraph -> networks.py -> modules.py over the eager
    stand-in tests/golden/tf_standin.py (torch CPU primitives).  tests/golden/ref_host.py
    loads the Python-2 sources in memory (lib2to3 + a py2-division pass) and supplies
    the fake tf.Session.  Reduced max_N / max_T, FULL channel widths.
  * front-end goldens: reference configuration.load_config + data_load.load_data
    (mode='synthesis') on the transcript fixture tests/golden/test_transcript.csv.

Only small outputs + seeds are stored; weights are regenerated from the seed by
oracle.ophelia_oracle.random_weights on both sides.

Usage (from the repo root, in the build container) -- ONE command regenerates every fixture:
    python -B tests/golden/make_golden.py
"""
import io
import json
import os
import sys
import contextlib

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

import numpy as np

import tf_standin as tf
tf.install()
sys.path.append(REF)          # AFTER the stand-in's directory

from oracle import ophelia_oracle as O     # only for seeded weights / text generation

import configuration as ref_configuration   # reference
with contextlib.redirect_stdout(io.StringIO()):
    import architectures as ref_arch         # reference (imports networks, modules, data_load, utils)
    import data_load as ref_data_load        # reference


import ref_host
HOST = ref_host.ReferenceHost(tf, ref_arch)


def run_case(tag, cfg, B, max_N, max_T, wseed, tseed, min_len, max_len, stop, speaker_ix=None, override=None):
    hp = ref_configuration.load_config(os.path.join(REF, "config", cfg))
    hp.max_N, hp.max_T = max_N, max_T
    hp.vocoder = "griffin_lim"
    for k, v in (override or {}).items():           # synthetic variants no shipped config uses (recorded in the json)
        setattr(hp, k, v)
    # synthesize() builds SSRNGraph under hp.norm = 'layer' whatever the config says (synthesize.py:513-534): the
    # variable inventory (and the seeded weights) must be those of that graph
    W = O.random_weights(hp, wseed)
    tf.VARS.clear(); tf.VARS.update(W); tf.REQUESTED[:] = []
    L = O.random_text(hp, B, tseed, min_len=min_len, max_len=max_len)
    ends0 = np.array([np.where(L[i] == 0)[0][0] for i in range(B)])
    durations = None
    dur_lists = []
    if hp.use_external_durations:
        # per-symbol durations in fine frames -> hard attention matrix, padded and subsampled by r, exactly as the
        # reference's load_data does (data_load.py:243-251) with the reference's own helpers
        rng = np.random.default_rng(tseed + 1000)
        durations = np.zeros((B, hp.max_T, hp.max_N), np.int32)
        for i in range(B):
            n = int(ends0[i])
            budget = hp.max_T * hp.r - rng.integers(0, 3 * hp.r)
            d = rng.integers(0, 2 * budget // n + 1, size=n)
            while d.sum() > budget:
                d[rng.integers(0, n)] //= 2
            d = d.astype(np.int32)
            dm = ref_data_load.durations_to_hard_attention_matrix(d)
            dm = ref_data_load.end_pad_for_reduction_shape_sync(dm, hp)[0::hp.r, :]
            durations[i, :dm.shape[0], :dm.shape[1]] = dm
            dur_lists.append(d)

    # --- the REFERENCE's synthesize() (synthesize.py:449-617) around the REFERENCE graphs ---
    bases = ["utt_%s_%03d" % (tag, i) for i in range(B)]
    speaker_id = hp.speaker_list[speaker_ix] if hp.multispeaker else ""
    if hp.multispeaker:                               # synthesize.py:496-501
        assert dict(zip(hp.speaker_list, range(len(hp.speaker_list))))[speaker_id] == speaker_ix
    rec = HOST.synthesize(hp, L, bases, speaker_id=speaker_id, durations=durations)
    K, V = rec["encode_text"]
    Y, lengths, alignments = rec["synth_codedtext2mel"]
    Z = np.nan_to_num(rec["synth_mel2mag"])            # synthesize.py:578-579
    ends = np.asarray(rec["get_text_lengths"])
    assert np.array_equal(ends, ends0)
    steps = rec["decode_runs"]
    trace = rec["trace"]
    t2m_names, ssrn_names = rec["t2m_names"], rec["ssrn_names"]
    assert len(rec["waves"]) == B and [w[0] for w in rec["waves"]] == [os.path.join(rec["outdirs"][0], b + ".wav") for b in bases]
    for i, (_, mag) in enumerate(rec["waves"]):       # per-utterance trimming mag[:lengths[i]*r] (synthesize.py:608)
        assert np.array_equal(mag, Z[i, :lengths[i] * hp.r, :])

    out = dict(L=L, ends=ends, K=K, V=V, Q_step0=rec["Q_step0"], Y=Y, alignments=alignments,
               max_attentions_trace=np.array(trace, np.int32), t_ends=np.array(lengths, np.int32),
               steps_run=np.int32(steps), Z=Z,
               cdp=np.array(rec["cdp"], np.float64), ain=np.array([a[0] for a in rec["ap"]], np.float64),
               aout=np.array([a[1] for a in rec["ap"]], np.float64),
               wav_rows=np.array([w[1].shape[0] for w in rec["waves"]], np.int32))
    if hp.multispeaker:
        out["speakers"] = np.asarray(rec["speakers"]).astype(np.int32)
    if durations is not None:
        out["durations"] = durations
        out["duration_lists"] = np.concatenate(dur_lists)
    np.savez_compressed(os.path.join(HERE, "wiring_%s.npz" % tag), **out)
    report = [l for l in rec["stdout"].splitlines() if " | " in l]
    meta = dict(tag=tag, cfg=cfg, B=B, max_N=max_N, max_T=max_T, weight_seed=wseed, text_seed=tseed,
                min_len=min_len, max_len=max_len, stop=bool(stop), speaker_ix=speaker_ix, override=override or {},
                speaker_id=speaker_id, bases=bases, outdir=rec["outdirs"][0], wav_files=[w[0] for w in rec["waves"]],
                plot_files=[p[0] for p in rec["plots"]], plot_shapes=[list(p[1]) for p in rec["plots"]],
                report_lines=report, ssrn_batches=rec["ssrn_batches"], reference_report_crashes=bool(rec["report_crashes"]),
                variables=[[n, list(W[n].shape)] for n in t2m_names + ssrn_names],
                n_params_t2m=int(sum(W[n].size for n in t2m_names)),
                n_params_ssrn=int(sum(W[n].size for n in ssrn_names)))
    with open(os.path.join(HERE, "wiring_%s.json" % tag), "w") as f:
        json.dump(meta, f, indent=1)
    print(tag, "steps", steps, "t_ends", list(lengths), "trace[-1]", trace[-1].tolist(),
          "nvars", len(meta["variables"]), "params", meta["n_params_t2m"], meta["n_params_ssrn"], "outdir", meta["outdir"])


def frontend_case():
    """reference load_config + load_data(mode='synthesis') on the fixture transcript."""
    out = {}
    for cfg in ("lj_tutorial.cfg", "lj_test.cfg"):
        hp = ref_configuration.load_config(os.path.join(REF, "config", cfg))
        hp.test_transcript = os.path.join(HERE, "test_transcript_%s.csv" % cfg.split(".")[0])
        with contextlib.redirect_stderr(io.StringIO()):
            ds = ref_data_load.load_data(hp, mode="synthesis")
        tag = cfg.split(".")[0]
        out[tag + "_L"] = ds["texts"]
        out[tag + "_bases"] = np.array([os.path.basename(p)[:-4] for p in ds["fpaths"]])
        out[tag + "_text_lengths"] = np.array(ds["text_lengths"], np.int32)
        print(tag, "front-end:", ds["texts"].shape, ds["text_lengths"])
    config_snapshot(("lj_tutorial.cfg", "lj_test.cfg", "vctk_01.cfg"), fresh=True)
    np.savez_compressed(os.path.join(HERE, "frontend.npz"), **out)


def sdp_frontend_case():
    """'speaker_dependent_phones' in hp.multispeaker (data_load.py:42-46, 61-64, 153-154): the reference's load_vocab on such a
    configuration, and what its load_data(mode='synthesis') does with it (an exception: `speaker` is only assigned when the speaker
    is read from the transcript, which synthesis never does, data_load.py:80-81, 145-154)."""
    hp = ref_configuration.load_config(os.path.join(REF, "config", "lj_tutorial.cfg"))
    hp.multispeaker = ["speaker_dependent_phones"]
    hp.speaker_list = ["<PADDING>", "spkA", "spkB"]
    hp.test_transcript = os.path.join(HERE, "test_transcript_lj_tutorial.csv")
    c2i, i2c = ref_data_load.load_vocab(hp)
    raised = None
    try:
        with contextlib.redirect_stderr(io.StringIO()), contextlib.redirect_stdout(io.StringIO()):
            ref_data_load.load_data(hp, mode="synthesis")
    except BaseException as e:       # noqa: BLE001 (sys.exit included: whatever the reference does is the golden)
        raised = [type(e).__name__, str(e)]
    with open(os.path.join(HERE, "frontend_sdp.json"), "w") as f:
        json.dump({"speaker_list": hp.speaker_list, "char2idx": c2i, "n_vocab": len(i2c), "synthesis_load_data_raises": raised}, f, indent=1, sort_keys=True)
    print("speaker-dependent phones:", len(i2c), "symbols; load_data(mode='synthesis') ->", raised)


def config_snapshot(cfgs, fresh=False):
    """config attribute snapshot (the drop-in config API): every simple-typed attribute"""
    path = os.path.join(HERE, "config_snapshot.json")
    snap = {} if fresh else json.load(open(path))
    for cfg in cfgs:
        hp = ref_configuration.load_config(os.path.join(REF, "config", cfg))
        d = {}
        for k, v in sorted(hp.__dict__.items()):
            if isinstance(v, (int, float, str, bool, list, dict, type(None))) and "dir" not in k \
                    and k not in ("transcript", "test_transcript", "waveforms", "logdir", "topworkdir"):
                d[k] = v
        snap[cfg] = d
    with open(path, "w") as f:
        json.dump(snap, f, indent=1, sort_keys=True)


# option variants of the shipped configs beyond C1-C5.  The reference's load_config re-executes
# every config file in ONE module namespace (imp.load_source('config', ...)), so attributes a config does not set
# survive from the previously loaded one: every case below therefore runs in its own interpreter, exactly like the
# reference (one config per process).
VARIANTS = {
    # norm=None + turn_off_monotonic_for_synthesis (config/project/baseline.cfg)
    "proj_nomono": dict(cfg="project/baseline.cfg", B=3, max_N=20, max_T=14, wseed=41, tseed=42, min_len=6, max_len=17,
                        stop=True),
    # ssw10/G1ABC_01.cfg: norm=None (+ whatever else that file really sets)
    "g1abc_nonorm": dict(cfg="ssw10/G1ABC_01.cfg", B=2, max_N=14, max_T=18, wseed=43, tseed=44, min_len=3, max_len=7,
                         stop=True),
    # multispeaker ['text_encoder_input', 'audio_decoder_input'] (config/nancyplusnick_01..03.cfg)
    "nn_spk_in": dict(cfg="nancyplusnick_01.cfg", B=2, max_N=16, max_T=12, wseed=45, tseed=46, min_len=8, max_len=15,
                      stop=True, speaker_ix=1),
    # multispeaker ['text_encoder_towards_end', 'audio_decoder_input'] (config/vctk_02.cfg)
    "vctk02_spk_end": dict(cfg="vctk_02.cfg", B=2, max_N=16, max_T=12, wseed=47, tseed=48, min_len=8, max_len=15,
                           stop=True, speaker_ix=5),
    # use_external_durations: FixedAttention (config/ssw10/G1AB_03.cfg, project/fa_as_attention.cfg)
    "g1ab_extdur": dict(cfg="ssw10/G1AB_03.cfg", B=3, max_N=14, max_T=16, wseed=51, tseed=52, min_len=4, max_len=11,
                        stop=True),
    # 'audio_encoder_input' (networks.py:237-245): no shipped config uses it -- vctk_01.cfg with multispeaker overridden
    "vctk_spk_audioenc": dict(cfg="vctk_01.cfg", B=2, max_N=14, max_T=12, wseed=53, tseed=54, min_len=6, max_len=13,
                              stop=True, speaker_ix=9, override={"multispeaker": ["audio_encoder_input", "audio_decoder_input"]}),
    # multispeaker ['learn_channel_contributions'] (config/vctk_03_lcc.cfg, nancyplusnick_04_lcc.cfg)
    "vctk03_lcc": dict(cfg="vctk_03_lcc.cfg", B=3, max_N=16, max_T=12, wseed=49, tseed=50, min_len=8, max_len=15,
                       stop=True, speaker_ix=3),
    # hp.concatenate_query = False (networks.py:317-321: AudioDec reads the context alone, C_1's kernel is (1, d, d)); no shipped
    # config sets it -- lj_tutorial.cfg with the attribute overridden
    "lj_noconcat": dict(cfg="lj_tutorial.cfg", B=3, max_N=14, max_T=16, wseed=55, tseed=56, min_len=3, max_len=9,
                        stop=True, override={"concatenate_query": False}),
    # hp.squash_output_t2m = hp.squash_output_ssrn = False (networks.py:430-433, 533-536: Y = Y_logits, Z = Z_logits)
    "lj_nosquash": dict(cfg="lj_tutorial.cfg", B=3, max_N=14, max_T=16, wseed=57, tseed=58, min_len=3, max_len=9,
                        stop=True, override={"squash_output_t2m": False, "squash_output_ssrn": False}),
    # 'ssrn_input' in hp.multispeaker (networks.py:457-465).  The reference's synth_mel2mag feeds g.mels only (synthesize.py:257), so
    # under TensorFlow this configuration dies there with "You must feed a value for placeholder tensor" (g.speakers); the eager
    # stand-in's session hands the SSRN graph the speaker codes of the Text2Mel run, i.e. the Z stored here is
    # sess.run(g.Z, {g.mels: Y, g.speakers: codes}) -- the graph-surface value
    "vctk_spk_ssrn": dict(cfg="vctk_01.cfg", B=2, max_N=14, max_T=12, wseed=59, tseed=60, min_len=6, max_len=13,
                          stop=True, speaker_ix=4, override={"multispeaker": ["audio_decoder_input", "ssrn_input"]}),
}


def durations_frontend_case(cfg):
    """reference load_data(mode='synthesis') with hp.use_external_durations on a transcript carrying the 6th field"""
    hp = ref_configuration.load_config(os.path.join(REF, "config", cfg))
    hp.test_transcript = os.path.join(HERE, "test_transcript_durations.csv")
    with contextlib.redirect_stderr(io.StringIO()), contextlib.redirect_stdout(io.StringIO()):
        ds = ref_data_load.load_data(hp, mode="synthesis")
    np.savez_compressed(os.path.join(HERE, "frontend_durations.npz"), L=ds["texts"], durations=ds["durations"],
                        bases=np.array([os.path.basename(p)[:-4] for p in ds["fpaths"]]),
                        text_lengths=np.array(ds["text_lengths"], np.int32))
    print("durations front-end:", ds["texts"].shape, ds["durations"].shape, ds["durations"].sum(axis=(1, 2)))


def variant_cases():
    import subprocess
    for tag in VARIANTS:
        subprocess.check_call([sys.executable, "-B", os.path.abspath(__file__), "case", tag])


def chunking_case():
    """synth_mel2mag's batch splitting (synthesize.py:250-260) as the reference computes it under Python 2:
    for n utterances and a batch size, the list of chunk lengths handed to sess.run."""
    res = {}

    class G(object):
        Z, mels = "Z", "mels"

    class S(object):
        def __init__(self):
            self.sizes = []

        def run(self, fetch, feed):
            self.sizes.append(len(feed["mels"]))
            return np.zeros((len(feed["mels"]), 1, 1), np.float32)
    for n in (1, 2, 3, 5, 7, 8, 16, 127, 128, 129, 255, 256, 257, 300):
        for bs in (128, 32, 2, 1, 0, -1):
            s_ = S()
            HOST.mod.synth_mel2mag(None, np.zeros((n, 1, 1), np.float32), G, s_, batchsize=bs)
            res["%d,%d" % (n, bs)] = s_.sizes
    with open(os.path.join(HERE, "mel2mag_chunks.json"), "w") as f:
        json.dump(res, f, sort_keys=True)
    print("synth_mel2mag chunking:", len(res), "cases")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "sdp":
        sdp_frontend_case()
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[1] == "case":           # one variant case, in this (fresh) interpreter
        v = dict(VARIANTS[sys.argv[2]])
        config_snapshot((v["cfg"],))
        if sys.argv[2] == "g1ab_extdur":
            durations_frontend_case(v["cfg"])
        run_case(sys.argv[2], v.pop("cfg"), **v)
        sys.exit(0)
    frontend_case()
    sdp_frontend_case()
    chunking_case()
    # free-running, no early stop reached within max_T (long texts)
    run_case("lj_free", "lj_tutorial.cfg", B=2, max_N=24, max_T=16, wseed=11, tseed=12,
             min_len=18, max_len=23, stop=True)
    # short texts so that attention reaches `ends`: exercises t_ends / break / zero tail
    run_case("lj_stop", "lj_tutorial.cfg", B=3, max_N=12, max_T=24, wseed=21, tseed=22,
             min_len=2, max_len=5, stop=True)
    # multispeaker (audio_decoder_input) wiring: vctk_01.cfg
    run_case("vctk_spk", "vctk_01.cfg", B=2, max_N=16, max_T=12, wseed=31, tseed=32,
             min_len=8, max_len=15, stop=True, speaker_ix=7)
    variant_cases()          # each in its own interpreter (see VARIANTS); they add their configs to the snapshot
