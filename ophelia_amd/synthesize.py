#!/usr/bin/env python
"""Synthesis driver with the reference's interface: counterpart of synthesize.py
(main_work 635-683, synthesize 442-632, encode_text 232-240, synth_codedtext2mel 150-230,
synth_text2mel 62-132, synth_mel2mag 250-260, get_text_lengths 242-247, split_batch 263-268,
make_mel_batch 270-285, list2batch 287-299, restore_* 302-330, synth_wave 433-440).

    python -m ophelia_amd.synthesize -c config/X.cfg [-N n] [-speaker id] [-odir d]
           [-t2m_epoch e] [-ssrn_epoch e] [-max_N n] [-max_T t] [-tr transcript] [-ncores k]

Same flags, same output directory naming ({sampledir|odir/cfg}/t2m{E}_ssrn{E}[_speaker-{id}]/),
same trimming rules, same {base}.wav files (16-bit PCM) from Griffin-Lim -- which runs batched on the GPU
(ophelia_amd.vocoder) instead of on `-ncores` CPU processes -- and the same per-utterance
"File | CDP | Ain" report.  With hp.store_synth_features the trimmed magnitudes {base}.npy are stored as in the
reference (synthesize.py:436-437); {base}.png attention plots are written when matplotlib is importable.  The default
output set is exactly the reference's; hp.store_synth_extras (an extension, off unless a config sets it) adds
{base}.mel.npy and {base}.alignment.npy.  Not provided: the WORLD vocoder (external binaries).  Under torchrun (WORLD_SIZE>1) the utterances are sharded
over the GPUs of the node (ophelia_amd.parallel).
"""
from __future__ import print_function

import os
import sys
import timeit
from argparse import ArgumentParser

import numpy as np

from . import _lib
from . import parallel
from . import vocoder
from .architectures import (SSRNGraph, Session, Text2MelGraph, restore_archived_model_parameters,
                            restore_latest_model_parameters)
from .calculate_CDP_Ain_Aout import getAP, getCDP
from .configuration import load_config
from .data_load import load_data
from .libutil import basename, safe_makedir
from .utils import plot_alignment

try:
    import matplotlib  # noqa: F401
    _HAVE_MATPLOTLIB = True
except ImportError:             # the plots are a convenience output: without matplotlib they are skipped, not faked
    _HAVE_MATPLOTLIB = False


def start_clock(comment):
    print("%s... " % (comment), end="")
    return (timeit.default_timer(), comment)


def stop_clock(clock, width=40):
    start_time, comment = clock
    padding = (width - len(comment)) * " "
    print("%s--> took %.2f seconds" % (padding, (timeit.default_timer() - start_time)))


def _check_scope(hp):
    if getattr(hp, "merlin_label_dir", "") or "position_in_phone" in getattr(hp, "history_type", ""):
        raise NotImplementedError("Merlin labels / position-in-phone history are not supported")


def get_text_lengths(L):
    """Index of the first padding id (0) in each row; IndexError if a row has no padding."""
    ends = []
    for i in range(len(L)):
        ends.append((np.where(L[i, :] == 0)[0][0]))
    return np.array(ends)


def encode_text(hp, L, g, sess, speaker_data=None, labels=None):
    """K, V = sess.run([g.K, g.V], {g.L: L [, g.speakers]})"""
    _check_scope(hp)
    return sess.ensure_ready().encode_text(L, speaker_data)


def synth_codedtext2mel(hp, K, V, ends, g, sess, speaker_data=None, duration_data=None,
                        labels=None, position_in_phone_data=None):
    """Autoregressive decode of the coded text.  Returns (Y, t_ends list, alignments) exactly as the
    reference: Y (B,max_T,n_mels) with frames after the break step left at 0, t_ends[i] = first step at
    which the attention peak reached the end of text i (max_T if never), alignments (B,max_N,max_T)."""
    _check_scope(hp)
    eng = sess.ensure_ready()
    dist_on = parallel._dist() is not None
    if getattr(hp, "use_external_durations", False):
        # FixedAttention: the utterance lengths are known up front (synthesize.py:168-169); the loop runs until the
        # longest utterance of the WHOLE batch is through (211-216) -- across all shards when the batch is sharded
        assert duration_data is not None, "hp.use_external_durations: duration_data (B, max_T, max_N) required"
        n_steps = 0
        if dist_on:
            longest = int(np.asarray(duration_data).sum(axis=(1, 2)).max()) if len(duration_data) else 0
            n_steps = min(hp.max_T, parallel.global_max_int(longest, device=sess.device) + 1)
        Y, t_ends, alignments, _ = eng.text2mel_durations(K, V, duration_data, speaker_data, n_steps=n_steps)
        return (Y, [int(t) for t in t_ends], alignments)
    if not dist_on:
        # hp.synth_stop_mode (this package's extension, default 0 = the reference's break rule): 1 = fixed length, every
        # utterance runs max_T steps -- the configuration bench.py quotes its metric on
        stop_mode = int(getattr(hp, "synth_stop_mode", _lib.STOP_REFERENCE))
        Y, t_ends, alignments, _ = eng.text2mel(K, V, ends, speaker_data, stop_mode)
        return (Y, [int(t) for t in t_ends], alignments)
    # sharded batch: reproduce the batch-coupled break (synthesize.py:225-228) across ranks
    state = {}

    def local():
        Y, t_ends, al, steps = eng.text2mel(K, V, ends, speaker_data, _lib.STOP_REFERENCE)
        state["out"] = (Y, t_ends, al)
        return steps

    def resume(t0, t1):
        eng.B = len(K)
        eng.decode_steps(t0, t1, _lib.STOP_NEVER)
        state["out"] = eng.fetch_mel()

    parallel.sharded_text2mel(local, resume, None, hp.max_T, device=sess.device)
    Y, t_ends, alignments = state["out"]
    return (Y, [int(t) for t in t_ends], alignments)


def synth_text2mel(hp, L, g, sess, speaker_data=None, duration_data=None, labels=None, position_in_phone_data=None):
    """The reference keeps this slower variant (K/V recomputed every step) for validation; results are
    identical to encode_text + synth_codedtext2mel, which is what runs here.  Returns (Y, t_ends)."""
    K, V = encode_text(hp, L, g, sess, speaker_data=speaker_data, labels=labels)
    Y, t_ends, _ = synth_codedtext2mel(hp, K, V, get_text_lengths(L), g, sess, speaker_data=speaker_data,
                                       duration_data=duration_data)
    return (Y, t_ends)


def synth_mel2mag(hp, Y, g, sess, batchsize=128):
    if hp is not None and "ssrn_input" in (getattr(hp, "multispeaker", None) or []):
        # The reference's function feeds {g.mels: Y_batch} alone (synthesize.py:257, "#assert speaker_data==None ## TODO"): with the
        # speaker embedding wired into SSRN (networks.py:457-465) TensorFlow ends the run here.  Same place, same kind of error;
        # the speaker-conditioned SSRN itself is reachable through sess.run(g.Z, {g.mels: Y, g.speakers: codes}).
        from .architectures import InvalidArgumentError
        raise InvalidArgumentError("You must feed a value for placeholder tensor 'speakers' ('ssrn_input' in hp.multispeaker; "
                                   "synth_mel2mag feeds g.mels only, synthesize.py:257)")
    eng = sess.ensure_ready()
    if getattr(eng, "is_resident_mel", lambda y: False)(Y):
        # the very array synth_codedtext2mel returned: its frames are still in HBM and SSRN has been running over them
        # while the decoder produced the later ones.  (The reference splits into chunks of `batchsize` utterances only to
        # bound TF's memory; utterances are independent, the result is the same array.)
        return eng.ssrn(Y)
    if batchsize > 0:
        nbatches = max(1, len(Y) // batchsize)       # the reference's Python-2 integer division
        batches = np.array_split(Y, nbatches)
    else:
        batches = [Y]
    return np.concatenate([eng.ssrn(Y_batch) for Y_batch in batches])


def split_batch(synth_batch, end_indices):
    return [predmel[:end_indices[i], :] for i, predmel in enumerate(synth_batch)]


def make_mel_batch(hp, fnames, oracle=True):
    """Stack coarse mels into a zero-padded (n, max_T, n_mels) batch; lengths are in full-rate frames (x r).
    oracle=True reads {hp.coarse_audio_dir}/{base}.npy for each name, otherwise `fnames` are .npy paths."""
    paths = [os.path.join(hp.coarse_audio_dir, basename(f) + ".npy") for f in fnames] if oracle else list(fnames)
    batch = np.zeros((len(paths), hp.max_T, hp.n_mels), np.float32)
    lengths = []
    for row, path in zip(batch, paths):
        mel = np.load(path)
        row[:mel.shape[0]] = mel
        lengths.append(mel.shape[0] * hp.r)
    return batch, lengths


def list2batch(inlist, pad_length):
    """Zero-pad a list of (len_i, dim) arrays to one (n, pad_length, dim) float32 batch (0 = longest)."""
    dim = inlist[0].shape[1]
    longest = max(a.shape[0] for a in inlist)
    pad_length = pad_length or longest
    assert longest <= pad_length and all(a.shape[1] == dim for a in inlist)
    batch = np.zeros((len(inlist), pad_length, dim), np.float32)
    for row, a in zip(batch, inlist):
        row[:a.shape[0]] = a
    return batch


# utterances vocoded per Griffin-Lim launch are bounded by this many spectrogram frames (~30 KB of HBM per frame)
GL_MAX_FRAMES_PER_CALL = 32768


def synth_wave(hp, mag, outfile, wav=None):
    """synthesize.py:433-440.  `wav` may carry the already (batch-)vocoded samples for this utterance."""
    if hp.vocoder == "griffin_lim":
        if wav is None:
            wav = vocoder.spectrogram2wav(hp, mag, device=int(os.environ.get("LOCAL_RANK", "0")))
        if hp.store_synth_features:          # To synthesize using WaveRNN save the mag spectrum created by SSRN
            np.save(outfile.replace(".wav", ".npy"), mag)
        vocoder.write_wav(outfile, wav, hp.sr)
    elif hp.vocoder == "world":
        raise NotImplementedError("the WORLD vocoder shells out to external binaries (synthesize.py:366-430); "
                                  "only hp.vocoder == 'griffin_lim' is supported")


def synth_waves(hp, mags, outfiles, device=0):
    """The loop of synthesize.py:604-617 with the utterances vocoded together on the GPU instead of one process per
    utterance (`-ncores`)."""
    if hp.vocoder != "griffin_lim":
        return [synth_wave(hp, m, f) for m, f in zip(mags, outfiles)]
    # An utterance whose attention reached the end of the text at step 0 has no frames (t_end = 0), one frame gives no
    # samples either (hop * (T - 1)): they get an empty wav and stay out of the Griffin-Lim batch, which needs >= 2 frames
    short = [k for k, m in enumerate(mags) if len(m) < 2]
    if short:
        for k in short:
            print("Warning: %s has %d spectrogram frame(s); writing an empty wav" % (outfiles[k], len(mags[k])))
            synth_wave(hp, mags[k], outfiles[k], wav=np.zeros((0,), np.float32))
        keep = [k for k in range(len(mags)) if k not in set(short)]
        mags, outfiles = [mags[k] for k in keep], [outfiles[k] for k in keep]
    voc = vocoder._vocoder_for(hp, device)
    i = 0
    while i < len(mags):
        j, frames = i, 0
        while j < len(mags) and (j == i or frames + len(mags[j]) <= GL_MAX_FRAMES_PER_CALL):
            frames += len(mags[j])
            j += 1
        for m, f, w in zip(mags[i:j], outfiles[i:j], voc.spectrogram2wav_batch(mags[i:j])):
            synth_wave(hp, m, f, wav=w)
        i = j


def synthesize(hp, speaker_id="", num_sentences=0, ncores=1, topoutdir="", t2m_epoch=-1, ssrn_epoch=-1,
               weights=None, device=None):
    """topoutdir: store samples under here; defaults to hp.sampledir.
    t2m_epoch / ssrn_epoch: -1 = latest, else archived epoch.
    weights: optional {TF name: array} -- or a function of the variable inventory returning one -- bypassing the checkpoint lookup
    (tests / random-init runs)."""
    assert hp.vocoder in ["griffin_lim", "world"], "Other vocoders than griffin_lim/world not yet supported"
    _check_scope(hp)
    dist = parallel._dist()
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist else (0, 1)
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))

    dataset = load_data(hp, mode="synthesis")
    fpaths, L = dataset["fpaths"], dataset["texts"]
    if num_sentences > 0:
        assert num_sentences <= len(fpaths)
        L = L[:num_sentences, :]
        fpaths = fpaths[:num_sentences]
    bases = [basename(fpath) for fpath in fpaths]
    duration_data = None
    if getattr(hp, "use_external_durations", False):          # synthesize.py:451-455
        duration_data = dataset["durations"]
        if num_sentences > 0:
            duration_data = duration_data[:num_sentences, :, :]
    if world > len(L):
        # more ranks than utterances would leave ranks with an empty shard: they cannot stage a batch, and skipping the
        # collectives of the others would hang them -- refuse up front, on every rank, before any collective
        sys.exit("%d ranks for %d utterance(s): start at most one rank per utterance (or raise -N)" % (world, len(L)))
    lo, hi = parallel.shard_range(len(L), rank, world)       # contiguous utterance shard of this GPU
    L, bases = L[lo:hi], bases[lo:hi]
    if duration_data is not None:
        duration_data = duration_data[lo:hi]

    if speaker_id:
        speaker2ix = dict(zip(hp.speaker_list, range(len(hp.speaker_list))))
        speaker_data = np.ones((len(L), 1)) * speaker2ix[speaker_id]
    else:
        speaker_data = None

    g1 = Text2MelGraph(hp, mode="synthesize"); print("Graph 1 (t2m) loaded")
    g2 = SSRNGraph(hp, mode="synthesize"); print("Graph 2 (ssrn) loaded")

    with Session(hp, device=device) as sess:
        if weights is not None:
            if callable(weights):          # weights as a function of the variable inventory (random-init runs: no probe engine needed)
                weights = weights(sess.inventory()) if rank == 0 or world == 1 else None
            if world > 1:
                weights = parallel.broadcast_weights(weights if rank == 0 else None, sess.inventory(), src=0, device=device)
            sess.assign(weights)
            t2m_epoch = ssrn_epoch = "rand" if t2m_epoch == -1 else t2m_epoch
        else:
            if t2m_epoch > -1:
                restore_archived_model_parameters(sess, hp, "t2m", t2m_epoch)
            else:
                t2m_epoch = restore_latest_model_parameters(sess, hp, "t2m")
            if ssrn_epoch > -1:
                restore_archived_model_parameters(sess, hp, "ssrn", ssrn_epoch)
            else:
                ssrn_epoch = restore_latest_model_parameters(sess, hp, "ssrn")

        t = start_clock("Text2Mel generating...")
        text_lengths = get_text_lengths(L)
        K, V = encode_text(hp, L, g1, sess, speaker_data=speaker_data)
        Y, lengths, alignments = synth_codedtext2mel(hp, K, V, text_lengths, g1, sess, speaker_data=speaker_data,
                                                     duration_data=duration_data)
        stop_clock(t)

        t = start_clock("Mel2Mag generating...")
        Z = synth_mel2mag(hp, Y, g2, sess)
        stop_clock(t)
        if (np.isnan(Z).any()):
            Z = np.nan_to_num(Z)

        if not topoutdir:
            topoutdir = hp.sampledir
        outdir = os.path.join(topoutdir, "t2m%s_ssrn%s" % (t2m_epoch, ssrn_epoch))
        if speaker_id:
            outdir += "_speaker-%s" % (speaker_id)
        safe_makedir(outdir)
        print("Plot attention, will save to following dir: %s" % (outdir))
        print("File |  CDP | Ain")
        for i in range(len(Z)):
            trimmed_alignment = alignments[i, :text_lengths[i], :lengths[i]]
            if _HAVE_MATPLOTLIB:                               # synthesize.py:594-595
                plot_alignment(hp, trimmed_alignment, utt_idx=lo + i + 1, t2m_epoch=t2m_epoch, dir=outdir,
                               outfile=os.path.join(outdir, bases[i]))
            CDP = getCDP(trimmed_alignment)
            APin, APout = getAP(trimmed_alignment)
            print("%s | %.2f | %.2f" % (bases[i], CDP, APin))
            if getattr(hp, "store_synth_extras", False):     # (extension, off by default: the reference's output set is .wav / .png / .npy)
                np.save(os.path.join(outdir, bases[i] + ".alignment.npy"), trimmed_alignment)

        print("Generating wav files, will save to following dir: %s" % (outdir))
        t = start_clock("Griffin-Lim generating...")
        mags = [mag[:lengths[i] * hp.r, :] for i, mag in enumerate(Z)]          # trim to generated length
        synth_waves(hp, mags, [os.path.join(outdir, b + ".wav") for b in bases], device=device)
        stop_clock(t)
        if getattr(hp, "store_synth_extras", False):
            for i, b in enumerate(bases):
                np.save(os.path.join(outdir, b + ".mel.npy"), Y[i, :lengths[i], :])
    return outdir


def main_work():
    a = ArgumentParser()
    a.add_argument("-c", dest="config", required=True, type=str)
    a.add_argument("-speaker", default="", type=str)
    a.add_argument("-N", dest="num_sentences", default=0, type=int)
    a.add_argument("-babble", action="store_true")
    a.add_argument("-ncores", type=int, default=1, help="Number of CPUs for Griffin-Lim stage (accepted and ignored: Griffin-Lim runs batched on the GPU)")
    a.add_argument("-odir", type=str, default="", help="Alternative place to put output samples")
    a.add_argument("-t2m_epoch", default=-1, type=int, help="Default: use latest (-1)")
    a.add_argument("-ssrn_epoch", default=-1, type=int, help="Default: use latest (-1)")
    a.add_argument("-max_N", default=-1, type=int, help="Default: use max_N from config")
    a.add_argument("-max_T", default=-1, type=int, help="Default: use max_T from config")
    a.add_argument("-tr", default="", type=str, help="Default:use test_transcript from config")
    a.add_argument("-random_init", default=-1, type=int, metavar="SEED",
                   help="(extension) seeded random-init weights instead of a checkpoint")
    opts = a.parse_args()

    hp = load_config(opts.config)
    if (opts.max_N != -1):
        hp.max_N = opts.max_N
    if (opts.max_T != -1):
        hp.max_T = opts.max_T
    if (opts.tr != ""):
        hp.test_transcript = opts.tr
    print("max_N=" + str(hp.max_N))
    print("max_T=" + str(hp.max_T))
    print("test_transcript=" + str(hp.test_transcript))

    outdir = opts.odir
    if outdir:
        outdir = os.path.join(outdir, basename(opts.config))
    if hp.multispeaker:
        assert opts.speaker, "Please specify a speaker from speaker_list with -speaker flag"
        assert opts.speaker in hp.speaker_list
    if opts.babble:
        sys.exit("babbling is outside the hot-path scope")

    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch
        import torch.distributed as dist
        use_gpu = torch.cuda.is_available()
        if use_gpu:
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl" if use_gpu else "gloo")

    weights = None
    if opts.random_init >= 0:
        from . import weights as WT
        seed = opts.random_init
        # (a function of the session's inventory rather than a probe engine of its own: one handle per process -- creating CU-masked
        #  streams again after another handle's were destroyed is a pattern that has hung in the HIP runtime, oph_api.hip)
        weights = lambda inventory: WT.random_weights(inventory, seed)
    synthesize(hp, speaker_id=opts.speaker, num_sentences=opts.num_sentences, ncores=opts.ncores,
               topoutdir=outdir, t2m_epoch=opts.t2m_epoch, ssrn_epoch=opts.ssrn_epoch, weights=weights)


def _hang_dump():
    """OPH_HANG_DUMP_S=n: if the process is still running after n seconds, every thread's Python stack goes to stderr and the
    process exits (diagnostics for a stuck device call: the innermost frame names the ctypes call that does not return)."""
    s = os.environ.get("OPH_HANG_DUMP_S")
    if s:
        import faulthandler
        faulthandler.dump_traceback_later(float(s), exit=True)


if __name__ == "__main__":
    _hang_dump()
    main_work()
