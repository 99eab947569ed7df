"""Transcript front-end of the synthesis path: counterpart of the reference's
data_load.py load_vocab 40-50, text_normalize 52-59, phones_normalize 61-69 and
load_data 71-265 restricted to mode='synthesis' (training / validation input
pipelines are out of scope).

Transcript format (README 'Data preparation'): UTF-8 lines
    name|unnormalised text|normalised text|phones[|speaker[|durations]]
blank lines are skipped; a line has either 1 field (audio only) or >= 3 fields."""
import codecs
import os
import re
import sys
import unicodedata

import numpy as np


def load_vocab(hp):
    vocab = hp.vocab
    if "speaker_dependent_phones" in hp.multispeaker:
        vocab = [hp.vocab[0]] + ["%s_%s" % (phone, spk) for spk in hp.speaker_list[1:] for phone in hp.vocab[1:]]
    # duplicate symbols (e.g. '<_START_>' twice in lj_tutorial.cfg): the LATER index wins,
    # exactly what a dict comprehension over enumerate() does in the reference (data_load.py:48)
    char2idx = {}
    for idx, char in enumerate(vocab):
        char2idx[char] = idx
    idx2char = dict(enumerate(vocab))
    return char2idx, idx2char


def text_normalize(text, hp):
    text = "".join(ch for ch in unicodedata.normalize("NFD", text) if unicodedata.category(ch) != "Mn")
    text = text.lower()
    text = re.sub("[^{}]".format(hp.vocab), " ", text)
    return re.sub("[ ]+", " ", text)


def phones_normalize(text, char2idx, speaker_code=""):
    phones = re.split(r"\s+", text.strip(" \n"))
    if speaker_code:
        phones = ["%s_%s" % (p, speaker_code) for p in phones]
    for p in phones:
        if p not in char2idx:
            print(text)
            sys.exit("Phone %s not listed in phone set" % (p))
    return phones


def load_data(hp, mode="synthesis"):
    """Returns {'texts': L (n_utts, max_N) int32 zero-padded, 'fpaths': [...], 'text_lengths': [...]}.
    Speaker identity at synthesis comes from the -speaker flag, never from the transcript
    (data_load.py:80); utterances longer than max_N are silently dropped (165-168)."""
    assert mode in ("train", "synthesis", "validation")
    if mode != "synthesis":
        raise NotImplementedError("only mode='synthesis' is on the hot path (training input pipeline is out of scope)")
    if getattr(hp, "merlin_label_dir", ""):
        raise NotImplementedError("Merlin labels are not supported")
    use_durations = getattr(hp, "use_external_durations", False)
    durations = []
    char2idx, _ = load_vocab(hp)
    with codecs.open(hp.test_transcript, "r", "utf-8") as f:
        lines = f.readlines()
    fpaths, text_lengths, texts = [], [], []
    for line in lines:
        line = line.strip("\n\r |")
        if line == "":
            continue
        fields = line.strip().split("|")
        if len(fields) > 1:
            assert len(fields) >= 3, fields
        fname = fields[0]
        norm_text = fields[2] if len(fields) > 1 else None
        if getattr(hp, "validpatt", "") and False:
            pass  # validpatt filtering applies to train/validation only (data_load.py:135-141)
        if norm_text is None:
            ids = []
        elif hp.input_type == "phones":
            assert len(fields) >= 4, fields
            spk_code = ""
            if "speaker_dependent_phones" in hp.multispeaker:
                # data_load.py:153-154 takes `speaker_code = speaker`, but in mode='synthesis' get_speaker_codes is False (:80-81) and
                # `speaker` was never assigned (:145-149): the reference dies here with this very exception.  load_vocab above has built
                # the speaker-dependent vocabulary (data_load.py:42-46) exactly as the reference does before it gets here.
                raise UnboundLocalError("local variable 'speaker' referenced before assignment")
            ids = [char2idx[p] for p in phones_normalize(fields[3], char2idx, speaker_code=spk_code)]
        elif hp.input_type == "letters":
            ids = [char2idx[ch] for ch in text_normalize(norm_text, hp) + "E"]      # E: EOS
        else:
            raise ValueError("unknown input_type %r" % hp.input_type)
        if len(ids) > hp.max_N:
            continue
        if use_durations:            # 6th field: one duration (in un-reduced frames) per input symbol (data_load.py:181-193)
            assert len(fields) >= 6, fields
            dur = np.array([int(v) for v in re.split(r"\s+", fields[5].strip(" "))], np.int32)
            assert len(dur) == len(ids), (len(dur), len(ids), fname)
            durations.append(dur)
        texts.append(np.array(ids, np.int32))
        fpaths.append(os.path.join(hp.waveforms, fname + ".wav"))
        text_lengths.append(len(ids))
    L = np.zeros((len(texts), hp.max_N), np.int32)
    for i, t in enumerate(texts):
        L[i, :len(t)] = t
    dataset = {"texts": L, "fpaths": fpaths, "text_lengths": text_lengths, "audio_lengths": [], "label_lengths": []}
    if use_durations:                # (n, max_T, max_N) hard attention matrices, data_load.py:243-251
        stacked = np.zeros((len(texts), hp.max_T, hp.max_N), np.int32)
        for i, dur in enumerate(durations):
            A = durations_to_hard_attention_matrix(dur)
            A = end_pad_for_reduction_shape_sync(A, hp)[0::hp.r, :]
            stacked[i, :A.shape[0], :A.shape[1]] = A
        dataset["durations"] = stacked
    return dataset


def durations_to_hard_attention_matrix(durations):
    """(nphones,) frame counts -> (nframes, nphones) 0/1 matrix whose row t selects the phone frame t belongs to
    (utils.py:197-219; zero-duration phones get no row)."""
    durations = np.asarray(durations)
    owner = np.repeat(np.arange(len(durations)), durations)
    A = np.zeros((len(owner), len(durations)), np.float32)
    A[np.arange(len(owner)), owner] = 1.0
    return A


def end_pad_for_reduction_shape_sync(data, hp):
    """zero rows appended so that the number of frames is a multiple of hp.r (utils.py:190-194)"""
    short = (-data.shape[0]) % hp.r
    return np.pad(data, [[0, short], [0, 0]], mode="constant")
