"""GPU tests of the driver level: synthesize() output contract (directory / file naming, trimming)
and utterance sharding with the batch-coupled global stop, checked against the oracle."""
import os
import socket

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from oracle import ophelia_oracle as O

pytestmark = pytest.mark.gpu


def _hp(max_T=24):
    from ophelia_amd.configuration import load_config
    hp = load_config(os.path.join(GOLDEN, "cfg_unit.cfg"))
    hp.max_T = max_T
    hp.store_synth_features = True          # keep {base}.npy next to the .wav (synthesize.py:436-437)
    hp.store_synth_extras = True            # ... and this package's opt-in extras: {base}.mel.npy / .alignment.npy
    return hp


def test_synthesize_driver_outputs(tmp_path):
    from ophelia_amd import synthesize as S
    from ophelia_amd.data_load import load_data
    hp = _hp()
    W = O.random_weights(hp, 41)
    outdir = S.synthesize(hp, num_sentences=4, topoutdir=str(tmp_path / "out" / "cfg_unit"), weights=W)
    assert outdir.endswith(os.path.join("cfg_unit", "t2mrand_ssrnrand"))
    L = load_data(hp, mode="synthesis")["texts"][:4]
    ends = O.get_text_lengths(L)
    K, V = O.encode_text(hp, W, L)
    Y0, t_ends, _ = O.synth_codedtext2mel_incremental(hp, W, K, V, ends)
    Z0 = O.synth_mel2mag(hp, W, Y0)
    import wave
    from oracle import griffin_lim_oracle as GL
    names = ["LJ003-0043", "LJ050-0001", "LJ050-0002", "LJ050-0003"]
    for i, base in enumerate(names):
        mag = np.load(os.path.join(outdir, base + ".npy"))
        mel = np.load(os.path.join(outdir, base + ".mel.npy"))
        assert mag.shape == (t_ends[i] * hp.r, hp.full_dim) and mag.dtype == np.float32     # synthesize.py:608
        assert mel.shape == (t_ends[i], hp.n_mels)
        assert np.abs(mag - Z0[i, :t_ends[i] * hp.r]).max() < 1e-4
        assert np.abs(mel - Y0[i, :t_ends[i]]).max() < 1e-4
        with wave.open(os.path.join(outdir, base + ".wav"), "rb") as f:                     # synthesize.py:433-438
            n = hp.hop_length * (t_ends[i] * hp.r - 1)
            assert (f.getnchannels(), f.getsampwidth(), f.getframerate(), f.getnframes()) == (1, 2, hp.sr, n)
            pcm = np.frombuffer(f.readframes(n), "<i2").astype(np.float64) / 32767.0
        if i == 0:                                           # vocoder parity on the driver's own output (mag as stored)
            ref = GL.spectrogram2wav(hp, mag)
            assert np.abs(pcm - np.clip(ref, -1, 1)).max() <= 2e-3 * np.abs(ref).max() + 1.0 / 32767
    have_png = os.path.exists(os.path.join(outdir, names[0] + ".png"))            # written when matplotlib is present
    assert len(os.listdir(outdir)) == (20 if have_png else 16)
    if have_png:
        assert open(os.path.join(outdir, names[0] + ".png"), "rb").read(8) == b"\x89PNG\r\n\x1a\n"


def test_synthesize_driver_with_external_durations(tmp_path):
    """hp.use_external_durations: durations come from the transcript's 6th field, utterance lengths from their sums"""
    import wave
    from ophelia_amd import synthesize as S
    from ophelia_amd.data_load import load_data
    hp = _hp()
    hp.use_external_durations = True
    hp.test_transcript = os.path.join(GOLDEN, "test_transcript_durations.csv")
    W = O.random_weights(hp, 44)
    outdir = S.synthesize(hp, topoutdir=str(tmp_path / "out"), weights=W)
    ds = load_data(hp, mode="synthesis")
    L, D = ds["texts"], ds["durations"]
    K, V = O.encode_text(hp, W, L)
    Y0, t0, _ = O.synth_codedtext2mel_incremental(hp, W, K, V, O.get_text_lengths(L), durations=D)
    assert t0 == [12, 11, 9]                                  # ceil(sum of the durations / r)
    for i, base in enumerate(["DUR-0001", "DUR-0002", "DUR-0003"]):
        mel = np.load(os.path.join(outdir, base + ".mel.npy"))
        assert mel.shape == (t0[i], hp.n_mels) and np.abs(mel - Y0[i, :t0[i]]).max() < 1e-4
        al = np.load(os.path.join(outdir, base + ".alignment.npy"))
        assert np.array_equal(al, D[i, :t0[i], :ds["text_lengths"][i]].T)
        with wave.open(os.path.join(outdir, base + ".wav"), "rb") as f:
            assert f.getnframes() == hp.hop_length * (t0[i] * hp.r - 1)


def test_synthesize_from_tf_format_checkpoints(tmp_path):
    """CLI-level restore path: latest t2m checkpoint + archived ssrn epoch, both in TF tensor-bundle format."""
    from ophelia_amd import synthesize as S, tf_checkpoint as T
    hp = _hp()
    hp.logdir = str(tmp_path / "work" / "train")
    hp.sampledir = str(tmp_path / "work" / "synth")
    W = O.random_weights(hp, 41)
    adam = {n + "/Adam": np.zeros_like(v) for n, v in W.items()}           # optimizer slots must be ignored
    T.write_checkpoint(hp.logdir + "-t2m/model_epoch_3", {**{n: v for n, v in W.items() if n.startswith("Text2Mel")}, **adam}, data_crc=False)
    T.write_checkpoint(hp.logdir + "-ssrn/archive/model_epoch_5", {n: v for n, v in W.items() if n.startswith("SSRN")}, data_crc=False)
    outdir = S.synthesize(hp, num_sentences=2, ssrn_epoch=5)
    assert outdir == os.path.join(hp.sampledir, "t2m3_ssrn5")
    ref = S.synthesize(hp, num_sentences=2, topoutdir=str(tmp_path / "ref"), weights=W)
    for f in sorted(os.listdir(ref)):
        if f.endswith(".png"):
            continue                                       # the plot title carries the epoch label
        assert open(os.path.join(ref, f), "rb").read() == open(os.path.join(outdir, f), "rb").read(), f
    hp.logdir = str(tmp_path / "nowhere" / "train")
    with pytest.raises(SystemExit, match="No t2m at"):
        S.synthesize(hp, num_sentences=2)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _short_texts(hp, lens=(2, 3, 4, 6, 8, 10)):
    """short texts so that attention reaches the end: shard 0 (first three) finishes long before shard 1"""
    rng = np.random.default_rng(1)
    L = np.zeros((len(lens), hp.max_N), np.int32)
    for i, n in enumerate(lens):
        L[i, :n] = rng.integers(1, len(hp.vocab), n)
    return L


def _shard_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)     # both ranks share the one GPU of the test box
    try:
        from ophelia_amd import parallel, synthesize as S
        from ophelia_amd.architectures import Session, Text2MelGraph
        from ophelia_amd.data_load import load_data
        hp = _hp(max_T=40)
        L = _short_texts(hp)
        lo, hi = parallel.shard_range(len(L), rank, world)
        W = O.random_weights(hp, 43) if rank == 0 else None
        with Session(hp, device=0) as sess:
            W = parallel.broadcast_weights(W, sess.inventory(), src=0)
            sess.assign(W)
            g = Text2MelGraph(hp, mode="synthesize")
            Ls = L[lo:hi]
            K, V = S.encode_text(hp, Ls, g, sess)
            Y, t_ends, al = S.synth_codedtext2mel(hp, K, V, S.get_text_lengths(Ls), g, sess)
        q.put((rank, Y, t_ends))
    finally:
        dist.destroy_process_group()


def test_sharded_decode_reproduces_global_stop():
    """2 ranks (gloo, sharing the GPU): shard outputs concatenated == one unsharded batch, including the
    frames an early-finishing shard generates while the other shard keeps the loop alive."""
    import torch.multiprocessing as mp
    from ophelia_amd.data_load import load_data
    hp = _hp(max_T=40)
    L = _short_texts(hp)
    W = O.random_weights(hp, 43)
    K, V = O.encode_text(hp, W, L)
    Y0, t0, _ = O.synth_codedtext2mel_incremental(hp, W, K, V, O.get_text_lengths(L))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted((q.get(timeout=300) for _ in range(2)), key=lambda r: r[0])
    for p in procs: p.join(timeout=60)
    Y = np.concatenate([r[1] for r in res]); t_ends = res[0][2] + res[1][2]
    assert t_ends == t0
    steps = [max(r[2]) + 1 for r in res]
    print("local stop steps", steps, "global", max(t0) + 1, "t_ends", t0)
    assert steps[0] < steps[1] <= hp.max_T          # the resume path of the early shard is exercised
    # frames of shard 0 between its own stop and the global stop exist only because shard 1 kept the loop alive
    assert np.abs(Y0[:3, steps[0]:steps[1]]).max() > 0
    assert np.abs(Y - Y0).max() < 1e-4
    assert not Y[:, steps[1]:].any()


def _dur_case(hp):
    """6 short texts with hard duration matrices: the first three (shard 0) are much shorter than the last three"""
    from ophelia_amd.data_load import durations_to_hard_attention_matrix, end_pad_for_reduction_shape_sync
    L = _short_texts(hp)
    rng = np.random.default_rng(9)
    D = np.zeros((len(L), hp.max_T, hp.max_N), np.int32)
    for b in range(len(L)):
        n = int(np.count_nonzero(L[b]))
        frames = (6 + b) * hp.r if b < 3 else (24 + 3 * b) * hp.r
        dur = np.zeros(n, np.int64)
        for _ in range(frames):
            dur[rng.integers(0, n)] += 1
        A = end_pad_for_reduction_shape_sync(durations_to_hard_attention_matrix(dur), hp)[0::hp.r]
        D[b, :len(A), :n] = A
    return L, D


def _shard_worker_durations(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ophelia_amd import parallel, synthesize as S
        from ophelia_amd.architectures import Session, Text2MelGraph
        hp = _hp(max_T=40)
        hp.use_external_durations = True
        L, D = _dur_case(hp)
        lo, hi = parallel.shard_range(len(L), rank, world)
        W = O.random_weights(hp, 43) if rank == 0 else None
        with Session(hp, device=0) as sess:
            W = parallel.broadcast_weights(W, sess.inventory(), src=0)
            sess.assign(W)
            g = Text2MelGraph(hp, mode="synthesize")
            K, V = S.encode_text(hp, L[lo:hi], g, sess)
            Y, t_ends, al = S.synth_codedtext2mel(hp, K, V, S.get_text_lengths(L[lo:hi]), g, sess, duration_data=D[lo:hi])
        q.put((rank, Y, t_ends))
    finally:
        dist.destroy_process_group()


def test_sharded_external_durations_run_to_the_global_longest():
    """with external durations every shard must run until the longest utterance of the WHOLE batch is through
    (synthesize.py:211-216), so the short shard keeps generating frames past its own utterances' ends"""
    import torch.multiprocessing as mp
    hp = _hp(max_T=40)
    hp.use_external_durations = True
    L, D = _dur_case(hp)
    W = O.random_weights(hp, 43)
    K, V = O.encode_text(hp, W, L)
    Y0, t0, _ = O.synth_codedtext2mel_incremental(hp, W, K, V, O.get_text_lengths(L), durations=D)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker_durations, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted((q.get(timeout=300) for _ in range(2)), key=lambda r: r[0])
    for p in procs: p.join(timeout=60)
    Y = np.concatenate([r[1] for r in res]); t_ends = res[0][2] + res[1][2]
    assert t_ends == t0 == D.sum(axis=(1, 2)).tolist()
    steps = min(hp.max_T, max(t0) + 1)
    assert max(t0[:3]) + 1 < steps                       # shard 0 alone would have stopped earlier
    assert np.abs(Y0[:3, max(t0[:3]) + 1:steps]).max() > 0
    assert np.abs(Y - Y0).max() < 1e-4 and not Y[:, steps:].any()


def test_cli_on_the_ten_line_test_transcript(tmp_path):
    """BASELINE configs[0] (C1) as the command line runs it: `python -m ophelia_amd.synthesize -c lj_test.cfg -N 10 -odir ...` on the
    10-line test transcript with lj_test.cfg's own dimensions (max_N 180, max_T 210), and the DEFAULT output set -- what the
    reference leaves in {odir}/{cfg}/t2m{E}_ssrn{E}/: one .wav per utterance (synthesize.py:433-438), one attention .png when
    matplotlib is there (:594-595), no .npy unless hp.store_synth_features (:436-437), nothing else."""
    import json
    import subprocess
    import sys
    import wave
    snap = json.load(open(os.path.join(GOLDEN, "config_snapshot.json")))["lj_test.cfg"]
    cfg = tmp_path / "lj_test.cfg"
    with open(cfg, "w") as f:                               # the reference's config format: an executable Python file
        for k, v in sorted(snap.items()):
            f.write("%s = %r\n" % (k, v))
        f.write("test_transcript = %r\nsampledir = %r\n" % (os.path.join(GOLDEN, "test_transcript_lj_test.csv"), str(tmp_path / "synth")))
        for k in ("topworkdir", "voicedir", "logdir", "datadir", "waveforms", "coarse_audio_dir", "full_audio_dir", "full_mel_dir", "attention_guide_dir"):
            f.write("%s = %r\n" % (k, str(tmp_path / "work" / k)))         # the paths the snapshot leaves out (machine-specific in the reference's file)
        f.write("transcript = %r\n" % str(tmp_path / "work" / "transcript.csv"))
    # (OPH_HANG_DUMP_S: a run that is still going after 4 minutes -- it takes seconds -- prints its Python stacks and exits non-zero)
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), OPH_HANG_DUMP_S="240")
    res = subprocess.run([sys.executable, "-m", "ophelia_amd.synthesize", "-c", str(cfg), "-N", "10", "-odir", str(tmp_path / "out"),
                          "-random_init", "5"], env=env, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:]
    assert "max_N=180" in res.stdout and "max_T=210" in res.stdout and "File |  CDP | Ain" in res.stdout
    outdir = tmp_path / "out" / "lj_test" / "t2mrand_ssrnrand"
    names = sorted(os.listdir(outdir))
    bases = [line.split("|")[0] for line in open(os.path.join(GOLDEN, "test_transcript_lj_test.csv")) if line.strip()]
    assert len(bases) == 10
    assert [n for n in names if n.endswith(".wav")] == sorted(b + ".wav" for b in bases)
    assert not [n for n in names if n.endswith(".npy")]     # lj_test.cfg does not set store_synth_features
    assert set(os.path.splitext(n)[1] for n in names) <= {".wav", ".png"}
    for b in bases:
        assert ("%s | " % b) in res.stdout                  # the per-utterance CDP / Ain report line
        with wave.open(str(outdir / (b + ".wav")), "rb") as f:
            assert (f.getnchannels(), f.getsampwidth(), f.getframerate()) == (1, 2, snap["sr"]) and f.getnframes() > 0
