#!/usr/bin/env python
"""bench.py -- mel-frames/sec of the Text2Mel + SSRN synthesis hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input:
encode_text -> 200-step autoregressive decode (fixed length, stop_mode=never) ->
SSRN, for 16 utterances per GPU (BASELINE.json configs[2]: lj_tutorial.cfg
Text2Mel+SSRN end-to-end, batch 16, Griffin-Lim off).  Two different seeded texts
alternate from step to step (nothing input-dependent can be reused across steps).

`value` is SURVEY 8(d)'s timed region -- the reference's own two clocks merged
(synthesize.py:553-576): host arrays in -> host arrays out, one oph_run_host call per
batch, strictly one batch after the other; every call returns with K, V, Y, t_ends,
alignments and Z of its batch in (pinned) host arrays, and the next batch's text goes
host -> device inside the timed region too.  Weights are resident before it starts.
Further legs, reported in `config`: the same batches with everything left in HBM and
consecutive batches pipelined (resident_value), strictly sequential in HBM, all-fp32,
and through the drop-in Python API (encode_text -> synth_codedtext2mel ->
synth_mel2mag, as synthesize.py:553-576 calls them).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  Multi-GPU = utterance sharding, no data-path
collective (weak scaling); RCCL is used once, to broadcast the weights from rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
PEAK_F32_MFMA_TFLOPS = 157.3  # dense fp32-input MFMA peak (v_mfma_f32_32x32x2_f32)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n, argv, device_check=True):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (torch.distributed.run, one process
    per GPU, rendezvous on 127.0.0.1) and relay rank 0's JSON line.  Never reports fewer GPUs than asked for: if the
    node has fewer than N devices (or a rank dies) this fails with a non-zero exit code."""
    import subprocess
    if device_check:
        import torch
        have = torch.cuda.device_count()
        if have < n:
            raise SystemExit("--gpus %d requested but this node exposes %d GPU(s)" % (n, have))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    res = subprocess.run(cmd, env=env)
    raise SystemExit(res.returncode)


def pin_rank_to_cores(local_rank, world):
    """Host pre-flight of a multi-rank run.  A rank needs most of one core for the thread that enqueues the cone's
    launches plus room for the HIP runtime's and RCCL's helper threads: every rank gets its own contiguous
    slice of the cores this process may run on (threads created later inherit it), so that eight ranks never compete for one
    core.  Returns {"cores": [...], "nproc": n, "warning": str or None}; nothing is pinned at world 1 or when the node has fewer
    cores than ranks (the warning says so)."""
    try:
        avail = sorted(os.sched_getaffinity(0))
    except AttributeError:                     # not Linux
        return {"cores": None, "nproc": os.cpu_count(), "warning": None}
    n = len(avail)
    info = {"cores": None, "nproc": n, "warning": None}
    if world <= 1:
        return info
    if n < 2 * world:
        info["warning"] = ("%d host cores for %d ranks: a rank keeps ~0.9 of a core busy enqueuing and needs a second one for the "
                           "runtime's threads; expect host-bound steps (need >= %d cores)" % (n, world, 2 * world))
    if n < world:
        return info
    per = n // world
    mine = avail[local_rank * per:(local_rank + 1) * per]
    try:
        os.sched_setaffinity(0, mine)
        info["cores"] = mine
    except OSError as e:
        info["warning"] = ((info["warning"] + "; ") if info["warning"] else "") + "sched_setaffinity failed: %s" % e
    return info


def thread_cpu_seconds():
    """{tid: (comm, user + system seconds)} of this process's threads (Linux /proc): which thread burns the host CPU a rank needs"""
    out = {}
    try:
        tick = os.sysconf("SC_CLK_TCK")
        for tid in os.listdir("/proc/self/task"):
            try:
                f = open("/proc/self/task/%s/stat" % tid).read()
                comm = f[f.index("(") + 1:f.rindex(")")]
                rest = f[f.rindex(")") + 2:].split()
                out[int(tid)] = (comm, (int(rest[11]) + int(rest[12])) / float(tick))
            except (OSError, ValueError, IndexError):
                pass
    except (OSError, ValueError, AttributeError):
        pass
    return out


class HP(object):
    pass


def load_hp(cfg="lj_tutorial.cfg"):
    """Hyper-parameters of the reference config, from the committed snapshot fixture
    (generated by tests/golden/make_golden.py with the reference's own load_config)."""
    snap = json.load(open(os.path.join(ROOT, "tests", "golden", "config_snapshot.json")))[cfg]
    hp = HP()
    for k, v in snap.items():
        setattr(hp, k, v)
    return hp


def synth_text(hp, B, seed, min_len=75, max_len=149):
    rng = np.random.Generator(np.random.PCG64(seed))
    V = len(hp.vocab)
    L = np.zeros((B, hp.max_N), np.int32)
    for b in range(B):
        n = int(rng.integers(min_len, max_len + 1))
        L[b, :n] = rng.integers(1, V, size=n)
    ends = np.array([np.where(L[i] == 0)[0][0] for i in range(B)], np.int32)
    return L, ends


def fallback_legs(hp, texts, B, device, steps=4):
    """frames/s of the launch paths a configuration outside dec_chain's geometry takes: the generic
    whole-decode kernel (dec_loop), the per-step paths (two launches per step / one launch per layer), a model without LayerNorm
    (hp.norm = None: dec_loop, unfused cone), and a decode on external durations (hp.use_external_durations: per-step paths with
    the attention replaced by the given selection).  Sequential resident batches, B utterances, fixed length; a fresh handle per
    leg (the launch path is chosen at oph_create), sharing the device's CU-masked streams with the main one."""
    import copy
    from ophelia_amd.engine import Engine
    from ophelia_amd import weights as WT
    out = {}

    def leg(name, options, hp_leg, durations=False):
        eng = None
        try:
            eng = Engine(hp_leg, device=device, options=options)
            eng.load_weights(WT.random_weights(eng.inventory(), seed=2))
            L, ends = texts[0]
            if durations:
                K, V = eng.encode_text(L)
                D = np.zeros((B, hp_leg.max_T, hp_leg.max_N), np.float32)      # a monotonic walk over the text, one key per step
                for b in range(B):
                    D[b, np.arange(hp_leg.max_T), np.minimum(np.arange(hp_leg.max_T) * int(ends[b]) // hp_leg.max_T, int(ends[b]) - 1)] = 1.0
                run = lambda: eng.text2mel_durations(K, V, D)
            else:
                eng.stage_text(L, ends)
                run = lambda: eng.run_resident(stop_mode=1, run_ssrn=True, pipelined=False)
            run(); eng.synchronize()
            c0 = eng.counters()
            t0 = time.perf_counter()
            for _ in range(steps):
                run()
            eng.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            c1 = eng.counters()
            out[name] = {"ms_per_batch": ms, "value": B * hp_leg.max_T / (ms * 1e-3), "whole_decode_launches": c1["loop_decodes"] - c0["loop_decodes"],
                         "recoveries": c1["recoveries"] - c0["recoveries"], "includes_ssrn": not durations}
        except Exception as e:                 # an untimed diagnostic leg never takes the line down
            out[name] = {"error": str(e)[:200]}
        finally:
            if eng is not None:
                eng.close()

    hp_nn = copy.copy(hp)
    hp_nn.norm = None
    leg("dec_loop (option NO_CHAIN: the generic whole-decode kernel, this model)", {"NO_CHAIN": 1}, hp)
    leg("two launches per step (option DECODE=runs, this model)", {"DECODE": "runs"}, hp)
    leg("one launch per layer (option DECODE=layers, this model)", {"DECODE": "layers"}, hp)
    leg("hp.norm = None (no LayerNorm anywhere: dec_loop, the cone's levels as contraction + epilogue launches)", {}, hp_nn)
    leg("external durations (text2mel only: attention replaced by the given selection, per-step launches)", {}, hp, durations=True)
    return out


def cpu_baseline(hp, W, L, ends, budget_s=25.0):
    """Reference-faithful CPU port (oracle/oph_cpu.c, OpenMP) timed on a bounded sample."""
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    from oracle import cpu_oracle
    cores = min(cpu_oracle.usable_cores(), 64)   # the port's row-blocked GEMMs stop scaling past ~64 threads
    m = cpu_oracle.CpuModel(hp, W, threads=cores)
    B = len(L)
    t0 = time.perf_counter()
    K, V = m.encode_text(L)
    t_enc = time.perf_counter() - t0
    t0 = time.perf_counter()
    m.text2mel(K, V, ends, stop=False, max_steps=1)          # probe: one reference step
    t1 = time.perf_counter() - t0
    # the WHOLE decode (max_T reference steps, ~11 s on 16 cores) when it fits the budget; a slower host runs what fits and says so
    budget_steps = int(max(1, min(hp.max_T, budget_s / max(t1, 1e-3))))
    if budget_steps < hp.max_T:
        budget_steps = min(budget_steps, 20)
    t0 = time.perf_counter()
    Y, _, _, steps = m.text2mel(K, V, ends, stop=False, max_steps=budget_steps)
    t_dec = time.perf_counter() - t0
    t0 = time.perf_counter()
    m.ssrn(Y)
    t_ssrn = time.perf_counter() - t0
    total = t_enc + t_ssrn + t_dec / steps * hp.max_T
    inc = cpu_incremental(hp, W, L, ends, cores, t_enc, t_ssrn)
    sampled = steps < hp.max_T
    return {
        "value": B * hp.max_T / total, "unit": "frames/s", "cores": int(m.threads), "kind": "port", "sampled": bool(sampled),
        "sample": "B=%d: full TextEnc (%.2fs) + full SSRN (%.2fs) + %d of %d reference decode steps (%.2fs), "
                  "each step recomputing all max_T positions as synthesize.py:181-183 does (constant cost per step)%s"
                  % (B, t_enc, t_ssrn, steps, hp.max_T, t_dec, (", extrapolated to %d steps" % hp.max_T) if sampled else ": the whole batch, nothing extrapolated"),
        "incremental": inc,
    }


def cpu_incremental(hp, W, L, ends, cores, t_enc, t_ssrn, n_steps=96):
    """The same outputs computed the O(T) way on the CPU (NumPy/BLAS restatement of the exact incremental
    algorithm the GPU path uses: AudioEnc cached, AudioDec re-evaluated over its 85-frame receptive field),
    so the GPU/CPU ratio can be split into an algorithmic and a hardware factor.  Bounded sample: the first
    n_steps steps are run; steps >= 85 have the full-window (steady-state) cost and are extrapolated."""
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:
        threadpool_limits = None
    from oracle import ophelia_oracle as O
    times = []
    K, V = O.encode_text(hp, W, L[:1])          # warm BLAS; TextEnc/SSRN times are taken from the C port above
    del K, V

    def run():
        K, V = O.encode_text(hp, W, L)
        O.synth_codedtext2mel_incremental(hp, W, K, V, ends, stop=False, max_steps=n_steps, step_times=times)
    if threadpool_limits is not None:
        with threadpool_limits(limits=cores):
            run()
    else:
        run()
    steady = float(np.mean(times[86:])) if len(times) > 90 else float(np.mean(times[-5:]))
    t_dec = float(np.sum(times)) + steady * (hp.max_T - len(times))
    total = t_enc + t_ssrn + t_dec
    return {"value": len(L) * hp.max_T / total, "unit": "frames/s", "cores": int(cores), "kind": "port",
            "sample": "exact incremental algorithm (oracle/ophelia_oracle.py, NumPy/BLAS): first %d of %d decode steps run "
                      "(%.2fs), steady-state step %.1f ms extrapolated; TextEnc/SSRN times of the C port" %
                      (len(times), hp.max_T, float(np.sum(times)), steady * 1e3)}


def vocoder_leg(hp, eng, B, cpu):
    """Griffin-Lim on the SSRN output still resident in HBM -- NOT part of `value` (the metric is
    quoted with Griffin-Lim off); reported so that the end-to-end cost of a batch including the waveform is on record."""
    from ophelia_amd.vocoder import Vocoder
    n_frames = np.full(B, hp.max_T * hp.r, np.int32)
    res = {"workload": "griffin_lim n_iter=%d, n_fft=%d hop=%d win=%d, %d utterances x %d frames, de-emphasis included"
                       % (hp.n_iter, hp.n_fft, hp.hop_length, hp.win_length, B, hp.max_T * hp.r)}
    with Vocoder(hp, device=int(os.environ.get("LOCAL_RANK", "0"))) as voc:
        for name, backend in (("fused", 0), ("hipfft", 1)):
            try:
                voc.set_backend(backend)
                voc.spectrogram2wav_from_engine(eng, n_frames)
                ms = []
                for _ in range(3):
                    t = time.perf_counter()
                    voc.spectrogram2wav_from_engine(eng, n_frames)
                    ms.append(((time.perf_counter() - t) * 1e3, voc.last_device_ms()))
                res[name + "_device_ms_per_batch"] = min(m[1] for m in ms)
                res[name + "_wall_ms_per_batch_incl_wav_download"] = min(m[0] for m in ms)
            except Exception as e:            # the comparison backend needs rocFFT's run-time kernel compilation, which fails on some boxes
                res[name + "_error"] = str(e)[:200]
    if "fused_device_ms_per_batch" not in res:
        return res
    res["spectrogram_frames_per_s"] = B * hp.max_T * hp.r / (res["fused_device_ms_per_batch"] * 1e-3)
    # dominant kernel gl_fused<false>: per frame and iteration it must read S (nbin floats) and the signal under the
    # window (win floats) and write the windowed segment (win floats)
    alg = B * hp.max_T * hp.r * 4.0 * (hp.n_fft // 2 + 1 + 2 * hp.win_length)
    res["gl_fused_algorithmic_bytes_per_launch"] = alg
    if cpu:
        sys.path.insert(0, ROOT)
        from oracle import griffin_lim_oracle as GL
        Z = eng.fetch_mag()
        T = 120                                        # bounded sample: one utterance's first 120 frames, all 50 iterations
        t = time.perf_counter()
        GL.spectrogram2wav(hp, Z[0, :T])
        dt = time.perf_counter() - t
        res["cpu_baseline"] = {"value": T / dt, "unit": "spectrogram frames/s", "cores": 1, "kind": "port",
                               "sample": "oracle/griffin_lim_oracle.py (NumPy restatement of librosa 0.6.2 + utils.py:69-116) "
                                         "on utterance 0's first %d frames, %d iterations" % (T, hp.n_iter)}
    return res


def pmc_traffic(kernel_class):
    """HBM bytes per launch of a kernel class from the committed rocprofv3 --pmc summary of this
    same command (profiles/rNN_traffic.json, written by profiles/summarize.py); None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic.json")))     # hot path only (not r01_vocoder_*)
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    n = b = 0.0
    for k, v in d["kernels"].items():
        kk = k.replace(" ", "")
        if kk.endswith(kernel_class) or kk.split("<")[0].endswith(kernel_class):
            n += v["launches"]
            b += v["launches"] * v["hbm_bytes_per_launch"]
    if n == 0:
        return None, None
    return b / n, "%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, %s" % (os.path.basename(files[-1]), d["correction"])


LOOP_KERNEL = "dec_chain (whole-decode launch; profile class dec_loop)"
ROOF_SOURCE = ("frac = algorithmic bytes per launch (every step's weights + activations, SURVEY 8d) / avg_launch_us, where avg_launch_us comes "
               "from one HIP event pair around every whole-decode launch of the timed region, on its launch stream; device_clock_us is the "
               "same launches timed by the kernel itself (first workgroup in -> last workgroup out on the 100 MHz s_memrealtime clock, "
               "oph_loop_clock) -- independent of HIP events and of any profiler; profiles/ holds the rocprofv3 kernel trace of the same command")


def note(msg):
    """progress on stderr (OPH_BENCH_VERBOSE=1): tells which leg a stuck run is in"""
    if os.environ.get("OPH_BENCH_VERBOSE"):
        sys.stderr.write("[bench %.3f] %s\n" % (time.perf_counter(), msg))
        sys.stderr.flush()


def main():
    # a run that is still going after OPH_HANG_DUMP_S seconds (default 30 minutes; the default command takes about two) dumps every
    # thread's Python stack to stderr and exits: a stuck device call names itself instead of sitting there
    import faulthandler
    faulthandler.dump_traceback_later(float(os.environ.get("OPH_HANG_DUMP_S", "1800")), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16, help="utterances per GPU")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="only the timed region (no sequential / all-fp32 / host-to-host legs): what profiles/collect_rNN.sh traces, so that "
                         "every dec_loop launch in the rocprofv3 summary belongs to the measured configuration")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-vocoder", action="store_true", help="skip the (untimed) Griffin-Lim leg")
    ap.add_argument("--spawn-selftest", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-pipeline", action="store_true",
                    help="the resident leg (config.resident_value) with strictly sequential batches (default: SSRN of batch i overlaps "
                         "TextEnc+decode of batch i+1); `value` is always sequential host -> host")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if launched and world != a.gpus:
        raise SystemExit("--gpus %d but the launcher started WORLD_SIZE=%d ranks" % (a.gpus, world))
    # OPH_BENCH_SHARED_GPU=1 (tests on a 1-GPU box): every rank drives GPU 0 and the process group is gloo (RCCL refuses two
    # ranks on one device); the JSON line says so.  Never set by the driver.
    shared_gpu = os.environ.get("OPH_BENCH_SHARED_GPU") == "1"
    if not launched and a.gpus > 1:
        spawn_ranks(a.gpus, sys.argv[1:], device_check=not (a.spawn_selftest or shared_gpu))      # does not return
    dist = None
    host = pin_rank_to_cores(local_rank, world)          # before any runtime thread exists: they inherit the rank's core slice
    if host["warning"] and rank == 0:
        sys.stderr.write("bench.py: " + host["warning"] + "\n")
    if a.spawn_selftest:
        # CPU-only check of the rank plumbing (tests/test_bench_spawn.py): gloo instead of RCCL, no engine -- ports, the rank ->
        # device map the real run uses (LOCAL_RANK), the per-rank core slices, one line from rank 0
        import torch
        import torch.distributed as dist
        mine = {"rank": rank, "local_rank": local_rank, "device": 0 if shared_gpu else local_rank, "cores": host["cores"], "pid": os.getpid()}
        if launched:
            dist.init_process_group("gloo")
            t = torch.tensor([float(rank + 1)], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ranks = [None] * world
            dist.all_gather_object(ranks, mine)
            dist.barrier()
            mx = float(t.item())
            dist.destroy_process_group()
        else:
            mx, ranks = 1.0, [mine]
        if rank == 0:
            print(json.dumps({"selftest": True, "n_gpus": world, "max_rank_plus_1": mx, "ranks": ranks, "host_nproc": host["nproc"],
                              "host_warning": host["warning"]}))
        return
    if launched:                                 # one rank per GPU over RCCL (torch.distributed.run, ours or the driver's)
        import torch
        import torch.distributed as dist
        if shared_gpu:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # Several ranks on one node compete for host cores, and with events the decode loop needs ~94 % of one core just to
    # enqueue: multi-rank runs use the cheaper stream write/wait-value dependencies -- unless a
    # profiler is attached (rocprofv3 --pmc deadlocks on wait-value packets) or the caller already chose.
    profiled = any(k.startswith(("ROCPROF", "ROCP_")) or k == "HSA_TOOLS_LIB" for k in os.environ)
    # The library reads no environment variable for its launch paths: the bench's own switches become oph_create_opts options.
    #   OPH_BENCH_OPTIONS="NAME=value ..."   options for the main handle (measurement scripts under profiles/)
    engine_options = dict(tok.split("=", 1) if "=" in tok else (tok, "1") for tok in os.environ.get("OPH_BENCH_OPTIONS", "").split())
    if world > 1 and not profiled:
        engine_options.setdefault("STREAM_VALUE", "1")
    if any(k.startswith("ROCPROF") for k in os.environ) and os.environ.get("OPH_BENCH_PMC"):
        # counter-collection passes serialise dispatches across queues: the whole-decode launch would wait for a side
        # stream that cannot start.  The collect scripts set OPH_BENCH_PMC=1 for those passes and OPH_HIPCC_FLAGS=-DOPH_ABLATE (a measurement
        # build of the library: a production build refuses the option): the decode kernel then runs without its cone (identical memory
        # traffic of its own; the mel values are not used).
        engine_options.setdefault("LOOP_ALONE", "1")

    from ophelia_amd.engine import Engine
    from ophelia_amd import weights as WT
    from ophelia_amd import parallel

    hp = load_hp()
    B = a.batch
    eng = Engine(hp, device=local_rank, options=engine_options)
    inv = eng.inventory()
    W = WT.random_weights(inv, seed=2) if rank == 0 else None
    if dist is not None:
        # one flat fp32 tensor root -> peers (RCCL over xGMI; gloo in the shared-GPU test mode); every rank hands its receive buffer to
        # the library on the device (oph_set_weights_device): repacked by device kernels, no host round trip.  Start-up only.
        # the process group must be what the command line asked for: one rank per GPU, `--gpus` of them, on RCCL
        if dist.get_world_size() != a.gpus:
            raise SystemExit("--gpus %d but torch.distributed reports a world of %d" % (a.gpus, dist.get_world_size()))
        if not shared_gpu and dist.get_backend() != "nccl":
            raise SystemExit("multi-GPU runs use RCCL (backend nccl); the process group reports %r" % dist.get_backend())
        collective = parallel.load_weights_broadcast(eng, W, src=0, device=None if shared_gpu else local_rank)
        if rank != 0:
            W = None
    else:
        eng.load_weights(W)
        collective = None

    # utterance shard of this rank: global batch = world*B utterances, contiguous shards.  Two different texts
    # (seeds 3 and 4) alternate, so that no step can reuse anything computed from the previous step's input.
    texts = []
    for seed in (3, 4):
        Lg, endsg = synth_text(hp, world * B, seed=seed)
        texts.append((Lg[rank * B:(rank + 1) * B], endsg[rank * B:(rank + 1) * B]))
    L, ends = texts[0]
    state = {"i": 0}

    def restage():
        """text 0 staged, text 1 staged behind it (both resident in HBM before a timed region starts)"""
        eng.synchronize()
        eng.stage_text(*texts[0])
        eng.stage_text_next(*texts[1])
        state["i"] = 0

    restage()
    pipelined = not a.no_pipeline

    def one_step(pipe=None):
        k = state["i"] & 1
        steps = eng.run_resident(stop_mode=1, run_ssrn=True, pipelined=pipelined if pipe is None else pipe)
        assert steps == hp.max_T
        eng.stage_text_next(*texts[k])          # text k has run: text 1-k becomes current (pre-encoded meanwhile), k is staged behind it
        state["i"] += 1

    def h2h_step():
        """host arrays in -> host arrays out (the reference's two clocks, synthesize.py:553-576): the staged text through TextEnc,
        the decode loop and SSRN; K, V, Y, t_ends, alignments, Z are in pinned host arrays when the call returns; then the text
        of the batch after next goes host -> device (oph_stage_text_next)."""
        k = state["i"] & 1
        out = eng.run_host(stop_mode=1, want_kv=True)
        assert out["steps"] == hp.max_T
        eng.stage_text_next(*texts[k])
        state["i"] += 1
        return out

    note("weights loaded, texts staged")
    for _ in range(a.warmup):
        h2h_step()
    eng.synchronize()
    note("warm-up done")
    if dist is not None:
        dist.barrier()
    eng.synchronize()
    eng.profile_reset()
    eng.profile_enable(2)                       # one HIP event pair per whole-decode launch, on the stream it is launched on
    eng.loop_clock(reset=True)                  # ... and the kernel's own first-in / last-out clock stamps
    t0 = time.perf_counter()
    cpu0 = time.process_time()
    thr0 = thread_cpu_seconds()
    for _ in range(a.steps):
        h2h_step()
    eng.synchronize()
    host_cpu_s = time.process_time() - cpu0         # host CPU this rank burnt driving its GPU (all threads of the process)
    thr1 = thread_cpu_seconds()
    t_region = time.perf_counter() - t0
    # ... and which threads: [name, cores] of the busiest four (the enqueue thread is this one, "python"; the rest belong to the runtime)
    thr_top = sorted(((c, round((s - thr0.get(tid, (c, 0.0))[1]) / t_region, 3)) for tid, (c, s) in thr1.items()), key=lambda x: -x[1])[:4]
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    note("timed region done")
    loop_prof = next((p for p in eng.profile() if p["name"] == "dec_loop" and p["launches"] > 0), None)
    clk_n, clk_us = eng.loop_clock()
    eng.profile_enable(False)
    rank_ms = [dt / a.steps * 1e3]
    rank_cpu = [host_cpu_s / dt]
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if shared_gpu else "cuda")
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        rank_ms = [float(x.item()) / a.steps * 1e3 for x in allt]        # per-rank step time: a straggler shows up here
        c = torch.tensor([host_cpu_s / dt], dtype=torch.float64, device="cpu" if shared_gpu else "cuda")
        allc = [torch.zeros_like(c) for _ in range(world)]
        dist.all_gather(allc, c)
        rank_cpu = [float(x.item()) for x in allc]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # who ran where, and whether any rank had to redo a tile after a co-residency time-out (oph_get_counters[9])
        import socket
        mine = {"rank": rank, "local_rank": local_rank, "device": eng.device, "host": socket.gethostname(), "pid": os.getpid(),
                "gpu": torch.cuda.get_device_name(eng.device) if torch.cuda.is_available() else None,
                "gpu_uuid": str(getattr(torch.cuda.get_device_properties(eng.device), "uuid", "")) if torch.cuda.is_available() else None,
                "recoveries": eng.counters()["recoveries"], "masked_streams": eng.counters()["masked_streams"],
                "degraded_left": eng.counters()["degraded_left"], "host_threads": thr_top, "cores": host["cores"] and [host["cores"][0], host["cores"][-1]]}
        rank_info = [None] * world
        dist.all_gather_object(rank_info, mine)
    else:
        rank_info = [{"rank": 0, "local_rank": local_rank, "device": eng.device, "recoveries": eng.counters()["recoveries"],
                      "masked_streams": eng.counters()["masked_streams"], "host_threads": thr_top}]

    # Watchdog over everything that follows (the supplementary legs, the per-class profile, the vocoder leg, the CPU baseline): the
    # timed region is over and its numbers are final, so a leg that gets stuck must not cost the run its line -- after
    # OPH_BENCH_WATCHDOG_S seconds (default 900) rank 0 prints the line with what it has ("incomplete" says why) and leaves.
    watchdog = None
    if rank == 0:
        import threading
        core = {"metric": "mel-frames/sec (Text2Mel+SSRN) at batch 16", "value": world * B * hp.max_T * a.steps / dt, "unit": "frames/s",
                "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "rank_ms_per_step": rank_ms,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32 (fp16x3-split products in the batched contractions)", "data": "synthetic",
                "config": {"workload": "lj_tutorial.cfg Text2Mel+SSRN end-to-end, batch %d per GPU, max_N=150, max_T=200, fixed-length decode "
                                       "(stop_mode=never), Griffin-Lim off, seeded random-init weights" % B,
                           "timed_region": "host arrays in -> host arrays out, one oph_run_host call per batch (SURVEY 8d)",
                           "rank_host_cores": rank_cpu, "utterances_per_gpu": B, "global_utterances": world * B},
                "cpu_baseline": None, "incomplete": "watchdog: a leg after the timed region did not finish; value / roofline are those of the timed region"}
        if loop_prof is not None:
            avg_s = loop_prof["total_ms"] * 1e-3 / loop_prof["launches"]
            ach = loop_prof["alg_bytes"] / loop_prof["launches"] / avg_s / 1e9
            core["roofline"] = {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS, "traffic": None,
                                "kernel": LOOP_KERNEL, "avg_launch_us": avg_s * 1e6, "device_clock_us": (clk_us / clk_n) if clk_n else None,
                                "algorithmic_per_launch": loop_prof["alg_bytes"] / loop_prof["launches"], "source": ROOF_SOURCE}

        def fire():
            sys.stdout.write(json.dumps(core) + "\n")
            sys.stdout.flush()
            os._exit(0)
        watchdog = threading.Timer(float(os.environ.get("OPH_BENCH_WATCHDOG_S", "900")), fire)
        watchdog.daemon = True
        watchdog.start()

    # the same batches with inputs and outputs left in HBM, consecutive batches pipelined (SSRN tail of batch i and TextEnc of the
    # staged text i+1 on the SSRN partition under decode i+1): what rounds 1-3 reported as `value`
    res_steps = max(2, min(a.steps, 20))
    res_ms = None
    if not a.no_extra_legs:
        restage(); one_step(); one_step(); eng.synchronize()
        t1 = time.perf_counter()
        eng.timer_start()
        for _ in range(res_steps):
            one_step()
        eng.timer_stop()
        eng.synchronize()
        res_ms = (time.perf_counter() - t1) / res_steps * 1e3
        note("resident leg done")
    seq_ms = None
    if pipelined and not a.no_extra_legs:          # the same batches strictly one after the other (every call returns with its SSRN done)
        one_step(False); eng.synchronize()
        t1 = time.perf_counter()
        for _ in range(res_steps):
            one_step(False)
        eng.synchronize()
        seq_ms = (time.perf_counter() - t1) / res_steps * 1e3
        note("sequential leg done")
    fp32_ms = api_ms = None
    # the same run with every contraction on the fp32-operand MFMA (no fp16 splitting anywhere)
    if not a.no_extra_legs:
        for which in ("ssrn", "cone", "textenc"):
            eng.set_precision(which, 0)
        # (sequential batches: fp32-operand SSRN does not fit under a decode on its 64-CU partition -- 34 ms against 23 -- so
        #  it streams there as far as it keeps up and finishes on the whole chip when the decode is over)
        one_step(False); one_step(False); eng.synchronize()
        t2 = time.perf_counter()
        for _ in range(res_steps):
            one_step(False)
        eng.synchronize()
        fp32_ms = (time.perf_counter() - t2) / res_steps * 1e3
        note("fp32 leg done")
        for which in ("ssrn", "cone", "textenc"):
            eng.set_precision(which, 2)
        one_step(False); eng.synchronize()
    h2h_steps = max(2, min(a.steps, 20))
    # the drop-in Python API, called as the reference's synthesize() calls it (synthesize.py:553-576): three session calls
    # per batch on host arrays, a different text every batch.  hp.synth_stop_mode = 1 is this package's fixed-length switch
    # (every utterance runs max_T steps, the configuration `value` is quoted on); everything else is the reference's call
    # sequence.  K, V and Y stay resident between the calls, SSRN streams under the decode.
    if not a.no_extra_legs:
        from ophelia_amd import synthesize as SY
        from ophelia_amd.architectures import Session, SSRNGraph, Text2MelGraph
        hp_api = load_hp()
        hp_api.synth_stop_mode = 1
        g1, g2 = Text2MelGraph(hp_api, mode="synthesize"), SSRNGraph(hp_api, mode="synthesize")
        sess = Session(hp_api, device=local_rank, engine=eng)      # the same handle (one GPU = one handle: a second one would take its own CU partitions)

        def api_step(k):
            Lk, _ = texts[k]
            text_lengths = SY.get_text_lengths(Lk)
            K, V = SY.encode_text(hp_api, Lk, g1, sess)
            Y, lengths, alignments = SY.synth_codedtext2mel(hp_api, K, V, text_lengths, g1, sess)
            Z = SY.synth_mel2mag(hp_api, Y, g2, sess)
            assert Z.shape == (B, hp.max_T * hp.r, hp.full_dim) and len(lengths) == B
        eng.synchronize(); api_step(0); api_step(1)
        t4 = time.perf_counter()
        for i in range(h2h_steps):
            api_step(i & 1)
        api_ms = (time.perf_counter() - t4) / h2h_steps * 1e3
        note("api leg done")
        restage()
    frames = world * B * hp.max_T * a.steps
    out = {
        "metric": "mel-frames/sec (Text2Mel+SSRN) at batch 16", "value": frames / dt, "unit": "frames/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "rank_ms_per_step": rank_ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        # fp32 data and fp32 accumulation everywhere.  The decoder chain (dec_loop) multiplies on the fp32-operand MFMA; the large
        # batched contractions (SSRN, TextEnc, the two many-row levels of the AudioDec cone) form each fp32 x fp32 product from
        # fp16 hi + lo terms (22 significant bits per operand, three fp16 MFMA products, fp32 accumulate): measured in the
        # accuracy class of the fp32-operand MFMA (config.arithmetic).  config.all_fp32_value: the same run with those
        # contractions on the fp32-operand MFMA too.
        "dtype": "f32 (fp16x3-split products in the batched contractions)",
        "data": "synthetic",
        "config": {"workload": "lj_tutorial.cfg Text2Mel+SSRN end-to-end, batch %d per GPU, max_N=150, max_T=200, "
                               "fixed-length decode (stop_mode=never), Griffin-Lim off, seeded random-init weights" % B,
                   "rank_host_cores": rank_cpu,      # host CPU seconds per wall second of the timed region, per rank (1.0 = one core busy)
                   "host": {"nproc": host["nproc"], "rank0_cores": host["cores"], "warning": host["warning"],
                            "note": "multi-rank runs give every rank its own contiguous slice of the host cores (sched_setaffinity before any "
                                    "runtime thread exists) and warn below 2 cores per rank"},
                   "utterances_per_gpu": B, "global_utterances": world * B, "frames_per_step": world * B * hp.max_T,
                   "parallelism": "utterance-shard x%d" % world + (" (test mode: all ranks on one GPU, gloo)" if shared_gpu else ""),
                   # the start-up weight broadcast as the process group reports it (None at N = 1: no collective anywhere on the path)
                   "collective": collective, "ranks": rank_info, "recoveries": [r["recoveries"] for r in rank_info],
                   "timed_region": "SURVEY 8(d) / synthesize.py:553-576: host arrays in -> host arrays out, one oph_run_host call per batch, "
                                   "strictly sequential: each call returns with K, V, Y, t_ends, alignments, Z of its batch in pinned host "
                                   "arrays (Z leaves chunk by chunk under the running decode); the next text's H2D is inside the region; "
                                   "two seeded texts alternate; SSRN streams on its own CU partition under the decode that feeds it, the "
                                   "TextEnc of the staged next text runs there too",
                   "resident_ms_per_step": res_ms, "resident_value": (world * B * hp.max_T / (res_ms * 1e-3)) if res_ms else None,
                   "resident_note": ("inputs staged and outputs left in HBM, consecutive batches pipelined (the SSRN tail of batch i under decode "
                                     "i+1); all batches complete inside the timed region; %d batches -- the figure rounds 1-3 reported as `value`" % res_steps)
                                    if pipelined else "inputs staged and outputs left in HBM, sequential batches; %d batches" % res_steps,
                   "cross_stream_sync": "stream value operations" if engine_options.get("STREAM_VALUE") == "1" else "in-kernel words (loop) / events (per-step paths)",
                   "arithmetic": "fp32 in, fp32 accumulate; SSRN / TextEnc / the cone's two many-row levels: every operand as hi+lo fp16 "
                                 "(22 of 24 significant bits), 3 fp16 MFMA products per fp32 product -- measured (profiles/r03_prec.py): mag vs "
                                 "the CPU oracle 5.0e-6 (fp32-operand MFMA: 4.4e-6; split-bf16: 2.9e-5; bar 1e-3); mel vs the fp32-MFMA decode "
                                 "6.7e-6 with identical attention traces (two fp32-MFMA decode flavours among themselves: 5.6e-6); "
                                 "oph_set_precision (or the options SSRN_PREC, CONE_PREC, TEXTENC_PREC = 0 of oph_create_opts) selects the fp32-operand MFMA",
                   "decode_mode": engine_options.get("DECODE", "loop (whole decode in one launch; falls back to two launches per step when "
                                                                 "the batch does not fit the critical stream's CUs)"),
                   "api_ms_per_step": api_ms, "api_value": (world * B * hp.max_T / (api_ms * 1e-3)) if api_ms else None,
                   "api_note": "ophelia_amd.synthesize.encode_text -> synth_codedtext2mel -> synth_mel2mag per batch on host arrays, a "
                               "different text each batch, as synthesize.py:553-576 calls them (hp.synth_stop_mode=1: fixed length); %d batches" % h2h_steps,
                   "all_fp32_ms_per_step": fp32_ms, "all_fp32_value": (world * B * hp.max_T / (fp32_ms * 1e-3)) if fp32_ms else None,
                   "sequential_ms_per_step": seq_ms,
                   "sequential_value": (world * B * hp.max_T / (seq_ms * 1e-3)) if seq_ms else None},
    }

    if rank == 0 and loop_prof is not None and a.no_profile:
        # dominant kernel = the whole-decode launch, timed inside the timed region (the per-class leg below is skipped)
        avg_s = loop_prof["total_ms"] * 1e-3 / loop_prof["launches"]
        ach = loop_prof["alg_bytes"] / loop_prof["launches"] / avg_s / 1e9
        traffic, tsrc = pmc_traffic("dec_chain")
        if traffic is None:
            traffic, tsrc = pmc_traffic("dec_loop")
        out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS,
                           "traffic": traffic, "traffic_source": tsrc, "kernel": LOOP_KERNEL, "launches_per_step": 1,
                           "avg_launch_us": avg_s * 1e6, "algorithmic_per_launch": loop_prof["alg_bytes"] / loop_prof["launches"],
                           "device_clock_us": (clk_us / clk_n) if clk_n else None, "device_clock_launches": clk_n,
                           "source": ROOF_SOURCE}
    if rank == 0 and not a.no_profile:
        # per-kernel-class HIP-event accounting on the launch stream: one extra (untimed) step
        eng.profile_reset()
        eng.profile_enable(True)
        one_step(False)
        prof = eng.profile()
        eng.profile_enable(False)
        tot = sum(p["total_ms"] for p in prof) or 1.0
        dom = max(prof, key=lambda p: p["total_ms"])
        avg_s = dom["total_ms"] * 1e-3 / max(dom["launches"], 1)
        timed = False
        if loop_prof is not None:       # (one launch spans the whole decode of a batch: it is the dominant kernel whatever the per-class sums of the extra step say)
            # the whole-decode launch: its duration comes from the TIMED region (one event pair per launch); the extra step
            # brackets every side-stream launch with events, which slows the cone the loop kernel waits for
            dom = dict(loop_prof)
            avg_s = dom["total_ms"] * 1e-3 / dom["launches"]
            timed = True
        if dom["name"].startswith(("conv_gemm_f32", "cone_gemm_ln")):
            ach = dom["alg_flops"] / dom["launches"] / avg_s / 1e12
            roof = {"bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / PEAK_F32_MFMA_TFLOPS}
        else:
            ach = dom["alg_bytes"] / dom["launches"] / avg_s / 1e9
            roof = {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS}
        traffic, tsrc = pmc_traffic("dec_chain" if timed else dom["name"])
        if traffic is None:
            traffic, tsrc = pmc_traffic(dom["name"])
        if traffic is not None and roof["unit"] == "GB/s":
            roof["traffic_over_algorithmic"] = traffic / (dom["alg_bytes"] / dom["launches"])
        roof.update({"traffic": traffic, "traffic_source": tsrc, "kernel": LOOP_KERNEL if timed else dom["name"],
                     "device_clock_us": ((clk_us / clk_n) if clk_n else None) if timed else None, "device_clock_launches": clk_n if timed else None,
                     "launches_per_step": 1 if timed else dom["launches"], "avg_launch_us": avg_s * 1e6,
                     "algorithmic_per_launch": (dom["alg_flops"] if roof["unit"] == "TFLOP/s" else dom["alg_bytes"]) / dom["launches"],
                     "share_of_step_kernel_time": None if timed else dom["total_ms"] / tot,
                     "source": ROOF_SOURCE if timed else
                               "HIP events on the launch streams in one extra profiled step after the timed region"})
        out["roofline"] = roof
        # The north star names one kernel: the SSRN transposed convolution (modules.py:209-258; D_4 at T=200, D_7 at T=400,
        # networks.py:483-486).  Both of its rooflines, both arithmetic flavours, measured on device-resident data through
        # the same launches the SSRN path uses (oph_bench_conv1d_transpose).
        import ctypes as C
        kr = {}
        for lname, T in (("conv1d_transpose_D4", hp.max_T), ("conv1d_transpose_D7", 2 * hp.max_T)):
            # fp16x3: the SSRN path's default -- the input arrives as fp16 hi / lo planes written by the previous layer's launch, both phases
            # are one plane_gemm problem, and (round 6) the layer's LayerNorm runs inside that launch, which writes fp32 rows AND planes;
            # fp16x3_two_launches: the round-5 form (plane_gemm writing raw rows + ln_rows); fp16x3_rows: the round-3 launches (fp32
            # rows split inside the paired contraction)
            for prec, pname in ((0, "fp32"), (2, "fp16x3"), (10, "fp16x3_two_launches"), (5, "fp16x3_rows"), (1, "bf16x3")):
                us, by, fl = C.c_double(), C.c_double(), C.c_double()
                rc = eng.lib.oph_bench_conv1d_transpose(local_rank, B, T, hp.c, hp.c, prec, 3, 20, C.byref(us), C.byref(by), C.byref(fl))
                if rc != 0:
                    continue
                t_s = us.value * 1e-6
                peak = PEAK_F32_MFMA_TFLOPS if prec == 0 else PEAK_BF16_MFMA_TFLOPS
                mfma_flops = fl.value * (1 if prec == 0 else 3)          # the split flavours issue three 16-bit products per fp32 product
                kr["%s_%s" % (lname, pname)] = {
                    "shape": [B, T, hp.c], "avg_us": us.value, "algorithmic_bytes": by.value, "algorithmic_flops": fl.value,
                    "hbm_GB_per_s": by.value / t_s / 1e9, "hbm_frac": by.value / t_s / 1e9 / PEAK_HBM_GBS,
                    "mfma_TFLOP_per_s": mfma_flops / t_s / 1e12, "mfma_frac": mfma_flops / t_s / 1e12 / peak,
                    "mfma_peak_TFLOP_per_s": peak,
                    "launches": "ONE launch: plane_gemm (both phases, operands as fp16 planes) with the LayerNorm inside (statistics exchanged "
                                "between the column tiles; fp32 rows and planes out)" if prec == 2 else
                                "plane_gemm (both phases, operands as fp16 planes) + LayerNorm rows (fp32 rows and planes out)" if prec == 10
                                else "even- and odd-phase contractions in one launch + LayerNorm rows"}
        out["kernel_rooflines"] = kr
        out["kernel_classes"] = [
            {"kernel": p["name"], "launches": p["launches"], "total_ms": round(p["total_ms"], 4),
             "avg_us": round(p["total_ms"] * 1e3 / max(p["launches"], 1), 3),
             "alg_GB_per_s": round(p["alg_bytes"] / max(p["total_ms"], 1e-9) / 1e6, 2),
             "alg_TFLOP_per_s": round(p["alg_flops"] / max(p["total_ms"], 1e-9) / 1e9, 3)} for p in prof]
    if rank == 0 and world == 1 and not a.no_extra_legs:
        eng.synchronize()
        out["config"]["fallback_values"] = fallback_legs(hp, texts, B, local_rank)
        note("fallback legs done")
        restage(); one_step(False); eng.synchronize()
    if rank == 0 and not a.no_vocoder:
        one_step(False); eng.synchronize()
        try:
            out["vocoder"] = vocoder_leg(hp, eng, B, cpu=(world == 1 and not a.no_cpu_baseline))
        except Exception as e:                # never let the (untimed) vocoder leg take the bench line down
            out["vocoder"] = {"error": str(e)[:300]}
        if seq_ms and "fused_device_ms_per_batch" in out["vocoder"]:
            out["vocoder"]["text_to_waveform_ms_per_batch"] = seq_ms + out["vocoder"]["fused_device_ms_per_batch"]
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(hp, W, L, ends)
    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.barrier()               # rank 0 ran the extra profiled step / CPU baseline: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
