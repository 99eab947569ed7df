"""bench.py's multi-rank form on the one GPU of the test box: `python bench.py --gpus 2` spawns its two ranks, which share
GPU 0 (OPH_BENCH_SHARED_GPU=1: gloo instead of RCCL, which refuses two ranks per device).  What must hold on a real node
too: the line reports both ranks, their per-rank step times, whole-job frames, and which cross-stream flavour ran."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_on_one_gpu():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "OPH_STREAM_VALUE"):
        env.pop(k, None)
    env["OPH_BENCH_SHARED_GPU"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2",
           "--no-extra-legs", "--no-cpu-baseline", "--no-vocoder", "--no-profile"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    # Two PROCESSES whose decode launches each need all their workgroups resident on the same 64 CUs (a configuration only this
    # test creates: a real node gives every rank its own GPU) can each be handed half of them.  Round 5's answer was recovery: the
    # bounded wait ends, the tile is redone on a launch path that needs no co-residency (oph_get_counters[9], config.recoveries) --
    # and every such run paid one 2 s time-out per rank.  Round 6 adds prevention: the library's processes take turns on a physical
    # GPU (an advisory lock per PCI bus id around the calls that put work on the device, oph_api.hip), so the ranks alternate whole
    # batches and nothing collides (measured: recoveries [0, 0] in 7 of 7 runs, profiles/r06_turns_soak.txt).  The ladder is still
    # there for other tenants.  No retry here: a product failure fails the test.
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                     # rank 0 alone prints
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert len(out["rank_ms_per_step"]) == 2 and all(ms > 0 for ms in out["rank_ms_per_step"])
    assert out["ms_per_step"] >= max(out["rank_ms_per_step"]) * 0.999          # the line is the MAX over ranks
    cfg = out["config"]
    assert cfg["global_utterances"] == 32 and cfg["frames_per_step"] == 32 * 200
    assert abs(out["value"] - cfg["frames_per_step"] / (out["ms_per_step"] * 1e-3)) < 1e-6 * out["value"]
    assert cfg["cross_stream_sync"] == "stream value operations"           # multi-rank runs select OPH_STREAM_VALUE
    assert "all ranks on one GPU" in cfg["parallelism"]
    assert len(cfg["recoveries"]) == 2 and all(n >= 0 for n in cfg["recoveries"])      # per rank; > 0 only when the ranks collided
    # host CPU per rank over the 10 timed steps: REPORTED (DESIGN section 7 has the measured figures per mode); two ranks time-slicing
    # one GPU is not a production shape and a recovery puts a rank on the per-step launch path, so nothing is asserted about
    # its size (measured on one box: 0.1 - 0.5 of a core on the whole-decode path, 1.5 - 1.8 while a rank is on the per-step paths after a
    # collision -- the enqueue thread plus a runtime thread; a throughput-hygiene number must not void the parity record)
    print("rank_host_cores", cfg["rank_host_cores"], "recoveries", cfg["recoveries"], "busiest threads", [r.get("host_threads") for r in cfg["ranks"]])
    assert len(cfg["rank_host_cores"]) == 2 and all(c > 0.0 for c in cfg["rank_host_cores"]), cfg["rank_host_cores"]


def test_bench_line_survives_a_stuck_supplementary_leg():
    """bench.py's watchdog: the timed region's numbers are final when it ends, so a leg after it that does not come back must not
    cost the run its line.  OPH_BENCH_WATCHDOG_S = 0.2 makes the watchdog fire while the supplementary legs are still running:
    ONE JSON line with the metric, the timed region's value, its roofline (HIP events and the kernel's own clock), marked incomplete."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["OPH_BENCH_WATCHDOG_S"] = "0.2"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-vocoder"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert "incomplete" in out and out["metric"].startswith("mel-frames/sec") and out["unit"] == "frames/s"
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["value"] > 0
    assert abs(out["value"] - 16 * 200 / (out["ms_per_step"] * 1e-3)) < 1e-6 * out["value"]
    roof = out["roofline"]
    assert roof["bound"] == "hbm" and 0 < roof["frac"] < 1 and roof["avg_launch_us"] > 0
    assert abs(roof["device_clock_us"] - roof["avg_launch_us"]) < 0.02 * roof["avg_launch_us"]      # the two witnesses of the same launches
