"""bench.py must start its own ranks when asked for --gpus N without a launcher, and must never report a different
rank count than it was asked for (`python bench.py --gpus 8` silently ran one rank).  CPU: gloo."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=300):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                          env=env, timeout=timeout)


def test_bench_spawns_the_requested_ranks():
    res = _run(["--gpus", "2", "--spawn-selftest"])
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 alone prints
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["max_rank_plus_1"] == 2.0


def test_bench_plumbing_at_world_8():
    """The shape the driver's 8-GPU run has: eight ranks from one command, one line from rank 0, every rank on its
    own device index and -- where the host has the cores -- on its own slice of them; a node with fewer than two cores per rank
    is reported (this container has 8)."""
    res = _run(["--gpus", "8", "--spawn-selftest"], timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["max_rank_plus_1"] == 8.0
    ranks = sorted(out["ranks"], key=lambda r: r["rank"])
    assert [r["rank"] for r in ranks] == list(range(8)) and [r["device"] for r in ranks] == [r["local_rank"] for r in ranks] == list(range(8))
    assert len(set(r["pid"] for r in ranks)) == 8
    nproc = out["host_nproc"]
    if nproc >= 8:
        slices = [tuple(r["cores"]) for r in ranks]
        assert all(len(s) == nproc // 8 for s in slices)
        assert len(set(c for s in slices for c in s)) == 8 * (nproc // 8)          # disjoint
    if nproc < 16:
        assert out["host_warning"] and "16" in out["host_warning"]
        assert "host cores for 8 ranks" in res.stderr


def test_bench_refuses_a_mismatching_launcher():
    res = _run(["--gpus", "4", "--spawn-selftest"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2",
                                                     "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert res.returncode != 0 and "WORLD_SIZE=2" in res.stderr


def test_bench_refuses_more_gpus_than_present():
    import torch
    if torch.cuda.device_count() >= 64:
        return
    res = _run(["--gpus", "64"])
    assert res.returncode != 0 and "exposes" in res.stderr
