"""Multi-GPU plumbing covered on CPU with 2 gloo processes: weight broadcast, utterance
sharding, and the batch-coupled global stop."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ophelia_amd import parallel, weights as WT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        inv = [("Text2Mel/A/conv1d/kernel", (1, 8, 4)), ("Text2Mel/A/conv1d/bias", (4,)), ("SSRN/B/normalize/gamma", (6,))]
        W = WT.random_weights(inv, 5) if rank == 0 else None
        W = parallel.broadcast_weights(W, inv, src=0)
        ref = WT.random_weights(inv, 5)
        ok_w = all(np.array_equal(W[n], ref[n]) for n, _ in inv)
        # utterance sharding: 7 utterances over 2 ranks, order preserved
        lo, hi = parallel.shard_range(7, rank, world)
        got = parallel.gather_arrays(np.arange(7)[lo:hi])
        # batch-coupled stop: rank 0's shard ends after 5 steps, rank 1's after 9 -> both must run 9
        local_steps = [5, 9][rank]
        log = []
        g = parallel.sharded_text2mel(lambda: local_steps, lambda a, b: log.append((a, b)), None, 20)
        q.put((rank, ok_w, got.tolist(), g, log, parallel.global_max_int(rank * 3)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    for rank, ok_w, got, g, log, mx in res:
        assert ok_w and got == list(range(7)) and g == 9 and mx == 3
    assert res[0][4] == [(5, 9)] and res[1][4] == []      # only the early-stopping rank resumes, to the global step


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 16, 128):
        for w in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_flatten_roundtrip():
    inv = [("a/kernel", (2, 3)), ("b/bias", (4,))]
    W = {"a/kernel": np.arange(6, dtype=np.float32).reshape(2, 3), "b/bias": np.ones(4, np.float32)}
    W2 = WT.unflatten(WT.flatten(W, inv), inv)
    assert all(np.array_equal(W[k], W2[k]) for k in W)
