"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" = RCCL on
ROCm, "gloo" on CPU in tests).  The hot path shards by utterance -- there is NO data-path
collective (every op is per-utterance).  Collectives are used only for
  * the start-up weight broadcast (one flat fp32 tensor, root -> peers over xGMI), and
  * the one-integer MAX exchange that reproduces the reference's batch-coupled early stop
    (synthesize.py:225-228 breaks when ALL utterances of the global batch have ended).
"""
import numpy as np


def shard_range(n_items, rank, world):
    """Contiguous utterance shard of `rank`: sizes differ by at most one, order preserved."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _dist():
    """torch.distributed when a process group is up, else None.  Single-GPU synthesis does not need torch at all."""
    try:
        import torch.distributed as dist
    except ImportError:
        return None
    return dist if (dist.is_available() and dist.is_initialized()) else None


def broadcast_weights(W, inventory, src=0, device=None):
    """Broadcast the weight dict from `src` as ONE flat fp32 tensor (RCCL when device is a GPU)."""
    import torch
    from . import weights as WT
    dist = _dist()
    if dist is None:
        return W
    n = int(sum(int(np.prod(s)) for _, s in inventory))
    dev = torch.device("cuda", device) if (device is not None and torch.cuda.is_available()) else torch.device("cpu")
    if dist.get_rank() == src:
        flat = torch.from_numpy(WT.flatten(W, inventory)).to(dev)
    else:
        flat = torch.empty(n, dtype=torch.float32, device=dev)
    dist.broadcast(flat, src=src)
    return W if dist.get_rank() == src else WT.unflatten(flat.cpu().numpy(), inventory)


def load_weights_broadcast(eng, W, src=0, device=None):
    """Start-up weight distribution of a multi-GPU run: rank `src` holds the weight dict, every rank ends up with a
    loaded engine.  ONE flat fp32 tensor is broadcast (RCCL over xGMI when `device` is a GPU) and the receive buffer is handed to
    the library as it is (oph_set_weights_device): the variables are repacked by device kernels, nothing goes back through the host
    (round 3 copied the 210 MB to the host, unflattened them and re-uploaded them variable by variable on every rank).
    Without a process group this is eng.load_weights(W).
    Returns what the collective did, as the process group itself reports it (a run's evidence that RCCL carried the broadcast:
    bench.py prints it as config.collective): backend, world size, bytes, the broadcast's wall time on this rank (barrier on both
    sides, so the slowest peer's) and a 4-byte sum check of the received tensor against the root's."""
    import time
    import torch
    from . import weights as WT
    dist = _dist()
    if dist is None:
        eng.load_weights(W)
        return {"backend": None, "world_size": 1, "bytes": 0, "weight_broadcast_ms": 0.0, "GB_per_s": None}
    inventory = eng.inventory()
    n = int(sum(int(np.prod(s)) for _, s in inventory))
    on_gpu = torch.cuda.is_available()
    dev = torch.device("cuda", device) if (device is not None and on_gpu) else torch.device("cpu")
    if dist.get_rank() == src:
        flat = torch.from_numpy(WT.flatten(W, inventory)).to(dev)
    else:
        flat = torch.empty(n, dtype=torch.float32, device=dev)
    def _sync():
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
    dist.barrier(); _sync()
    t0 = time.perf_counter()
    dist.broadcast(flat, src=src)
    _sync(); dist.barrier()
    ms = (time.perf_counter() - t0) * 1e3
    # every rank holds the root's bytes: compare a checksum (float64 sum of the fp32 tensor) across ranks, loudly
    chk = flat.double().sum().reshape(1)
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if float(lo.item()) != float(hi.item()):
        raise RuntimeError("weight broadcast: ranks hold different tensors (checksum min %r max %r)" % (float(lo.item()), float(hi.item())))
    if on_gpu:
        if flat.device.type != "cuda":                 # gloo process group (tests: ranks sharing one GPU): one upload of the flat tensor
            flat = flat.to(torch.device("cuda", eng.device))
        torch.cuda.synchronize(flat.device)
        eng.load_weights_device(flat.data_ptr(), n)    # `flat` stays alive until the repack is done
    else:
        eng.load_weights(WT.unflatten(flat.numpy(), inventory))
    nbytes = 4 * n
    return {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "tensor_device": str(dev), "bytes": nbytes,
            "weight_broadcast_ms": ms, "GB_per_s": nbytes / (ms * 1e-3) / 1e9 if ms > 0 else None, "checksum_equal_on_all_ranks": True}


def global_max_int(value, device=None):
    """MAX over ranks of one integer (the global stop step)."""
    import torch
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return int(value)
    dev = torch.device("cuda", device) if (device is not None and torch.cuda.is_available()) else torch.device("cpu")
    t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def gather_arrays(arr):
    """Concatenate per-rank numpy arrays on every rank, in rank order (host side)."""
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return arr
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, arr)
    return np.concatenate(out, axis=0)


def sharded_text2mel(decode_local, resume_local, steps_local, max_T, device=None):
    """Batch-coupled early stop across shards.  Each rank first decodes its shard
    until ITS utterances have all ended (`decode_local()` -> steps run); ranks then agree on the
    global number of steps (MAX) and the ones that stopped earlier resume to it
    (`resume_local(t_begin, t_end)`), so every shard's Y equals the single-batch reference run."""
    steps = decode_local()
    g = global_max_int(steps, device)
    if g > steps:
        resume_local(steps, g)
    return g
