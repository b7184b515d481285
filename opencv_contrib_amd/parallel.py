"""Batched-frames mode across the GPUs of one node: the path shards ONLY across independent image
pairs / frames -- one process per GPU, a static block partition of the batch, no data-path collective.
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests) is used for rendezvous,
barriers, the max-over-ranks timing reduction and (optionally) gathering per-rank results to rank 0.
"""
from __future__ import annotations

import os


def shard_range(n_items: int, world: int, rank: int) -> range:
    """Static block partition (pair i -> rank i // ceil(n/world)): contiguous, sizes differ by at most ... the last
    ranks may be short or empty when n is not a multiple of world."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    per = -(-n_items // world)
    lo = min(rank * per, n_items)
    hi = min(lo + per, n_items)
    return range(lo, hi)


def init_distributed(backend: str | None = None, device=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).  Returns (dist or None, rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1:
        return None, 0, 1, local
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl":
        torch.cuda.set_device(local)
        kw["device_id"] = torch.device("cuda", local)
    if not dist.is_initialized():
        dist.init_process_group(backend, **kw)
    return dist, rank, world, local


def max_over_ranks(dist, value: float, device=None) -> float:
    """Slowest rank's time (the job's wall time)."""
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_over_ranks(dist, value: float, device=None) -> list:
    """Every rank's value, in rank order (per-rank rates of a weak-scaling job)."""
    if dist is None:
        return [float(value)]
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    outs = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return [float(o.item()) for o in outs]


def sum_over_ranks(dist, value: float, device=None) -> float:
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def run_sharded(dist, rank: int, world: int, n_items: int, work, warmup: int, steps: int, sync=None, device=None):
    """Runs `work(range_of_items)` `warmup` + `steps` times on this rank's shard, bracketed by barriers, and returns
    (wall seconds = max over ranks, items processed by the whole job per step).  `sync` = device synchronisation."""
    import time
    shard = shard_range(n_items, world, rank)
    for _ in range(warmup):
        work(shard)
    if sync:
        sync()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        work(shard)
    if sync:
        sync()
    el = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    wall = max_over_ranks(dist, el, device)
    total = sum_over_ranks(dist, float(len(shard)), device)
    return wall, int(round(total))


def gather_to_rank0(dist, rank: int, world: int, tensor):
    """Gather per-rank result tensors (same shape except dim 0) on rank 0 -- the only data-path exchange of the
    batched mode (flows back to the caller's GPU).  Returns the concatenation on rank 0, None elsewhere."""
    if dist is None:
        return tensor
    import torch
    sizes = [None] * world
    dist.all_gather_object(sizes, int(tensor.shape[0]))
    if rank == 0:
        outs = [tensor]
        for r in range(1, world):
            buf = torch.empty((sizes[r],) + tuple(tensor.shape[1:]), dtype=tensor.dtype, device=tensor.device)
            if sizes[r]:
                dist.recv(buf, src=r)
            outs.append(buf)
        return torch.cat(outs, 0)
    if tensor.shape[0]:
        dist.send(tensor.contiguous(), dst=0)
    return None


# ---------------------------------------------------------------------------------------------------------------------
# Batched-frames mode with the inputs on ONE GPU (BASELINE north_star: "RCCL over xGMI for the scatter/gather only";
# scatter of the image pairs from GPU 0, gather of the flows back).  torch.distributed's scatter / gather on
# the "nccl" backend are grouped point-to-point sends / receives (ncclGroupStart ... ncclGroupEnd), which is what a
# point-to-point fabric wants: the root's 7 xGMI links run concurrently.  Both are issued asynchronously on the
# collective stream so that the exchange of step k+1 overlaps the compute of step k.

def scatter_from_rank0(dist, rank: int, world: int, out, parts=None, async_op: bool = False):
    """Every rank receives its part into `out`; `parts` (rank 0 only): `world` tensors shaped like `out`.
    Returns the Work handle when async_op (None in a single-process run)."""
    if dist is None:
        out.copy_(parts[0])
        return None
    return dist.scatter(out, scatter_list=list(parts) if rank == 0 else None, src=0, async_op=async_op)


def gather_on_rank0(dist, rank: int, world: int, tensor, outs=None, async_op: bool = False):
    """Rank 0 receives every rank's `tensor` into `outs[r]` (`world` tensors shaped like `tensor`)."""
    if dist is None:
        outs[0].copy_(tensor)
        return None
    return dist.gather(tensor, gather_list=list(outs) if rank == 0 else None, dst=0, async_op=async_op)


def run_exchange_pipeline(dist, rank: int, world: int, parts, local_in, local_out, root_out, compute, steps: int, sync=None):
    """`steps` rounds of scatter -> compute -> gather, double buffered: the scatter of round k+1 and the gather of round k-1
    run on the collective stream while round k computes.

    parts:     rank 0: `world` input tensors (one per rank), else None
    local_in:  two input buffers of this rank, local_out: two output buffers
    root_out:  rank 0: two lists of `world` output tensors, else None
    compute(inp, out): this rank's work for one round, enqueued on the current stream
    Returns the wall seconds of this rank (bracket with barriers and reduce with max_over_ranks at the call site)."""
    import time

    def wait(w):
        if w is not None:
            w.wait()      # nccl: the current stream waits for the collective; gloo: the host does

    sc = [None, None]
    ga = [None, None]
    if sync:
        sync()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    sc[0] = scatter_from_rank0(dist, rank, world, local_in[0], parts, async_op=True)
    for k in range(steps):
        b = k & 1
        wait(sc[b])
        if k + 1 < steps:                       # next round's inputs travel while this round computes
            sc[b ^ 1] = scatter_from_rank0(dist, rank, world, local_in[b ^ 1], parts, async_op=True)
        wait(ga[b])                             # round k-2's gather has drained local_out[b]
        compute(local_in[b], local_out[b])
        ga[b] = gather_on_rank0(dist, rank, world, local_out[b], root_out[b] if rank == 0 else None, async_op=True)
    wait(ga[0])
    wait(ga[1])
    if sync:
        sync()
    el = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    return el
