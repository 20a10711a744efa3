"""Multi-GPU plumbing: one process per GPU, bursts sharded with no data-path collective.

The only collective is the init-time broadcast of the constant-table blob: rank 0 builds
it, every other rank receives it over RCCL/xGMI (`nccl` backend) -- or over gloo in the CPU tests --
validates its checksum and creates its context from it.  torch.distributed is plumbing here.
"""
import os

import numpy as np


def shard_range(total, rank, world):
    """Contiguous, balanced [start, end) slice of `total` independent units for `rank`."""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def init_from_env(backend):
    """torch.distributed init from RANK/WORLD_SIZE/MASTER_* (torchrun contract)."""
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend)
    return dist.get_rank(), dist.get_world_size()


def broadcast_tables(pkg, sps, device=None, src=0):
    """Rank `src` builds the table blob on the host; everybody ends up with the same validated
    bytes.  device=None -> CPU tensor (gloo); otherwise a cuda device (nccl == RCCL).
    Returns (host uint8 numpy blob, device/CPU torch tensor holding it)."""
    import torch
    import torch.distributed as dist
    n = pkg.lib().trxsig_tables_bytes(sps)
    rank = dist.get_rank() if dist.is_initialized() else 0
    if rank == src:
        t = torch.from_numpy(pkg.build_tables_host(sps).copy())
    else:
        t = torch.zeros(n, dtype=torch.uint8)
    if device is not None:
        t = t.to(device)
    if dist.is_initialized():                               # (a one-rank group still runs the collective: the GPU test relies on it)
        dist.broadcast(t, src=src)
    blob = t.cpu().numpy()
    if not pkg.tables_valid(blob):
        raise pkg.TrxSigError("rank %d: broadcast table blob failed validation" % rank)
    return blob, t


def max_over_ranks(value, device=None):
    """MAX all-reduce of one python float (the bench's step time)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def ranks_seen(rank, device=None):
    """All-gather of the rank ids that took part (bench.py reports it next to n_gpus)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return [int(rank)]
    t = torch.tensor([rank], dtype=torch.int64, device=device if device is not None else "cpu")
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return sorted(int(o.item()) for o in out)


def gather_floats(value, device=None):
    """All-gather of one python float per rank, in rank order (bench.py: every rank's own step time)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]
