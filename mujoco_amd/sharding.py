"""Environment sharding across the GPUs of one node.

Environments are independent, so the partition is a plain contiguous block split: rank g owns
envs [g*n/G, (g+1)*n/G) of a global batch, steps them with no per-step exchange, and the only
collective is the gather of per-env results to rank 0 (RCCL over xGMI on GPUs; gloo in CPU tests).
"""
from __future__ import annotations


def env_slice(nenv_total: int, rank: int, world: int) -> slice:
    """contiguous block of the global env range owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(nenv_total, world)
    lo = rank * base + min(rank, rem)
    return slice(lo, lo + base + (1 if rank < rem else 0))


def gather_to_rank0(local, rank: int, world: int, dist=None):
    """gather equally-shaped per-rank torch tensors [nlocal, ...] on rank 0, concatenated in rank
    order; returns None on the other ranks.  `dist` = torch.distributed (None/uninitialised: world 1)."""
    import torch
    if world == 1 or dist is None:
        return local
    out = [torch.empty_like(local) for _ in range(world)] if rank == 0 else None
    dist.gather(local, out, dst=0)
    return torch.cat(out, dim=0) if rank == 0 else None
