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


class ChunkGather:
    """The north star's "end-of-step observation gather": after every rollout launch (a chunk of steps)
    each rank's per-step state array [nlocal][c][nstate] travels to rank 0, asynchronously -- the
    collective is enqueued behind the producing kernel (NCCL/RCCL orders it after the work already on the
    current stream and runs it on its own stream), so it overlaps the NEXT chunk's compute; wait() joins
    everything outstanding.  Rank 0 keeps a small pool of receive buffers per shape (a gather may still be
    in flight when the next one is submitted).  world == 1: nothing to move, submit() only counts bytes.
    """

    def __init__(self, rank: int, world: int, dist=None, depth: int = 2):
        self.rank, self.world, self.dist = rank, world, dist
        self.depth = max(1, depth)
        self._pool = {}          # shape -> list of receive-buffer sets (rank 0)
        self._next = {}
        self._pending = []       # (work handle, buffers)
        self.bytes_sent = 0      # bytes this rank contributed
        self.chunks = 0
        self.last = None         # rank 0: the most recently completed gather (list of per-rank tensors)

    def submit(self, local):
        import torch
        self.chunks += 1
        self.bytes_sent += local.numel() * local.element_size()
        if self.world == 1 or self.dist is None:
            self.last = [local]
            return
        bufs = None
        if self.rank == 0:
            key = (tuple(local.shape), local.dtype, str(local.device))
            sets = self._pool.setdefault(key, [])
            if len(sets) < self.depth:
                sets.append([torch.empty_like(local) for _ in range(self.world)])
            k = self._next.get(key, 0)
            self._next[key] = (k + 1) % self.depth
            bufs = sets[k % len(sets)]
            # a buffer set is reused only after the gather that last wrote it has completed
            while len(self._pending) >= self.depth:
                self._finish_one()
        work = self.dist.gather(local, bufs, dst=0, async_op=True)
        self._pending.append((work, bufs))

    def _finish_one(self):
        work, bufs = self._pending.pop(0)
        work.wait()
        if bufs is not None:
            self.last = bufs

    def wait(self):
        """join every outstanding gather; returns rank 0's last gathered per-rank tensors (else None)"""
        while self._pending:
            self._finish_one()
        return self.last if self.rank == 0 else None
