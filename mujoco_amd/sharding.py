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


def gather_to_rank0(local, rank: int, world: int, dist=None, nenv_total: int = 0):
    """gather per-rank torch tensors [nlocal, ...] on rank 0, concatenated in rank order; returns None on the
    other ranks.  `dist` = torch.distributed (None/uninitialised: world 1).  With `nenv_total` the ranks hold the
    (possibly ragged) blocks of env_slice(nenv_total, r, world): shorter blocks are padded for the collective and
    trimmed on rank 0."""
    import torch
    if world == 1 or dist is None:
        return local
    counts = None
    if nenv_total:
        counts = [(lambda sl: sl.stop - sl.start)(env_slice(nenv_total, r, world)) for r in range(world)]
        nmax = max(counts)
        if local.shape[0] < nmax:
            pad = torch.zeros((nmax - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            local = torch.cat([local, pad], dim=0)
    out = [torch.empty_like(local) for _ in range(world)] if rank == 0 else None
    dist.gather(local, out, dst=0)
    if rank != 0:
        return None
    if counts is not None:
        out = [o[:c] for o, c in zip(out, counts)]
    return torch.cat(out, dim=0)


class ChunkGather:
    """The north star's "end-of-step observation gather": after every rollout launch (a chunk of steps)
    each rank's per-step state array [nlocal][c][nstate] travels to rank 0, asynchronously -- the
    collective is enqueued behind the producing kernel (NCCL/RCCL orders it after the work already on the
    current stream and runs it on its own stream), so it overlaps the NEXT chunk's compute (on CUDA the gathers are ordered on a side
    stream: the compute stream never waits for one, whatever the number of slices); wait() joins everything outstanding.

    A chunk is streamed in SLICES of whole environments of at most `slice_bytes` per rank (default 64 MiB): rank 0
    keeps `depth` receive-buffer sets of one slice each -- depth x world x slice_bytes resident (1 GiB at 8 ranks)
    instead of depth x world x the chunk (7.3 GB for 4096 x 250 x 56 doubles per rank).  A completed slice is handed
    to `sink(chunk_index, env_lo, per_rank_tensors)` on rank 0 (the consumer copies out what it keeps: the buffers are
    recycled) and remembered in `last`.  At most `depth` gathers are in flight on EVERY rank; a rank's `local` tensor
    is referenced until its last slice has completed and must not be overwritten before that (wait(), or `depth`
    further slices).  world == 1: nothing to move, submit() only counts bytes.
    """

    def __init__(self, rank: int, world: int, dist=None, depth: int = 2, slice_bytes: int = 64 << 20, sink=None,
                 pad_to: int = 0, nenv_total: int = 0):
        self.rank, self.world, self.dist = rank, world, dist
        self.pad_to = int(pad_to)    # ragged splits: every rank pads its array to this many environments (the
                                     # collective needs equal shapes; rank 0's consumer trims by env_slice)
        if nenv_total and world > 1:
            # the split of env_slice(nenv_total, r, world): pad to its longest block, so that every rank -- one that owns
            # no environment included -- issues the same number of equally shaped gathers
            self.pad_to = max(self.pad_to, max((lambda sl: sl.stop - sl.start)(env_slice(nenv_total, r, world)) for r in range(world)))
        self._side = None            # CUDA: the stream the gathers are ordered on (the compute stream never waits for them)
        self.depth = max(1, depth)
        self.slice_bytes = max(1, int(slice_bytes))
        self.sink = sink
        self._pool = {}          # shape -> list of receive-buffer sets (rank 0)
        self._next = {}
        self._pending = []       # (work handle, buffers, send tensor, chunk index, env_lo)
        self.bytes_sent = 0      # bytes this rank contributed
        self.chunks = 0
        self.slices = 0
        self.pool_bytes = 0      # rank 0: bytes held by the receive pool
        self.last = None         # rank 0: the most recently completed slice (list of per-rank tensors)

    def _recv_set(self, piece):
        import torch
        key = (tuple(piece.shape), piece.dtype, str(piece.device))
        sets = self._pool.setdefault(key, [])
        if len(sets) < self.depth:
            sets.append([torch.empty_like(piece) for _ in range(self.world)])
            self.pool_bytes += self.world * piece.numel() * piece.element_size()
        k = self._next.get(key, 0)
        self._next[key] = (k + 1) % self.depth
        return sets[k % len(sets)]

    def submit(self, local):
        self.chunks += 1
        self.bytes_sent += local.numel() * local.element_size()
        if self.world == 1 or self.dist is None:
            self.last = [local]
            return
        if self.pad_to > local.shape[0]:
            import torch
            pad = torch.zeros((self.pad_to - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            local = torch.cat([local, pad], dim=0)
        nloc = local.shape[0]
        if nloc == 0:
            raise ValueError("ChunkGather.submit: a rank without environments has to pad (pass nenv_total or pad_to), "
                             "or the ranks issue different numbers of collectives")
        per_env = max(1, (local.numel() // max(1, nloc)) * local.element_size())
        step = max(1, min(nloc, self.slice_bytes // per_env))
        # On CUDA the whole exchange is ordered on a SIDE stream: it first waits for what the compute stream has queued
        # (the kernel that produces `local`), the collectives are enqueued behind it, and recycling a receive-buffer set
        # -- `work.wait()` makes the CURRENT stream wait for that gather -- stalls the side stream only.  The next rollout
        # launch on the compute stream is never queued behind a gather, however many slices a chunk has.
        ctx = None
        if local.is_cuda:
            import torch
            if self._side is None:
                self._side = torch.cuda.Stream(device=local.device)
            self._side.wait_stream(torch.cuda.current_stream(local.device))
            local.record_stream(self._side)
            ctx = torch.cuda.stream(self._side)
            ctx.__enter__()
        try:
            for lo in range(0, nloc, step):
                piece = local[lo:lo + step]          # whole environments: a contiguous view
                # (all ranks) a buffer set / send view is reused only after the gather that used it has completed
                while len(self._pending) >= self.depth:
                    self._finish_one()
                bufs = self._recv_set(piece) if self.rank == 0 else None
                work = self.dist.gather(piece, bufs, dst=0, async_op=True)
                self._pending.append((work, bufs, piece, self.chunks - 1, lo))
                self.slices += 1
        finally:
            if ctx is not None:
                ctx.__exit__(None, None, None)

    def _finish_one(self):
        work, bufs, piece, chunk, lo = self._pending.pop(0)
        ctx = None
        if piece.is_cuda and self._side is not None:
            import torch
            ctx = torch.cuda.stream(self._side)
            ctx.__enter__()
        try:
            work.wait()
            if bufs is not None:
                self.last = bufs
                if self.sink is not None:
                    self.sink(chunk, lo, bufs)       # (runs on the side stream: the buffers are recycled in its order)
        finally:
            if ctx is not None:
                ctx.__exit__(None, None, None)

    def wait(self):
        """join every outstanding gather; returns rank 0's last gathered per-rank tensors (else None)"""
        while self._pending:
            self._finish_one()
        if self._side is not None:
            import torch
            torch.cuda.current_stream(self._side.device).wait_stream(self._side)    # the consumer reads `last` on the compute stream
        return self.last if self.rank == 0 else None
