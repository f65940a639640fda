"""gloo worker of tests/test_distributed_cpu.py (CPU, world_size 2 or 4): each rank rolls out its (possibly ragged)
block of the golden humanoid batch on the host emulation of the kernels, rank 0 gathers and checks."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mujoco_amd import _capi as K
from mujoco_amd.sharding import ChunkGather, env_slice, gather_to_rank0


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ntot = int(os.environ.get("MJHIP_TEST_NTOT", "6"))
    lib = K.Lib(os.path.join(ROOT, "tests", "hostsim", "libmjhip_hostsim.so"))
    m = K.MjbModel(lib, os.path.join(ROOT, "tests", "golden", "humanoid.mjb"))
    m.set_option("solver", 0)
    dm = K.DeviceModel(lib, m)
    fx = np.load(os.path.join(ROOT, "tests", "golden", "humanoid_traj.npz"))
    T = 8
    sl = env_slice(ntot, rank, world)
    counts = [env_slice(ntot, r, world).stop - env_slice(ntot, r, world).start for r in range(world)]
    b = K.Batch(dm, sl.stop - sl.start)
    out = b.rollout_host(T, K.mjSTATE_CTRL, fx["state0"][:ntot][sl], None, fx["ctrl"][:ntot][sl][:, :T])
    full = gather_to_rank0(torch.from_numpy(out[:, -1].copy()), rank, world, dist, nenv_total=ntot)
    # the per-chunk observation gather of bench.py: three chunks of per-step states, asynchronously, streamed in
    # slices of ONE environment (slice_bytes below one environment's chunk) with a receive pool of depth 2 on rank 0,
    # so buffer sets are recycled within a chunk; the sink reassembles what it is handed
    nstate = out.shape[2]
    got = {}

    def sink(chunk, lo, bufs):
        for r, t in enumerate(bufs):
            got.setdefault(chunk, {})[(r, lo)] = t.numpy().copy()

    obs = ChunkGather(rank, world, dist, depth=2, slice_bytes=8, sink=sink, nenv_total=ntot)     # (pads ragged blocks itself)
    assert obs.pad_to == max(counts)
    chunks = ((0, 3), (3, 6), (6, T))
    for ci, (c0, c1) in enumerate(chunks):
        obs.submit(torch.from_numpy(np.ascontiguousarray(out[:, c0:c1])))
        if ci == 1:
            obs.wait()                 # join mid-way
            # every rank: nothing may stay referenced after a join (the advisor's unbounded-growth finding)
            assert not obs._pending
        assert len(obs._pending) <= obs.depth
    obs.wait()
    ok = torch.tensor([1])
    if rank == 0:
        ref = fx["state"][:ntot, T - 1]
        ok[0] = int(full.shape == ref.shape and np.array_equal(full.numpy(), ref))
        for ci, (c0, c1) in enumerate(chunks):
            rows = []
            for r in range(world):
                for lo in range(counts[r]):
                    rows.append(got[ci][(r, lo)])
            arr = np.concatenate(rows, axis=0)
            ok[0] &= int(np.array_equal(arr, fx["state"][:ntot, c0:c1]))
        ok[0] &= int(obs.chunks == 3 and obs.bytes_sent == out.nbytes and obs.slices == 3*max(counts))
        # the pool never holds more than depth x world slices
        ok[0] &= int(obs.pool_bytes <= 2*world*3*nstate*8*2)
    dist.broadcast(ok, 0)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(ok[0]) == 1 else 3)


if __name__ == "__main__":
    main()
