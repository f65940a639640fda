"""world_size-2 worker of tests/test_distributed_cpu.py (gloo, CPU): each rank rolls out its block of
the golden humanoid batch on the host emulation of the kernels, rank 0 gathers and checks."""
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
    lib = K.Lib(os.path.join(ROOT, "tests", "hostsim", "libmjhip_hostsim.so"))
    m = K.MjbModel(lib, os.path.join(ROOT, "tests", "golden", "humanoid.mjb"))
    m.set_option("solver", 0)
    dm = K.DeviceModel(lib, m)
    fx = np.load(os.path.join(ROOT, "tests", "golden", "humanoid_traj.npz"))
    ntot, T = 6, 8
    sl = env_slice(ntot, rank, world)
    b = K.Batch(dm, sl.stop - sl.start)
    out = b.rollout_host(T, K.mjSTATE_CTRL, fx["state0"][:ntot][sl], None, fx["ctrl"][:ntot][sl][:, :T])
    full = gather_to_rank0(torch.from_numpy(out[:, -1].copy()), rank, world, dist)
    # the per-chunk observation gather of bench.py: three chunks of per-step states, asynchronously, with a
    # receive-buffer pool of depth 2 on rank 0 (so the third submit has to recycle the first set)
    obs = ChunkGather(rank, world, dist, depth=2)
    got = []
    for c0, c1 in ((0, 3), (3, 6), (6, T)):
        obs.submit(torch.from_numpy(np.ascontiguousarray(out[:, c0:c1])))
        if c1 == 6:
            last = obs.wait()          # join mid-way: chunks (0,3) and (3,6) are complete now
            if rank == 0:
                got.append(torch.cat(last, dim=0).numpy().copy())
    last = obs.wait()
    ok = torch.tensor([1])
    if rank == 0:
        ref = fx["state"][:ntot, T - 1]
        ok[0] = int(full.shape == ref.shape and np.array_equal(full.numpy(), ref))
        got.append(torch.cat(last, dim=0).numpy())
        ok[0] &= int(np.array_equal(got[0], fx["state"][:ntot, 3:6]) and np.array_equal(got[1], fx["state"][:ntot, 6:T]))
        ok[0] &= int(obs.chunks == 3 and obs.bytes_sent == out.nbytes)
    dist.broadcast(ok, 0)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(ok[0]) == 1 else 3)


if __name__ == "__main__":
    main()
