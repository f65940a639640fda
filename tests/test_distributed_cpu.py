"""Multi-process sharding on CPU: the N>1 path of bench.py / mujoco_amd.sharding with world_size 2
over gloo.  The per-rank compute is the host emulation of the kernels."""
import os
import socket
import subprocess
import sys

from conftest import ROOT
from mujoco_amd.sharding import env_slice


def test_env_slices_partition_the_batch():
    for n in (1, 7, 4096, 4097):
        for w in (1, 2, 3, 8):
            sl = [env_slice(n, r, w) for r in range(w)]
            assert sl[0].start == 0 and sl[-1].stop == n
            assert all(a.stop == b.start for a, b in zip(sl, sl[1:]))
            sizes = [s.stop - s.start for s in sl]
            assert max(sizes) - min(sizes) <= 1


def _run_world(world, ntot):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "dist_worker.py")]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", MJHIP_TEST_NTOT=str(ntot))
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


def test_two_rank_gloo_rollout_matches_golden(hostsim_lib):
    _run_world(2, 6)


def test_four_rank_gloo_ragged_split_and_sliced_gather(hostsim_lib):
    # 7 environments over 4 ranks (2, 2, 2, 1): padded collectives, per-environment slices, pool of depth 2
    _run_world(4, 7)
