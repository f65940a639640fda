#!/usr/bin/env python3
"""PCIe-inclusive rollout rate: host numpy arrays in (state0, control), host array out (state of every
step) through the C ABI's host-buffer entry point (mjhip_batch_rollout_sensors without
MJHIP_ROLLOUT_ON_DEVICE: pooled pinned staging, H2D, one rollout kernel, D2H).  Reported next to the
device-resident rate of bench.py; it is never bench.py's `value`.
usage (GPU box): python tools/pcie_rate.py [nenv] [nstep] [reps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mujoco_amd as ma          # noqa: E402

nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nstep = int(sys.argv[2]) if len(sys.argv) > 2 else 100
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
lib = ma.lib()
model = ma.MjbModel(lib, os.path.join(ROOT, "tests", "golden", "humanoid.mjb"))
model.set_option("solver", 0)
dm = ma.DeviceModel(lib, model)
b = ma.Batch(dm, nenv)
nstate, nu = dm.size("nstate"), dm.size("nu")
rng = np.random.default_rng(0)
b.reset()
s0 = np.concatenate([b.get("time")[:, :1], b.get("qpos"), b.get("qvel")], axis=1)
s0[:, 1 + 7:1 + dm.size("nq")] += rng.normal(0, .05, (nenv, dm.size("nq") - 7))
ctrl = rng.uniform(-1, 1, (nenv, nstep, nu))
out = b.rollout_host(nstep, ma.mjSTATE_CTRL, s0, None, ctrl)      # warm-up: staging buffers, code objects
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    out = b.rollout_host(nstep, ma.mjSTATE_CTRL, s0, None, ctrl)
    ts.append(time.perf_counter() - t0)
t = min(ts)
mb_out = out.nbytes / 1e6
mb_in = (s0.nbytes + ctrl.nbytes) / 1e6
print(f"humanoid, {nenv} envs x {nstep} steps, host arrays in/out: best of {reps}: {t*1e3:.1f} ms "
      f"-> {nenv*nstep/t/1e6:.3f} M env-steps/s PCIe-inclusive ({mb_in:.0f} MB in, {mb_out:.0f} MB out, "
      f"{(mb_in + mb_out)/t/1e3:.1f} GB/s of host traffic incl. the numpy output allocation); all reps ms: "
      + ", ".join(f"{x*1e3:.1f}" for x in ts))
