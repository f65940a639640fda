#!/usr/bin/env python3
"""Rate of the drop-in entry point with host arrays (mujoco_amd.rollout.rollout -> mjhip_rollout), next to the device-resident
rate of the same rollout; $MJHIP_ROLLOUT_CHUNK selects the chunk length of the overlapped copies (0: one launch, copies
before and after).  The caller's mjModel / mjData come from the compiled reference (the caller's MuJoCo).
usage (GPU box): [MJHIP_ROLLOUT_CHUNK=n] python tools/api_rate.py [nenv] [nstep]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                               # noqa: E402  (first: its HIP runtime has to be the one that initialises)
torch.cuda.set_device(0)
import mujoco_amd as ma                    # noqa: E402
from mujoco_amd import rollout as ro       # noqa: E402
from oracle import refbind as rb           # noqa: E402  (the caller's MuJoCo objects)
import bench                               # noqa: E402

nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nstep = int(sys.argv[2]) if len(sys.argv) > 2 else 250
cfg = bench.CONFIGS["humanoid"]
path = os.path.join(ROOT, "tests", "golden", cfg["mjb"])
lib = ma.lib()
model = ma.MjbModel(lib, path); model.set_option("solver", 0)
dm = ma.DeviceModel(lib, model)
batch = ma.Batch(dm, nenv)
s0 = bench.initial_states(batch.get("qpos")[0], dm.nv, nenv, seed=1234)
res = bench.api_regime(path, 0, 0, nenv, nstep, s0, cfg, batch, torch.cuda.current_stream().cuda_stream, torch.device("cuda", 0))
print("chunk", os.environ.get("MJHIP_ROLLOUT_CHUNK", "default"), {k: res[k] for k in ("value", "device_resident_value", "ratio_to_device_resident", "all_seconds", "identical_to_device_resident")})
