"""Workload for per-STAGE hardware counters (run under rocprofv3 --pmc ...; tools/gpu_sq.sh).

rocprofv3 counters are per dispatch, so the stages of one mj_step are issued as separate dispatches
of the forward kernel (mjhip_batch_forward with a stage mask, LDS residency plan on), REPS times
each, on a batch that first ran SETTLE rollout steps (so contacts / constraint rows are those of the
benchmark regime).  tools/sq_summary.py maps dispatch order back to stage names.
Prints the stage order as JSON on the last line.
"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mujoco_amd as ma
from mujoco_amd import _capi as K
from bench import CONFIGS, initial_states, ctrl_noise

cfg = CONFIGS[os.environ.get("MODEL", "humanoid")]
REGIME = os.environ.get("REGIME", "uniform")       # uniform: the metric's regime; testspeed: OU-Halton noise after settling
SETTLE = int(os.environ.get("SETTLE", 120 if REGIME == "uniform" else 1020))
REPS = int(os.environ.get("REPS", 3))
lib = ma.lib()
model = ma.MjbModel(lib, os.path.join(ROOT, "tests", "golden", cfg["mjb"]))
if cfg["solver"]:
    model.set_option("solver", 0)
dm = ma.DeviceModel(lib, model)
nenv = int(os.environ.get("NENV", cfg["nenv"]))
b = ma.Batch(dm, nenv)
s0 = initial_states(b.get("qpos")[0], dm.nv, nenv, 1234, cfg["free_root"])
dev = torch.device("cuda", 0)
st0 = torch.from_numpy(s0).to(dev)
rng = np.random.Generator(np.random.PCG64(4321))
lo, hi = cfg["ctrl"]
if REGIME == "uniform":
    c = torch.from_numpy(rng.uniform(lo, hi, size=(nenv, SETTLE, dm.nu))).to(dev)
else:
    seq = ctrl_noise(SETTLE, dm.nu, cfg["dt"], lo*np.ones(dm.nu), hi*np.ones(dm.nu))
    c = torch.from_numpy(seq).to(dev)[None].expand(nenv, SETTLE, dm.nu).contiguous()
b.rollout_device(SETTLE, ma.mjSTATE_CTRL, st0.data_ptr(), 0, c.data_ptr(), 0, 0)
b.sync()
cnt = b.get("counts")
STAGES = [("kinematics", K.STAGE_KINEMATICS), ("collision", K.STAGE_COLLISION), ("transmission", K.STAGE_TRANSMISSION),
          ("velocity", K.STAGE_VELOCITY), ("inertia", K.STAGE_INERTIA), ("actuation", K.STAGE_ACTUATION),
          ("make", K.STAGE_MAKE), ("project", K.STAGE_PROJECT), ("reference", K.STAGE_REFERENCE),
          ("constraint", K.STAGE_CONSTRAINT | K.STAGE_FINISH), ("all", K.STAGE_ALL)]
for name, mask in STAGES:
    for _ in range(REPS):
        b.forward(mask, lds=True)
print(json.dumps({"stages": [n for n, _ in STAGES], "reps": REPS, "nenv": nenv, "regime": REGIME, "settle": SETTLE,
                  "variant": b.kernel_variant(), "mean_ncon": float(cnt[:, 0].mean()), "mean_nefc": float(cnt[:, 1].mean()),
                  "mean_niter": float(cnt[:, 5].mean())}))
