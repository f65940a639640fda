"""Workload for per-STAGE hardware counters (run under rocprofv3 --pmc ...; tools/gpu_sq.sh).

rocprofv3 counters are per dispatch, so one mj_step is issued as dispatches of the forward kernel
(mjhip_batch_forward with a stage mask, LDS residency plan on) over PREFIXES of the stage list --
kinematics; kinematics + collision; ... ; everything -- REPS times each, on a batch that first ran
SETTLE rollout steps (so contacts / constraint rows are those of the benchmark regime).  A stage's
counters are the difference of two consecutive prefixes (tools/sq_summary.py): every dispatch runs
all the stages its last one depends on, in the same launch, so LDS-resident intermediates exist and
the stage solves the benchmark's problem.  (Round 4 dispatched single-stage masks: a stage whose
inputs live only in the LDS plan of the previous stages -- the constraint rows -- then worked on
whatever the fresh LDS block held; its "constraint" row exceeded the row of the whole step.)
A last group of dispatches runs STAGE_ALL as the cross-check of the final prefix.
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
STAGES = [(n, m) for n, m in STAGES if n != "all"]
prefix = 0
for name, mask in STAGES:
    prefix |= mask
    for _ in range(REPS):
        b.forward(prefix, lds=True)
for _ in range(REPS):
    b.forward(K.STAGE_ALL, lds=True)
print(json.dumps({"stages": [n for n, _ in STAGES] + ["all"], "prefix": True, "reps": REPS, "nenv": nenv, "regime": REGIME, "settle": SETTLE,
                  "variant": b.kernel_variant(), "mean_ncon": float(cnt[:, 0].mean()), "mean_nefc": float(cnt[:, 1].mean()),
                  "mean_niter": float(cnt[:, 5].mean())}))
