"""Per-stage instruction counts: launches mjhip_batch_forward once per stage bit (all-global mode) on
a settled 4096-env humanoid batch; run under `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU
SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES` and read the per-dispatch rows in launch
order (tools/gpu_stage_insts.sh prints the table)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mujoco_amd as ma
from mujoco_amd import _capi as K
from bench import initial_states

lib = ma.lib()
model = ma.MjbModel(lib, os.path.join(ROOT, "tests", "golden", "humanoid.mjb"))
model.set_option("solver", 0)
dm = ma.DeviceModel(lib, model)
nenv = 4096
b = ma.Batch(dm, nenv)
s0 = initial_states(b.get("qpos")[0], dm.nv, nenv, 1234)
rng = np.random.Generator(np.random.PCG64(4321))
dev = torch.device("cuda", 0)
st0 = torch.from_numpy(s0).to(dev)
W = 150
cw = torch.from_numpy(rng.uniform(-1, 1, size=(nenv, W, dm.nu))).to(dev)
b.rollout_device(W, ma.mjSTATE_CTRL, st0.data_ptr(), 0, cw.data_ptr(), 0, 0)
b.sync()
STAGES = [("kinematics", K.STAGE_KINEMATICS), ("inertia", K.STAGE_INERTIA), ("transmission", K.STAGE_TRANSMISSION),
          ("velocity", K.STAGE_VELOCITY), ("actuation+accel", K.STAGE_ACTUATION), ("collision", K.STAGE_COLLISION),
          ("make", K.STAGE_MAKE), ("project", K.STAGE_PROJECT), ("reference", K.STAGE_REFERENCE),
          ("constraint", K.STAGE_CONSTRAINT), ("finish", K.STAGE_FINISH), ("euler", K.STAGE_EULER)]
b.forward()            # everything valid once
for name, bit in STAGES:
    b.forward(stages=bit, lds=True)
print("STAGE_ORDER " + ",".join(n for n, _ in STAGES))
c = b.get("counts")
print("mean nefc", c[:, 1].mean(), "mean niter", c[:, 5].mean())
