"""Which workgroups share a SIMD?  (needs the -DMJH_PROFILE library; launch order = identity with
MJHIP_BALANCE=0, so workgroup w steps environment w and the hardware ids recorded per environment give
the map w -> (xcc, se, sh, cu, simd))"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mujoco_amd as ma
from bench import initial_states
lib = ma.lib()
model = ma.MjbModel(lib, os.path.join(ROOT, "tests", "golden", "humanoid.mjb"))
model.set_option("solver", 0)
dm = ma.DeviceModel(lib, model)
nenv = 4096
b = ma.Batch(dm, nenv)
s0 = initial_states(b.get("qpos")[0], dm.nv, nenv, 1234)
dev = torch.device("cuda", 0)
st0 = torch.from_numpy(s0).to(dev)
ck = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, size=(nenv, 10, dm.nu))).to(dev)
for rep in range(2):
    b.set("prof", np.zeros((nenv, 32)))
    b.rollout_device(10, ma.mjSTATE_CTRL, st0.data_ptr(), 0, ck.data_ptr(), 0, 0)
    b.sync()
    p = b.get("prof")
    hw = p[:, 27].astype(np.int64); xcc = p[:, 26].astype(np.int64) & 0xf
    cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1; simd = (hw >> 4) & 3
    key = ((xcc * 8 + se) * 2 + sh) * 64 + cu * 4 + simd
    print("rep", rep, "first 24 workgroups: (xcc,se,sh,cu,simd) =", [(int(xcc[w]), int(se[w]), int(sh[w]), int(cu[w]), int(simd[w])) for w in range(24)])
    groups = {}
    for w in range(nenv):
        groups.setdefault(int(key[w]), []).append(w)
    gl = sorted(groups.values(), key=lambda g: g[0])
    print("  SIMD groups (first 12):", gl[:12])
    d = np.array([np.diff(g) for g in gl if len(g) == 4])
    vals, cnts = np.unique(d, return_counts=True)
    print("  differences between the workgroup ids sharing a SIMD:", dict(zip(vals.tolist(), cnts.tolist())))
    # same CU
    keycu = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    g2 = {}
    for w in range(nenv): g2.setdefault(int(keycu[w]), []).append(w)
    print("  one CU's workgroups:", sorted(g2.values(), key=lambda g: g[0])[0])
