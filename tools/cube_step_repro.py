"""Diagnostic: reproduce one (env, step) of the cube bench workload -- oracle trajectory up to that step, then the
step itself on the chosen backend from identical inputs, field by field.  HOSTSIM=1 uses the host emulation."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import refbind as rb
import mujoco_amd
from mujoco_amd import _capi as K
from bench import initial_states
from parity_utils import FORWARD_FIELDS
GOLDEN = os.path.join(ROOT, "tests", "golden")
env, step = int(sys.argv[1]), int(sys.argv[2])
lib = K.Lib(os.path.join(ROOT, "tests", "hostsim", "libmjhip_hostsim.so")) if os.environ.get("HOSTSIM") == "1" else mujoco_amd.lib()
mm = K.MjbModel(lib, os.path.join(GOLDEN, "cube_3x3x3.mjb"))
dm = K.DeviceModel(lib, mm)
m = rb.MjModel.from_binary_path(os.path.join(GOLDEN, "cube_3x3x3.mjb"))
nenv, W, Kst = 2048, 20, 100
b1 = K.Batch(dm, 1)
qpos0 = b1.get("qpos")[0]
s0 = initial_states(qpos0, m.nv, nenv, seed=1234, free_root=False)
crng = np.random.Generator(np.random.PCG64(4321))
uw = crng.uniform(-0.05, 0.05, size=(nenv, W, m.nu)); uk = crng.uniform(-0.05, 0.05, size=(nenv, Kst, m.nu))
u = np.concatenate([uw, uk], axis=1)[env]
d = rb.MjData(m)
rb.mj_setState(m, d, s0[env], rb.mjSTATE_FULLPHYSICS)
for t in range(step):
    d.ctrl[:] = u[t]; rb.mj_step(m, d)
pre = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS); warm = np.array(d.qacc_warmstart)
d.ctrl[:] = u[step]
nq, nv = m.nq, m.nv
b1.set("qpos", pre[None, 1:1+nq]); b1.set("qvel", pre[None, 1+nq:1+nq+nv]); b1.set("qacc_warmstart", warm[None]); b1.set("ctrl", u[None, step])
b1.forward()
rb.mj_forward(m, d)
c = b1.get("counts")[0]
print("counts", c[:9], "ref ncon/nefc/niter/nisl/nJ", d.ncon, d.nefc, d.solver_niter[:3], d.nisland, d.nJ)
bad = []
for f in [x for x in FORWARD_FIELDS if x not in ("ten_length", "ten_J", "ten_velocity")] + ["efc_pos", "efc_margin", "efc_diagA", "efc_R", "efc_D", "efc_vel", "efc_aref", "efc_b", "efc_force"]:
    try:
        got = b1.get(f)[0]; r = np.asarray(getattr(d, f)).ravel()
    except Exception:
        continue
    if not np.array_equal(got[:r.size], r): bad.append("%s %.3g" % (f, np.abs(got[:r.size] - r).max()))
print("fields not exact:", bad)
print("island sizes ref", np.asarray(d.island_nv)[:d.nisland], np.asarray(d.island_nefc)[:d.nisland])
