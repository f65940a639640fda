"""GPU diagnostic: which per-env field is the first not to match the compiled reference bit for bit (cube_3x3x3)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import refbind as rb
import mujoco_amd
from mujoco_amd import _capi as K
from parity_utils import FORWARD_FIELDS
GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
lib = mujoco_amd.lib() if os.environ.get("HOSTSIM", "0") == "0" else K.Lib(os.path.join(os.path.dirname(GOLDEN), "hostsim", "libmjhip_hostsim.so"))
fx = np.load(os.path.join(GOLDEN, "cube_3x3x3_steps.npz"))
mm = K.MjbModel(lib, os.path.join(GOLDEN, "cube_3x3x3.mjb"))
dm = K.DeviceModel(lib, mm)
m = rb.MjModel.from_binary_path(os.path.join(GOLDEN, "cube_3x3x3.mjb"))
idx = [1, 40, 77, 120]
b = K.Batch(dm, len(idx))
nq, nv = m.nq, m.nv
b.set("qpos", fx["state"][idx, 1:1+nq]); b.set("qvel", fx["state"][idx, 1+nq:1+nq+nv]); b.set("qacc_warmstart", fx["warmstart"][idx]); b.set("ctrl", fx["ctrl"][idx])
b.forward()
d = rb.MjData(m)
fields = [f for f in FORWARD_FIELDS if f not in ("ten_length", "ten_J", "ten_velocity")] + ["efc_pos", "efc_margin", "efc_diagA", "efc_R", "efc_D", "efc_KBIP", "efc_vel", "efc_aref", "efc_b", "efc_force"]
for k, i in enumerate(idx):
    rb.mj_setState(m, d, fx["state"][i], rb.mjSTATE_FULLPHYSICS)
    d.qacc_warmstart[:] = fx["warmstart"][i]; d.ctrl[:] = fx["ctrl"][i]
    rb.mj_forward(m, d)
    bad = []
    for f in fields:
        try:
            got = b.get(f)[k]; r = np.asarray(getattr(d, f)).ravel()
        except Exception as ex:
            continue
        if not np.array_equal(got[:r.size], r): bad.append("%s %.2g" % (f, np.abs(got[:r.size] - r).max()))
    nc = d.ncon
    c = d.contact[:nc]
    for f, key in (("con_dist", "dist"), ("con_pos", "pos"), ("con_frame", "frame")):
        got = b.get(f)[k]; r = np.asarray(c[key]).ravel()
        if not np.array_equal(got[:r.size], r): bad.append("%s %.2g" % (f, np.abs(got[:r.size] - r).max()))
    spJ = b.get("sp_J")[k]; J = np.asarray(d.efc_J)[:d.nJ]
    if not np.array_equal(spJ[:d.nJ], J): bad.append("efc_J %.2g" % np.abs(spJ[:d.nJ] - J).max())
    print("step", i, "ncon", nc, "nefc", d.nefc, "niter", b.get("counts")[k][5], d.solver_niter[0], "NOT exact:", bad)
# one full step: which part of the state deviates
out = b.rollout_host(1, K.mjSTATE_CTRL, fx["state"][idx], fx["warmstart"][idx], fx["ctrl"][idx][:, None])[:, 0]
ref = fx["next"][idx]
for k, i in enumerate(idx):
    dq = np.abs(out[k, 1:1+nq] - ref[k, 1:1+nq]); dv = np.abs(out[k, 1+nq:1+nq+nv] - ref[k, 1+nq:1+nq+nv])
    print("step", i, "qpos maxdiff %.3g (%d entries differ)" % (dq.max(), (dq > 0).sum()), "qvel maxdiff %.3g (%d differ)" % (dv.max(), (dv > 0).sum()),
          "first qvel diffs at dofs", np.nonzero(dv)[0][:8], "qpos idx", np.nonzero(dq)[0][:8])
