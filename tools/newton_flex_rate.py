#!/usr/bin/env python3
"""Rate of the Newton solver beyond 128 dofs (mjh_newtonx.h) on the device: a 9 x 9 shell flex (243 dofs) resting on a sphere,
a box, a capsule and a cylinder -- the scene of tests/test_flex_hostsim.py::test_newton_beyond_128_dofs -- nenv copies with
perturbed velocities, `pre` steps on the reference first (into the contact phase), then nstep timed steps on the device.
The reference's time for the same steps of ONE environment on one host core is printed beside it (test infrastructure:
oracle/_ref).  usage (GPU box): python tools/newton_flex_rate.py [nenv] [nstep] [cone]"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mujoco_amd as ma                    # noqa: E402
from mujoco_amd import _capi as K          # noqa: E402
from oracle import refbind as rb           # noqa: E402
import test_flex_hostsim as fh             # noqa: E402

nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nstep = int(sys.argv[2]) if len(sys.argv) > 2 else 100
cone = sys.argv[3] if len(sys.argv) > 3 else "pyramidal"
pre = 110
option = f'solver="{os.environ.get("SOLVER", "Newton")}" cone="{cone}" tolerance="1e-8" timestep=".001" integrator="Euler"'
with tempfile.TemporaryDirectory() as td:
    p = os.path.join(td, "newton.xml")
    open(p, "w").write(fh.shell_xml("9 9 1", fh.SHELL_GEOMS, option=option))
    m = rb.MjModel.from_xml_path(p)
lib = ma.lib()
dm = K.DeviceModel(lib, m)
b = K.Batch(dm, nenv)
d = rb.MjData(m)
for _ in range(pre): rb.mj_step(m, d)
s = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
rng = np.random.default_rng(0)
qvel = np.tile(s[None, 1 + m.nq:1 + m.nq + m.nv], (nenv, 1))
qvel[1:] += rng.normal(0, 1e-3, (nenv - 1, m.nv))
b.set("time", np.tile(s[None, :1], (nenv, 1))); b.set("qpos", np.tile(s[None, 1:1 + m.nq], (nenv, 1))); b.set("qvel", qvel)
b.set("qacc_warmstart", np.tile(d.qacc_warmstart[None, :], (nenv, 1)))
b.step()                                   # (first launch: module load, LDS plan)
rb.mj_step(m, d)
try:
    prof0 = b.get("prof")
    b.set("prof", np.zeros_like(prof0))
except Exception:
    prof0 = None
t0 = time.perf_counter()
for _ in range(nstep): b.step()
t1 = time.perf_counter()
c0 = time.perf_counter()
for _ in range(nstep): rb.mj_step(m, d)
c1 = time.perf_counter()
cnt = b.get("counts")
solver = os.environ.get("SOLVER")
ok = np.array_equal(b.get("qpos")[0], d.qpos) and np.array_equal(b.get("qvel")[0], d.qvel)
print(f"newton beyond 128 dofs: nv {m.nv} cone {cone} nenv {nenv} x {nstep} steps: device {nenv*nstep/(t1 - t0):.1f} env-steps/s "
      f"({(t1 - t0)/nstep*1e3:.2f} ms per step of the batch); reference, one core, one environment {nstep/(c1 - c0):.1f} steps/s; "
      f"environment 0 bit-identical to the reference after {nstep + 1} steps: {ok}; last step: mean ncon {cnt[:, 0].mean():.1f} nefc {cnt[:, 1].mean():.1f} "
      f"Newton iterations {cnt[:, 5].mean():.2f}; warnings {int(b.get('warning').sum())}")

if prof0 is not None and prof0.size:
    # (MJHIP_LIB=tools/variants/libmjhip_prof.so: per-stage accumulators of the step, microseconds per env-step)
    NAMES = ["begin", "kin", "collision", "compos", "tendon", "transmission", "tavel", "comvel", "passive", "rne", "crb", "factor",
             "actuation", "accel", "make", "project", "reference", "constraint", "finish", "euler", "end"]
    p = b.get("prof")
    nst = p[:, 30].mean() if p[:, 30].mean() > 0 else float(nstep)
    print("  stages (us per env-step):", ", ".join(f"{n} {p[:, i].mean()/nst:.0f}" for i, n in enumerate(NAMES) if p[:, i].mean()/nst >= 1), f"| total in kernel {p[:, 31].mean()/nst:.0f}")
    print("  primal solver: set-up %.0f  Hessian+factor %.0f  factor solves %.0f  incremental updates %.0f  line search %.0f  constraint update+grad %.0f us"
          % tuple(p[:, k].mean()/nst for k in (32, 33, 34, 35, 36, 37)))
    print("  outside the stages: state checks %.0f  compressed rows + islands %.0f us; row lengths %.0f  columns and values %.0f  transpose %.0f us"
          % tuple(p[:, k].mean()/nst for k in (48, 49, 57, 58, 59)))
