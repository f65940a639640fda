#!/usr/bin/env python3
"""Generate the committed fixtures under tests/golden/ from the reference (runs only where
/root/reference and the compiled oracle exist; the fixtures travel to the GPU box).

  <model>.mjb                  compiled model, written by the reference's own mj_saveModel
  <model>_traj.npz             seeded initial states + random controls + the reference mj_step
                               trajectory (PGS, Euler, fp64) and per-step integer observables
                               (ncon, nefc, solver_niter, contact geom pairs)

Usage: python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refbind as rb  # noqa: E402

REF = os.environ.get("MUJOCO_REF", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")

MODELS = {
    "humanoid": "model/humanoid/humanoid.xml",
    "slider_crank": "model/slider_crank/slider_crank.xml",
}


def initial_states(m, d, nenv, rng):
    """reset state + hinge perturbation N(0,0.05^2), qvel ~ N(0,0.1^2); every other env
    starts from a keyframe (if the model has any) so contact-rich states are covered."""
    nstate = rb.mj_stateSize(m, rb.mjSTATE_FULLPHYSICS)
    s0 = np.zeros((nenv, nstate))
    for e in range(nenv):
        if m.nkey and e % 2 == 1:
            rb.mj_resetDataKeyframe(m, d, (e // 2) % m.nkey)
        else:
            rb.mj_resetData(m, d)
        jt = m.jnt_type
        for j in range(m.njnt):
            if jt[j] in (2, 3):
                d.qpos[m.jnt_qposadr[j]] += rng.normal(0, 0.05)
        d.qvel[:] = rng.normal(0, 0.1, size=m.nv)
        s0[e] = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
    return s0


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, rel in MODELS.items():
        m = rb.MjModel.from_xml_path(os.path.join(REF, rel))
        m.save_binary(os.path.join(OUT, name + ".mjb"))
        m.opt.solver = 0  # PGS (BASELINE config 2)
        d = rb.MjData(m)
        rng = np.random.Generator(np.random.PCG64(1234))
        nenv, nstep = 8, 120
        s0 = initial_states(m, d, nenv, rng)
        lo, hi = m.actuator_ctrlrange[:, 0], m.actuator_ctrlrange[:, 1]
        crng = np.random.Generator(np.random.PCG64(4321))
        ctrl = crng.uniform(lo, hi, size=(nenv, nstep, m.nu))
        nstate = s0.shape[1]
        traj = np.zeros((nenv, nstep, nstate))
        ncon = np.zeros((nenv, nstep), np.int32)
        nefc = np.zeros((nenv, nstep), np.int32)
        niter = np.zeros((nenv, nstep), np.int32)
        geoms = -np.ones((nenv, nstep, 32, 2), np.int32)
        for e in range(nenv):
            rb.mj_resetData(m, d)
            rb.mj_setState(m, d, s0[e], rb.mjSTATE_FULLPHYSICS)
            for t in range(nstep):
                d.ctrl[:] = ctrl[e, t]
                rb.mj_step(m, d)
                traj[e, t] = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
                ncon[e, t], nefc[e, t], niter[e, t] = d.ncon, d.nefc, d.solver_niter[0]
                c = d.contact
                k = min(len(c), 32)
                if k:
                    geoms[e, t, :k] = c["geom"][:k]
        np.savez_compressed(os.path.join(OUT, name + "_traj.npz"), state0=s0, ctrl=ctrl, state=traj,
                            ncon=ncon, nefc=nefc, niter=niter, geoms=geoms)
        print(name, "nq", m.nq, "nv", m.nv, "ncon max", ncon.max(), "nefc max", nefc.max(),
              "niter max", niter.max())


if __name__ == "__main__":
    main()
