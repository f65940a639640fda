#!/usr/bin/env python3
"""Generate the committed fixtures under tests/golden/ from the reference (runs only where
/root/reference and the compiled oracle exist; the fixtures travel to the GPU box).

  <model>.mjb                  compiled model, written by the reference's own mj_saveModel
  <model>_traj.npz             seeded initial states + random controls + the reference mj_step
                               trajectory (PGS, Euler, fp64) and per-step integer observables
                               (ncon, nefc, solver_niter, contact geom pairs)

  cube_3x3x3.mjb / _steps.npz  BASELINE config 4 (model/cube/cube_3x3x3.xml as shipped: Newton, implicitfast,
                               26 convex mesh cubelets).  The .mjb is compiled from the reference's XML with
                               the <texture>/<material> assets removed (28 PNG skins = 36 MB of pixels that
                               mj_step never reads); every array mj_step does read is asserted equal to the
                               unstripped model's.  The fixture holds single steps: (state, warm start, ctrl)
                               -> next state + ncon / nefc / solver_niter along one reference trajectory of
                               random-action steps (contact sets flip at the 1e-16 level in this model, so
                               free-running trajectories are only comparable between bit-identical engines).

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


def strip_visuals(xml_text):
    """remove <texture>, <material> elements and material="..." attributes (rendering only)"""
    import re
    xml_text = re.sub(r"\s*<texture\b[^>]*/>", "", xml_text)
    xml_text = re.sub(r"\s*<material\b[^>]*/>", "", xml_text)
    return re.sub(r'\smaterial="[^"]*"', "", xml_text)


PHYSICS_ARRAYS = ["body_pos", "body_quat", "body_mass", "body_inertia", "body_ipos", "body_iquat", "jnt_axis", "jnt_pos",
                  "dof_armature", "dof_damping", "dof_frictionloss", "geom_pos", "geom_quat", "geom_size", "geom_rbound",
                  "geom_aabb", "geom_friction", "geom_solref", "geom_solimp", "geom_margin", "mesh_vert", "mesh_graph",
                  "mesh_polynormal", "mesh_polyvert", "mesh_polymap", "actuator_gear", "actuator_ctrlrange", "qpos0"]


def make_cube():
    import tempfile
    src = os.path.join(REF, "model/cube/cube_3x3x3.xml")
    full = rb.MjModel.from_xml_path(src)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "cube_3x3x3.xml")
        with open(path, "w") as f:
            f.write(strip_visuals(open(src).read()))
        m = rb.MjModel.from_xml_path(path)
    for name in PHYSICS_ARRAYS:
        assert np.array_equal(getattr(m, name), getattr(full, name)), name
    assert (m.nq, m.nv, m.nu, m.ngeom, m.nmesh) == (full.nq, full.nv, full.nu, full.ngeom, full.nmesh)
    m.save_binary(os.path.join(OUT, "cube_3x3x3.mjb"))
    d = rb.MjData(m)
    rng = np.random.Generator(np.random.PCG64(99))
    nstep = 160
    nstate = rb.mj_stateSize(m, rb.mjSTATE_FULLPHYSICS)
    lo, hi = m.actuator_ctrlrange[:, 0], m.actuator_ctrlrange[:, 1]
    state = np.zeros((nstep, nstate)); warm = np.zeros((nstep, m.nv)); ctrl = np.zeros((nstep, m.nu))
    nxt = np.zeros((nstep, nstate)); ints = np.zeros((nstep, 3), np.int32)
    rb.mj_resetData(m, d)
    for t in range(nstep):
        state[t] = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
        warm[t] = d.qacc_warmstart
        ctrl[t] = rng.uniform(lo, hi)
        d.ctrl[:] = ctrl[t]
        rb.mj_step(m, d)
        nxt[t] = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
        ints[t] = d.ncon, d.nefc, d.solver_niter[0]
    np.savez_compressed(os.path.join(OUT, "cube_3x3x3_steps.npz"), state=state, warmstart=warm, ctrl=ctrl, next=nxt, ints=ints)
    print("cube_3x3x3", "nv", m.nv, "ncon max", ints[:, 0].max(), "nefc max", ints[:, 1].max(), "niter max", ints[:, 2].max())


def make_jelly(name="jelly"):
    """BASELINE config 5: model/flex/jelly.xml as shipped (512-vertex solid flex, nv 1536, CG, dt 1 ms); with another name:
    that model of model/flex (trilinear / quadratic / sphere_trilinear: the interpolated flexes of round 5's GPU tests)."""
    import re
    import tempfile
    src = os.path.join(REF, "model/flex/%s.xml" % name)
    full = rb.MjModel.from_xml_path(src)
    # the included scene carries two 512 x 512 builtin textures (7 MB of pixels mj_step never reads): compiled from the
    # reference's XML with the <texture> / <material> elements removed, physics arrays asserted equal
    scene = open(os.path.join(REF, "model/flex/scene.xml")).read()
    scene = re.sub(r"<texture\b[^>]*?/>", "", scene, flags=re.S)
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "scene.xml"), "w") as f:
            f.write(strip_visuals(scene))
        with open(os.path.join(td, name + ".xml"), "w") as f:
            f.write(open(src).read())
        m = rb.MjModel.from_xml_path(os.path.join(td, name + ".xml"))
    for arr in PHYSICS_ARRAYS + ["flex_vert", "flex_elem", "flex_edge", "flex_stiffness", "flexedge_length0", "flex_vertbodyid",
                                  "bvh_child", "bvh_nodeid", "flex_elemlayer", "geom_type", "body_invweight0", "flex_vert0", "flex_node0",
                                  "flex_nodebodyid", "flex_interp", "flex_cellnum"]:
        assert np.array_equal(getattr(m, arr), getattr(full, arr)), arr
    assert (m.nq, m.nv, m.ngeom, m.nflexelem, m.nbvh) == (full.nq, full.nv, full.ngeom, full.nflexelem, full.nbvh)
    m.save_binary(os.path.join(OUT, name + ".mjb"))
    print(name, "nv", m.nv, "nflexvert", m.nflexvert, "nflexelem", m.nflexelem)


def main():
    os.makedirs(OUT, exist_ok=True)
    make_jelly()
    for other in ("trilinear", "quadratic", "sphere_trilinear"):
        make_jelly(other)
    if "--jelly-only" in sys.argv or "--flex-only" in sys.argv:
        return
    make_cube()
    if "--cube-only" in sys.argv:
        return
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
