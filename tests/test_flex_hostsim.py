"""Flexes (mjh_flex.h, mjh_flexcol.h) on the host wavefront emulation against the oracle.

Everything is held to bit equality: vertex positions, edge lengths / Jacobians / velocities, finite-element stretch,
shell bending, edge dampers; contact lists; and the constraint solve -- these models (nv of several hundred to 1536,
one island spanning the flex, CG) keep the Jacobian compressed with explicit column indices (mjh_csr.h) and sum in the
reference's order, so states, contact / row counts and CG iteration counts of free-running trajectories are identical."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from mujoco_amd import _capi as K

VERTS = 'count="{n}" spacing=".05 .05 .05"'


def flex_xml(count, pos, extra_world="", flex_attr='dim="3"', flex_body='<edge damping="1"/><contact selfcollide="none"/><elasticity young="5e4"/>',
             option='solver="CG" tolerance="1e-6" timestep=".001" integrator="Euler"'):
    return f"""
<mujoco>
  <option {option}/>
  <size memory="10M"/>
  <worldbody>
    <geom name="floor" type="plane" size="0 0 .05"/>
    {extra_world}
    <flexcomp type="grid" count="{count}" spacing=".05 .05 .05" pos="{pos}" {flex_attr} radius=".005" mass="2" name="soft">
      {flex_body}
    </flexcomp>
  </worldbody>
</mujoco>"""


def _fields_exact(b, d, names):
    bad = []
    for f in names:
        a = b.get(f)[0]
        r = np.asarray(getattr(d, f)).ravel()
        if not np.array_equal(a[:len(r)], r):
            bad.append((f, float(np.abs(a[:len(r)] - r).max())))
    return bad


def _free_motion(rb, lib, m, nstep=3):
    dm = K.DeviceModel(lib, m)
    b = K.Batch(dm, 1)
    d = rb.MjData(m)
    rng = np.random.default_rng(0)
    d.qvel[:] = rng.normal(0, .1, m.nv)
    s = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
    b.set("qpos", s[None, 1:1 + m.nq]); b.set("qvel", s[None, 1 + m.nq:])
    rb.mj_forward(m, d); b.forward()
    assert _fields_exact(b, d, ["flexvert_xpos", "flexedge_length", "flexedge_velocity", "flexedge_J", "qfrc_spring",
                                "qfrc_damper", "qfrc_passive", "qacc"]) == []
    assert np.abs(d.qfrc_spring).max() > 0 or np.abs(d.qfrc_damper).max() > 0
    for _ in range(nstep):
        b.step(); rb.mj_step(m, d)
    assert np.array_equal(b.get("qpos")[0], d.qpos) and np.array_equal(b.get("qvel")[0], d.qvel)


def test_jelly_free_motion_bit_exact(rb, hostsim_lib):
    """BASELINE config 5's model: 512 vertices, 2863 edges, 2058 tetrahedra; stretch + edge damping, no contact yet."""
    m = rb.MjModel.from_binary_path(os.path.join(GOLDEN, "jelly.mjb"))
    assert (m.nv, m.nflexvert, m.nflexelem) == (1536, 512, 2058)
    _free_motion(rb, hostsim_lib, m)


def test_shell_bending_free_motion_bit_exact(rb, hostsim_lib, tmp_path):
    """a triangle shell with bending stiffness, Rayleigh damping and pinned corners (mj_flexPassiveBend, :459-547)"""
    xml = tmp_path / "shell.xml"
    xml.write_text(flex_xml("6 6 1", "0 0 1", flex_attr='dim="2"',
                            flex_body='<edge equality="false" damping="10"/><contact contype="0" conaffinity="0"/>'
                                      '<elasticity young="3e5" poisson="0" thickness="1e-2" damping="1e-3" elastic2d="both"/><pin id="0 5 30 35"/>'))
    m = rb.MjModel.from_xml_path(str(xml))
    assert m.flex_dim[0] == 2 and m.flex_bendingadr[0] >= 0
    _free_motion(rb, hostsim_lib, m)


def _resync_steps(rb, lib, m, pre, nstep, mocap=None, csr=1):
    """the oracle's trajectory from reset: after `pre` steps the kernels take over its state once and both run `nstep`
    steps freely; states, ncon / nefc and CG iteration counts identical at every step"""
    dm = K.DeviceModel(lib, m)
    assert dm.size("csr") == csr
    b = K.Batch(dm, 1)
    d = rb.MjData(m)
    if mocap is not None: d.mocap_pos[:] = mocap
    for _ in range(pre): rb.mj_step(m, d)
    s = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
    b.set("time", s[None, :1]); b.set("qpos", s[None, 1:1 + m.nq]); b.set("qvel", s[None, 1 + m.nq:])
    b.set("qacc_warmstart", d.qacc_warmstart[None, :])
    if m.nmocap: b.set("mocap_pos", d.mocap_pos.reshape(1, -1)); b.set("mocap_quat", d.mocap_quat.reshape(1, -1))
    maxcon = 0; kinds = set(); iters = 0
    for t in range(nstep):
        b.step(); rb.mj_step(m, d)
        c = b.get("counts")[0]
        assert (c[0], c[1], c[5]) == (d.ncon, d.nefc, d.solver_niter[0]), (t, c[:6], d.ncon, d.nefc, d.solver_niter[0])
        assert not b.get("warning")[0].any()
        assert np.array_equal(b.get("qpos")[0], d.qpos) and np.array_equal(b.get("qvel")[0], d.qvel), t
        maxcon = max(maxcon, d.ncon); iters += int(d.solver_niter[0])
        if d.ncon:
            con = d.contact[:d.ncon]
            kinds |= {"vert"} if (np.asarray(con["vert"])[:, 1] >= 0).any() else set()
            kinds |= {"elem"} if (np.asarray(con["elem"])[:, 1] >= 0).any() else set()
    assert iters > 0
    return maxcon, kinds


def _contact_lists(rb, hostsim_lib, tmp_path):
    xml = tmp_path / "mix.xml"
    xml.write_text(flex_xml("5 5 3", "0 0 .12", extra_world='''
    <geom name="wall" type="plane" size=".5 .5 .05" zaxis="1 0 0" pos="-.12 0 0"/>
    <body mocap="true" pos=".06 .06 .03"><geom type="sphere" size=".05" condim="1"/></body>
    <body mocap="true" pos="-.05 -.05 .03"><geom type="box" size=".04 .03 .03" euler="10 20 30"/></body>
    <body mocap="true" pos=".05 -.06 .04" zaxis="1 .3 0"><geom type="capsule" size=".03 .05"/></body>'''))
    m = rb.MjModel.from_xml_path(str(xml))
    dm = K.DeviceModel(hostsim_lib, m)
    b = K.Batch(dm, 1)
    d = rb.MjData(m)
    seen = set()
    for pre in (60, 90, 140):
        rb.mj_resetData(m, d)
        for _ in range(pre): rb.mj_step(m, d)
        s = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
        b.set("qpos", s[None, 1:1 + m.nq]); b.set("qvel", s[None, 1 + m.nq:]); b.set("qacc_warmstart", d.qacc_warmstart[None, :])
        b.set("mocap_pos", d.mocap_pos.reshape(1, -1)); b.set("mocap_quat", d.mocap_quat.reshape(1, -1))
        rb.mj_forward(m, d); b.forward()
        n = d.ncon
        assert b.get("counts")[0][0] == n and n > 0
        con = d.contact[:n]
        assert np.array_equal(b.get("con_dist")[0][:n], con["dist"])
        assert np.array_equal(b.get("con_pos")[0][:3*n], np.asarray(con["pos"]).ravel())
        assert np.array_equal(b.get("con_frame")[0][:9*n], np.asarray(con["frame"]).ravel())
        assert np.array_equal(b.get("con_geom")[0][:2*n].reshape(-1, 2), np.asarray(con["geom"]))
        cf = b.get("con_flex")[0][:6*n].reshape(-1, 6)
        assert np.array_equal(cf[:, 0], np.asarray(con["flex"])[:, 1])
        assert np.array_equal(cf[:, 1], np.asarray(con["elem"])[:, 1]) and np.array_equal(cf[:, 2], np.asarray(con["vert"])[:, 1])
        assert _fields_exact(b, d, ["efc_pos", "efc_margin", "efc_D", "efc_R"]) == []
        seen |= set(np.asarray(con["geom"])[:, 0].tolist())
    assert len(seen) >= 3, seen


def test_flex_contact_lists_exact(rb, hostsim_lib, tmp_path):
    """vertex-plane and element-geom (sphere, box, capsule: GJK / EPA against tetrahedra) contacts of one forward pass:
    every contact record and every constraint row parameter identical to the oracle's"""
    _contact_lists(rb, hostsim_lib, tmp_path)


def test_flex_on_floor_keeps_fifty_contacts(rb, hostsim_lib, tmp_path):
    """81 vertices of the bottom layer reach the floor: filterFlexContacts keeps mjMAXCONPAIR = 50 of them (deepest first,
    then farthest-point sampling with the reference's positional bookkeeping), sorted by vertex"""
    xml = tmp_path / "floor.xml"
    xml.write_text(flex_xml("9 9 2", "0 0 .035"))
    m = rb.MjModel.from_xml_path(str(xml))
    maxcon, kinds = _resync_steps(rb, hostsim_lib, m, pre=30, nstep=40)
    assert maxcon == 50 and kinds == {"vert"}


def _edge_equality(rb, lib, tmp_path, nstep=40):
    xml = tmp_path / "softbox.xml"
    xml.write_text(flex_xml("5 4 4", "0 0 .09", extra_world='<body mocap="true" pos=".02 .01 .04"><geom type="sphere" size=".03"/></body>',
                            flex_body='<edge equality="true"/><contact selfcollide="none"/>'))
    m = rb.MjModel.from_xml_path(str(xml))
    assert m.neq == 1 and m.eq_type[0] == 4 and m.flex_edgeequality[0] == 1 and m.nv > 128
    d = rb.MjData(m)
    rb.mj_forward(m, d)
    assert d.ne == int((np.asarray(m.flexedge_rigid) == 0).sum()) and d.ne > 100
    return _resync_steps(rb, lib, m, pre=60, nstep=nstep)


def test_flex_edge_equality_constraints(rb, hostsim_lib, tmp_path):
    """mjEQ_FLEX (model/flex/softbox.xml's kind): one equality row per non-rigid edge -- the edge's flexedge_J row, position
    error length - length0, diagApprox from flexedge_invweight0 -- in front of the contact rows; the box lands on the floor
    and on a sphere: states, ncon / nefc and CG iteration counts identical to the oracle's at every step"""
    maxcon, kinds = _edge_equality(rb, hostsim_lib, tmp_path)
    assert maxcon > 0


def test_jelly_on_the_capsule(rb, hostsim_lib):
    """jelly.xml falling onto its capsule: element contacts, one island of 1536 dofs under CG"""
    m = rb.MjModel.from_binary_path(os.path.join(GOLDEN, "jelly.mjb"))
    maxcon, kinds = _resync_steps(rb, hostsim_lib, m, pre=385, nstep=30)
    assert maxcon >= 20 and "elem" in kinds


def _actuated_body(rb, hostsim_lib, tmp_path):
    xml = tmp_path / "drum.xml"
    xml.write_text(f"""
<mujoco>
  <option solver="CG" tolerance="1e-6" timestep=".001"/>
  <size memory="20M"/>
  <worldbody>
    <geom type="plane" size="0 0 .05"/>
    <flexcomp type="grid" count="8 3 3" spacing=".05 .05 .05" pos="0 0 .36" radius="0" name="soft" dim="3" mass="3">
      <contact condim="3" solref="0.01 1" solimp=".95 .99 .0001" selfcollide="none"/>
      <elasticity young="5e4" damping="0.002" poisson="0.2"/>
    </flexcomp>
    <body><joint name="hinge" pos="0 0 .15" axis="0 1 0" damping="5"/><geom type="cylinder" size=".12" fromto="0 -.2 .15 0 .2 .15" density="300"/></body>
  </worldbody>
  <actuator><motor joint="hinge" ctrlrange="-10 10"/></actuator>
</mujoco>""")
    m = rb.MjModel.from_xml_path(str(xml))
    dm = K.DeviceModel(hostsim_lib, m)
    b = K.Batch(dm, 1)
    d = rb.MjData(m)
    d.ctrl[:] = 3.0
    for _ in range(160): rb.mj_step(m, d)
    assert d.ncon > 0
    s = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
    b.set("time", s[None, :1]); b.set("qpos", s[None, 1:1 + m.nq]); b.set("qvel", s[None, 1 + m.nq:])
    b.set("qacc_warmstart", d.qacc_warmstart[None, :]); b.set("ctrl", d.ctrl[None, :])
    for t in range(40):
        b.step(); rb.mj_step(m, d)
        c = b.get("counts")[0]
        assert (c[0], c[1], c[5]) == (d.ncon, d.nefc, d.solver_niter[0]), t
        assert np.array_equal(b.get("qpos")[0], d.qpos) and np.array_equal(b.get("qvel")[0], d.qvel), t
    assert d.ncon > 0


def test_flex_against_an_actuated_body(rb, hostsim_lib, tmp_path):
    """a solid flex draped over a hinged, motor-driven cylinder (the geom's own dofs enter the element rows with weight -1)"""
    _actuated_body(rb, hostsim_lib, tmp_path)


MULTI_ISLAND_XML = """
<mujoco>
  <option solver="CG" tolerance="1e-6" timestep=".001" integrator="Euler"/>
  <size memory="10M"/>
  <worldbody>
    <geom name="floor" type="plane" size="0 0 .05"/>
    <body pos=".6 0 .05"><freejoint/><geom type="box" size=".05 .04 .03"/></body>
    <body pos=".6 .4 .04"><freejoint/><geom type="sphere" size=".04"/></body>
    <body pos="-.6 0 .2"><joint type="hinge" axis="0 1 0"/><geom type="capsule" size=".03 .1"/></body>
    <flexcomp type="grid" count="6 6 3" spacing=".05 .05 .05" pos="0 0 .08" dim="3" radius=".005" mass="2" name="soft">
      <edge damping="1"/><contact selfcollide="none"/><elasticity young="5e4"/>
    </flexcomp>
  </worldbody>
</mujoco>"""


def _multi_island(rb, lib, tmp_path):
    xml = tmp_path / "multi.xml"
    xml.write_text(MULTI_ISLAND_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    d = rb.MjData(m)
    rb.mj_forward(m, d)
    _resync_steps(rb, lib, m, pre=20, nstep=60)
    d = rb.MjData(m)
    for _ in range(60): rb.mj_step(m, d)
    assert d.nisland >= 2


def test_flex_island_next_to_rigid_islands(rb, hostsim_lib, tmp_path):
    """a flex resting on the floor (one island of all its trees), a box and a sphere on the floor (an island each) and an
    unconstrained pendulum: island-local dot products over each island's dof list"""
    _multi_island(rb, hostsim_lib, tmp_path)


def _two_flexes(rb, lib, tmp_path):
    """two flexes on the floor: the mass matrix is diagonal (every dof a slider), so the solver lays its vectors out for the
    fused passes of mjh_csrpass.h -- but neither island spans every dof, so each is solved by the unfused code on that
    layout (Ma / Mv / Mgrad in global memory, the difference vectors in the product vectors' bytes): states, counts and CG
    iteration counts identical to the oracle's"""
    xml = tmp_path / "two.xml"
    xml.write_text("""
<mujoco>
  <option solver="CG" tolerance="1e-6" timestep=".001" integrator="Euler"/>
  <size memory="10M"/>
  <worldbody>
    <geom name="floor" type="plane" size="0 0 .05" contype="3" conaffinity="3"/>
    <flexcomp type="grid" count="4 4 3" spacing=".05 .05 .05" pos="-.3 0 .06" dim="3" radius=".005" mass="1" name="a">
      <edge damping="1"/><contact selfcollide="none" contype="1" conaffinity="1"/><elasticity young="5e4"/>
    </flexcomp>
    <flexcomp type="grid" count="4 3 3" spacing=".05 .05 .05" pos=".3 0 .07" dim="3" radius=".005" mass="1" name="b">
      <edge damping="1"/><contact selfcollide="none" contype="2" conaffinity="2"/><elasticity young="4e4"/>
    </flexcomp>
  </worldbody>
</mujoco>""")
    m = rb.MjModel.from_xml_path(str(xml))
    assert m.nflex == 2 and m.nv == 3*(48 + 36) and m.nv > 128
    d = rb.MjData(m)
    for _ in range(80): rb.mj_step(m, d)
    assert d.nisland == 2, d.nisland
    maxcon, kinds = _resync_steps(rb, lib, m, pre=60, nstep=40)
    assert maxcon >= 12 and kinds == {"vert"}


def test_two_flexes_are_two_islands_of_sliders(rb, hostsim_lib, tmp_path):
    _two_flexes(rb, hostsim_lib, tmp_path)


SHELL_GEOMS = """
    <body mocap="true" pos=".06 .06 .15"><geom type="sphere" size=".05"/></body>
    <body mocap="true" pos="-.07 -.05 .15"><geom type="box" size=".04 .03 .03" euler="10 20 30"/></body>
    <body mocap="true" pos=".05 -.08 .16" zaxis="1 .3 0"><geom type="capsule" size=".03 .05"/></body>
    <body mocap="true" pos="-.06 .07 .15" zaxis="1 .3 .2"><geom type="cylinder" size=".04 .05"/></body>"""


def shell_xml(count, geoms, dim=2, body='<edge equality="false" damping="1"/><contact selfcollide="none"/>'
                                        '<elasticity young="3e4" poisson="0" thickness="1e-2" elastic2d="both"/>',
              option='solver="CG" tolerance="1e-6" timestep=".001" integrator="Euler"', spacing=".04 .04 .04", pos="0 0 .25"):
    return f"""
<mujoco>
  <option {option}/>
  <size memory="50M"/>
  <worldbody>
    <geom name="floor" type="plane" size="0 0 .05"/>
    {geoms}
    <flexcomp type="grid" count="{count}" spacing="{spacing}" pos="{pos}" dim="{dim}" radius=".005" mass="1" name="soft">
      {body}
    </flexcomp>
  </worldbody>
</mujoco>"""


def _free_run(rb, lib, m, pre, nstep, csr=None):
    """as _resync_steps, for models on any Jacobian path; returns (max contacts, first-side geoms seen in contacts)"""
    dm = K.DeviceModel(lib, m)
    if csr is not None: assert dm.size("csr") == csr
    b = K.Batch(dm, 1)
    d = rb.MjData(m)
    for _ in range(pre): rb.mj_step(m, d)
    s = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
    b.set("time", s[None, :1]); b.set("qpos", s[None, 1:1 + m.nq]); b.set("qvel", s[None, 1 + m.nq:1 + m.nq + m.nv])
    b.set("qacc_warmstart", d.qacc_warmstart[None, :])
    if m.nmocap: b.set("mocap_pos", d.mocap_pos.reshape(1, -1)); b.set("mocap_quat", d.mocap_quat.reshape(1, -1))
    maxcon = 0; geoms = set()
    for t in range(nstep):
        b.step(); rb.mj_step(m, d)
        c = b.get("counts")[0]
        assert (c[0], c[1], c[5]) == (d.ncon, d.nefc, d.solver_niter[0]), (t, c[:6], d.ncon, d.nefc, d.solver_niter[0])
        assert not b.get("warning")[0].any()
        assert np.array_equal(b.get("qpos")[0], d.qpos) and np.array_equal(b.get("qvel")[0], d.qvel), t
        maxcon = max(maxcon, d.ncon)
        if d.ncon: geoms |= set(np.asarray(d.contact[:d.ncon]["geom"])[:, 0].tolist())
    return maxcon, geoms


def _shell_on_geoms(rb, lib, tmp_path, count="9 9 1", csr=1, geoms=SHELL_GEOMS):
    xml = tmp_path / "shellgeoms.xml"
    xml.write_text(shell_xml(count, geoms))
    m = rb.MjModel.from_xml_path(str(xml))
    maxcon, seen = _free_run(rb, lib, m, pre=60, nstep=100, csr=csr)
    assert seen >= {1, 2, 3, 4}, seen
    return maxcon


def test_shell_flex_on_sphere_box_capsule_cylinder(rb, hostsim_lib, tmp_path):
    """a triangle shell falling on a sphere, a box, a capsule (mjraw_SphereTriangle / BoxTriangle / CapsuleTriangle: up to
    1 / 11 / 5 contacts per triangle) and a cylinder (GJK / EPA against the triangle + mjc_fixNormal): contact, row and CG
    iteration counts and the states of 100 free-running steps identical to the oracle's; explicit-index rows (243 dofs)"""
    assert _shell_on_geoms(rb, hostsim_lib, tmp_path) > 40


def test_shell_flex_on_geoms_dense_rows(rb, hostsim_lib, tmp_path):
    """the same colliders with the constraint Jacobian dense (48 dofs)"""
    _shell_on_geoms(rb, hostsim_lib, tmp_path, count="4 4 1", csr=0, geoms=SHELL_GEOMS.replace(".06 .06 .15", ".02 .02 .15"))


SHELL_ELLIPSOIDS = """
    <body mocap="true" pos=".05 .05 .15" euler="20 -15 30"><geom type="ellipsoid" size=".06 .04 .03"/></body>
    <body mocap="true" pos="-.06 -.05 .16" euler="0 40 10"><geom type="ellipsoid" size=".03 .07 .05"/></body>"""


def _shell_on_ellipsoids(rb, lib, tmp_path):
    xml = tmp_path / "shellell.xml"
    xml.write_text(shell_xml("9 9 1", SHELL_ELLIPSOIDS))
    m = rb.MjModel.from_xml_path(str(xml))
    maxcon, seen = _free_run(rb, lib, m, pre=60, nstep=100, csr=1)
    assert seen >= {1, 2}, seen
    return maxcon


def test_shell_flex_on_ellipsoids(rb, hostsim_lib, tmp_path):
    """a triangle shell falling on two ellipsoids: GJK / EPA against the triangle, then mjc_fixNormal's ellipsoid case (the
    surface normal at the point closest to the contact: ray projection from inside, Newton's method on the QCQP
    multiplier from outside; engine_collision_convex.c:1306-1408) -- round 6"""
    assert _shell_on_ellipsoids(rb, hostsim_lib, tmp_path) > 10


def _line_on_cylinder(rb, lib, tmp_path):
    xml = tmp_path / "line.xml"
    xml.write_text(shell_xml("50 1 1", '<body mocap="true" pos="0 0 .15" zaxis="0 1 0"><geom type="cylinder" size=".05 .1"/></body>'
                                        '<body mocap="true" pos=".15 0 .12"><geom type="ellipsoid" size=".05 .08 .04"/></body>',
                             dim=1, body='<edge equality="false" damping=".5" stiffness="200"/><contact selfcollide="none"/>', spacing=".01 .01 .01"))
    m = rb.MjModel.from_xml_path(str(xml))
    maxcon, seen = _free_run(rb, lib, m, pre=100, nstep=100, csr=1)
    assert seen >= {1, 2}, seen


def test_line_flex_on_cylinder_and_ellipsoid(rb, hostsim_lib, tmp_path):
    """a cable (line elements: capsules of two vertices) on a cylinder and an ellipsoid: mjc_ConvexElem with two-corner elements"""
    _line_on_cylinder(rb, hostsim_lib, tmp_path)


def _self_collision(rb, lib, tmp_path, selfcollide, nstep=60):
    """a small cloth falling over a thin bar folds and touches itself; returns the largest number of flex : flex contacts"""
    xml = tmp_path / "fold.xml"
    bar = '<body mocap="true" pos="0 0 .2" zaxis="0 1 0"><geom type="capsule" size=".008 .3"/></body>'
    xml.write_text(shell_xml("7 7 1", bar, pos="0 0 .23", body=f'<edge equality="false" damping="1"/><contact selfcollide="{selfcollide}"/>'
                             '<elasticity young="3e4" poisson="0" thickness="1e-2" elastic2d="both"/>').replace('mass="1"', 'mass=".3"'))
    m = rb.MjModel.from_xml_path(str(xml))
    dm = K.DeviceModel(lib, m)
    assert dm.size("csr") == 1
    b = K.Batch(dm, 1)
    d = rb.MjData(m)
    for _ in range(190): rb.mj_step(m, d)
    s = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
    b.set("time", s[None, :1]); b.set("qpos", s[None, 1:1 + m.nq]); b.set("qvel", s[None, 1 + m.nq:1 + m.nq + m.nv])
    b.set("qacc_warmstart", d.qacc_warmstart[None, :])
    b.set("mocap_pos", d.mocap_pos.reshape(1, -1)); b.set("mocap_quat", d.mocap_quat.reshape(1, -1))
    nself = 0
    for t in range(nstep):
        b.step(); rb.mj_step(m, d)
        c = b.get("counts")[0]
        assert (c[0], c[1], c[5]) == (d.ncon, d.nefc, d.solver_niter[0]), (t, c[:6], d.ncon, d.nefc, d.solver_niter[0])
        assert not b.get("warning")[0].any()
        assert np.array_equal(b.get("qpos")[0], d.qpos) and np.array_equal(b.get("qvel")[0], d.qvel), t
        if d.ncon:
            con = d.contact[:d.ncon]
            both = np.asarray(con["flex"])[:, 0] >= 0
            nself = max(nself, int(both.sum()))
            if t == nstep - 1:
                # the contacts' identity: flex / element of side 1, then of side 0
                cf = b.get("con_flex")[0][:6*d.ncon].reshape(-1, 6)
                assert np.array_equal(cf[:, 1], np.asarray(con["elem"])[:, 1]) and np.array_equal(cf[:, 4], np.asarray(con["elem"])[:, 0])
                assert np.array_equal(cf[:, 3], np.asarray(con["flex"])[:, 0])
    return nself


def test_shell_flex_self_collision_sweep_and_prune(rb, hostsim_lib, tmp_path):
    """selfcollide = auto on a shell: mj_collideFlexSAP's sweep over float-rounded element boxes along the longest axis of
    the flex's root box (recomputed bottom-up like mj_updateDynamicBVH), triangle : triangle GJK / EPA, two-sided weighted
    contact rows, filterFlexContacts on more than 50 flex : flex contacts -- the sweep's emission order included"""
    assert _self_collision(rb, hostsim_lib, tmp_path, "auto") == 50


def test_shell_flex_self_collision_through_the_hierarchy(rb, hostsim_lib, tmp_path):
    """selfcollide = bvh (what auto selects for solid flexes): mj_collideTree of the flex's bounding volume hierarchy against
    itself (engine_collision_driver.c:855-858, :1053-1240) -- every pair of leaves whose boxes overlap, in the ORIENTATION the
    walk reaches it in (node pairs with node1 > node2 are dropped when popped) and in the walk's order, which decides which 50
    contacts filterFlexContacts keeps"""
    assert _self_collision(rb, hostsim_lib, tmp_path, "bvh") == 50


def test_shell_flex_self_collision_all_pairs(rb, hostsim_lib, tmp_path):
    """selfcollide = narrow: every pair of active elements in lexicographic order"""
    assert _self_collision(rb, hostsim_lib, tmp_path, "narrow", nstep=40) > 10


RADIAL_XML = """
<mujoco>
  <option solver="CG" tolerance="1e-6" timestep=".002"/>
  <size memory="50M"/>
  <worldbody>
    <geom name="floor" type="plane" size="0 0 .05"/>
    <body mocap="true" pos=".1 .05 .1"><geom type="sphere" size=".08"/></body>
    <body pos="0 0 .35" name="body">
      <freejoint/>
      <geom size=".05" contype="0" conaffinity="0"/>
      <flexcomp name="softbox" type="box" count="6 6 6" spacing=".04 .04 .04" radius="0.01" dim="3" dof="radial">
        <contact internal="false" selfcollide="none"/>
        <edge equality="true"/>
      </flexcomp>
    </body>
  </worldbody>
</mujoco>"""

ON_HINGE_XML = """
<mujoco>
  <option solver="CG" tolerance="1e-6" timestep=".001"/>
  <size memory="50M"/>
  <worldbody>
    <geom name="floor" type="plane" size="0 0 .05"/>
    <body mocap="true" pos=".25 .03 .42"><geom type="sphere" size=".06"/></body>
    <body name="pole" pos="0 0 .6">
      <joint type="hinge" axis="0 0 1" damping=".01"/>
      <joint type="slide" axis="0 0 1" damping="1" stiffness="50"/>
      <geom type="capsule" size=".01 .2" contype="0" conaffinity="0"/>
      <flexcomp type="grid" count="7 8 1" spacing=".04 .04 .04" pos=".14 0 0" zaxis="0 1 0" mass=".2" name="flag" radius="0.004" dim="2">
        <edge equality="false" damping=".02"/>
        <contact selfcollide="none"/>
        <elasticity young="3e4" poisson="0" thickness="4e-3" elastic2d="stretch"/>
        <pin id="0 1 2 3"/>
      </flexcomp>
    </body>
  </worldbody>
</mujoco>"""


def _articulated_vertices(rb, lib, tmp_path, which):
    xml = tmp_path / "art.xml"
    xml.write_text(RADIAL_XML if which == "radial" else ON_HINGE_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    vb = np.asarray(m.flex_vertbodyid)
    assert (np.asarray(m.body_simple)[vb] != 2).all() and m.nv > 128
    dm = K.DeviceModel(lib, m)
    b = K.Batch(dm, 1)
    d = rb.MjData(m)
    if which != "radial": d.qvel[:2] = (3.0, 0.5)
    pre = 100 if which == "radial" else 150
    for _ in range(pre): rb.mj_step(m, d)
    s = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
    def load():
        b.set("time", s[None, :1]); b.set("qpos", s[None, 1:1 + m.nq]); b.set("qvel", s[None, 1 + m.nq:1 + m.nq + m.nv])
        b.set("qacc_warmstart", d.qacc_warmstart[None, :])
        b.set("mocap_pos", d.mocap_pos.reshape(1, -1)); b.set("mocap_quat", d.mocap_quat.reshape(1, -1))
    load()
    rb.mj_forward(m, d); b.forward()
    assert _fields_exact(b, d, ["flexvert_xpos", "flexedge_length", "flexedge_velocity", "flexedge_J", "qfrc_spring", "qfrc_damper",
                                "qfrc_passive", "qacc_smooth", "qacc"]) == []
    load()
    maxcon = 0
    for t in range(80):
        b.step(); rb.mj_step(m, d)
        c = b.get("counts")[0]
        assert (c[0], c[1], c[5]) == (d.ncon, d.nefc, d.solver_niter[0]), (t, c[:6], d.ncon, d.nefc, d.solver_niter[0])
        assert not b.get("warning")[0].any()
        assert np.array_equal(b.get("qpos")[0], d.qpos) and np.array_equal(b.get("qvel")[0], d.qvel), t
        maxcon = max(maxcon, d.ncon)
    assert maxcon > 0
    return maxcon


def test_flex_vertices_on_radial_sliders_of_a_free_body(rb, hostsim_lib, tmp_path):
    """model/flex/softbox.xml's kind: every vertex a radial slider under one free body -- edge Jacobians over merged chains
    (common dofs kept), contact rows as weighted sums over the union of the corners' chains, edge equalities; lands on
    the floor and a sphere"""
    _articulated_vertices(rb, hostsim_lib, tmp_path, "radial")


def test_flex_vertices_riding_on_a_hinged_body(rb, hostsim_lib, tmp_path):
    """a cloth whose vertex bodies are children of a spinning, sliding pole (four of them pinned to it): the stretch force of
    every vertex goes through mj_applyFT to the pole's dofs, summed in vertex order; contact with a sphere"""
    _articulated_vertices(rb, hostsim_lib, tmp_path, "hinge")


def _flex_on_mask_path(rb, lib, tmp_path, solver, equality):
    """a shell of 75 dofs with jacobian = sparse: the 128-bit mask rows of mjh_sparse.h carry flex contact rows (weighted
    sums over the union of the corners' chains) and, with equality, flex edge constraint rows"""
    opt = f'solver="{solver}" tolerance="1e-8" timestep=".001" integrator="Euler" jacobian="sparse"'
    body = ('<edge equality="true" damping="1"/><contact selfcollide="none"/>' if equality else
            '<edge equality="false" damping="1"/><contact selfcollide="none"/><elasticity young="3e4" poisson="0" thickness="1e-2" elastic2d="both"/>')
    xml = tmp_path / "mask.xml"
    xml.write_text(shell_xml("5 5 1", SHELL_GEOMS.replace(".06 .06 .15", ".02 .02 .15"), option=opt, body=body))
    m = rb.MjModel.from_xml_path(str(xml))
    assert K.DeviceModel(lib, m).size("sparse") == 1
    maxcon, seen = _free_run(rb, lib, m, pre=60, nstep=80, csr=0)
    assert maxcon > 30 and seen >= {1, 2, 3, 4}


@pytest.mark.parametrize("solver,equality", [("Newton", True), ("Newton", False), ("CG", True)])
def test_flex_contacts_and_edge_constraints_on_mask_rows(rb, hostsim_lib, tmp_path, solver, equality):
    _flex_on_mask_path(rb, hostsim_lib, tmp_path, solver, equality)


def _hanging_cloth(rb, lib, tmp_path):
    """model/flex/flag.xml's kind: edge constraints plus a connect equality (and a weld) on the explicit-index rows"""
    xml = tmp_path / "hang.xml"
    xml.write_text(shell_xml("7 7 1", '<body mocap="true" pos=".05 0 .1"><geom type="sphere" size=".06"/></body>', pos="0 0 .3",
                             body='<edge equality="true" damping=".01"/><contact selfcollide="none"/>').replace(
        "</worldbody>", '</worldbody><equality><connect body1="soft_0" anchor="0 0 0"/><weld body1="soft_48" body2="soft_47"/></equality>'))
    m = rb.MjModel.from_xml_path(str(xml))
    assert m.neq == 3
    maxcon, seen = _free_run(rb, lib, m, pre=150, nstep=80, csr=1)
    assert maxcon > 0


def test_connect_and_weld_rows_next_to_flex_edge_constraints(rb, hostsim_lib, tmp_path):
    _hanging_cloth(rb, hostsim_lib, tmp_path)


CABLE_XML = """
<mujoco>
  <option solver="CG" tolerance="1e-6" timestep=".002" integrator="Euler" jacobian="sparse"/>
  <size memory="50M"/>
  <worldbody>
    <geom name="floor" type="plane" size="0 0 .05"/>
    <body mocap="true" pos=".0 .05 .12"><geom type="sphere" size=".05"/></body>
    <body mocap="true" pos="0 -.06 .06" zaxis="1 0 0"><geom type="capsule" size=".02 .15"/></body>
    <flexcomp name="cable" type="circle" count="COUNT 1 1" spacing=".03 1 1" dim="1" radius="0.008" pos="0 0 .3" mass=".2" euler="80 0 0">
      <edge equality="true" damping=".002"/>
      <contact selfcollide="auto"/>
    </flexcomp>
  </worldbody>
</mujoco>"""


def _cable(rb, lib, tmp_path, count, pre, kind=None):
    """a ring of line elements (model/flex/pulley.xml's kind) collapsing over a sphere and a capsule onto the floor: the
    elements are capsules made of vertex pairs (mj_makeCapsule: frame through mju_quatZ2Vec -- atan2, sin, cos), against
    the geoms (mjraw_SphereCapsule / CapsuleCapsule) and against each other (sweep-and-prune + mjraw_CapsuleCapsule)"""
    xml = tmp_path / "cable.xml"
    xml.write_text(CABLE_XML.replace("COUNT", str(count)))
    m = rb.MjModel.from_xml_path(str(xml), kind=kind)
    maxcon, seen = _free_run(rb, lib, m, pre, 160)
    assert -1 in seen and 1 in seen, seen
    return seen


def test_line_flex_self_collision_and_capsule_colliders(rb, hostsim_lib, tmp_path):
    assert 2 in _cable(rb, hostsim_lib, tmp_path, 30, 150)            # mask rows (87 dofs): sphere, capsule, itself


def test_line_flex_self_collision_explicit_index_rows(rb, hostsim_lib, tmp_path):
    _cable(rb, hostsim_lib, tmp_path, 50, 110)                        # 147 dofs


EFM_OPTION = 'solver="CG" tolerance="1e-6" timestep=".001" integrator="implicitfast"'


def _effective_metric(rb, lib, tmp_path, which):
    """mj_flexCG (engine_forward.c:1640): CG + implicitfast + a flex with stretch stiffness -> the solve runs in the metric
    M + K.  Checked field by field against the oracle's arena arrays (efm_K_val: the assembled stiffness; efm_L: the
    factored 3 x 3 blocks; efm_c: the shift through the stencil operators), qacc_smooth (PCG with the block
    preconditioner), then free-running steps with contacts (Ma / Mv / Mgrad through the metric, monolithic solve, mj_advance
    with the solver's qacc): states, contact / row counts and CG iteration counts identical."""
    xml = tmp_path / "efm.xml"
    if which == "solid":
        xml.write_text(flex_xml("5 5 3", "0 0 .09", option=EFM_OPTION,
                                extra_world='<body mocap="true" pos=".02 .01 .04"><geom type="sphere" size=".03"/></body>'
                                            '<body pos=".3 0 .1"><freejoint/><geom type="box" size=".04 .04 .04"/></body>',
                                flex_body='<edge damping="1"/><contact selfcollide="none"/><elasticity young="5e4" damping="1e-3"/><pin id="0 1"/>'))
    else:
        xml.write_text(shell_xml("8 8 1", SHELL_GEOMS, option=EFM_OPTION,
                                 body='<edge equality="false" damping="1"/><contact selfcollide="auto"/>'
                                      '<elasticity young="3e4" poisson="0" thickness="1e-2" damping="1e-3" elastic2d="both"/>'))
    m = rb.MjModel.from_xml_path(str(xml))
    dm = K.DeviceModel(lib, m)
    b = K.Batch(dm, 1)
    d = rb.MjData(m)
    d.qvel[:] = np.random.default_rng(0).normal(0, .05, m.nv)
    for _ in range(5): rb.mj_step(m, d)
    s = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
    b.set("time", s[None, :1]); b.set("qpos", s[None, 1:1 + m.nq]); b.set("qvel", s[None, 1 + m.nq:1 + m.nq + m.nv])
    b.set("qacc_warmstart", d.qacc_warmstart[None, :])
    if m.nmocap: b.set("mocap_pos", d.mocap_pos.reshape(1, -1)); b.set("mocap_quat", d.mocap_quat.reshape(1, -1))
    rb.mj_forward(m, d); b.forward()
    assert d.efm_active == 1 and d.nefmK > 1000 and d.nefmdof > 0
    for name in ("efm_K_val", "efm_L", "efm_c"):
        ref = np.asarray(getattr(d, name)).ravel()
        assert np.array_equal(b.get(name)[0][:ref.size], ref), name
    assert _fields_exact(b, d, ["qfrc_smooth", "qacc_smooth", "qacc"]) == []
    maxcon, seen = _free_run(rb, lib, m, pre=40 if which == "solid" else 60, nstep=80, csr=1)
    assert maxcon > 5
    return maxcon


def test_implicit_effective_metric_solid_flex(rb, hostsim_lib, tmp_path):
    _effective_metric(rb, hostsim_lib, tmp_path, "solid")


def test_implicit_effective_metric_shell_with_bending(rb, hostsim_lib, tmp_path):
    assert _effective_metric(rb, hostsim_lib, tmp_path, "shell") > 40


def _bending_only_metric(rb, lib, tmp_path, equality):
    """A flex with bending stiffness ONLY under mj_flexCG: the reference assembles no CSR (nefmK = 0) -- K vec is the
    bending stencil operator (mjd_flexBend_mul, engine_derivative.c:1358) and the preconditioner solves the covered dofs with
    the constant sparse factor of mj_setConst (effBlockApply's flg_bend branch :3288, mju_cholSolveSparse).  The shift, the PCG
    for qacc_smooth and free-running steps on the geoms: fields / states / counts / CG iteration counts identical."""
    xml = tmp_path / "efm0.xml"
    xml.write_text(shell_xml("8 8 1", SHELL_GEOMS, option=EFM_OPTION,
                             body=f'<edge equality="{equality}" damping="1"/><contact selfcollide="none"/>'
                                  '<elasticity young="3e4" poisson="0" thickness="1e-2" damping="1e-3" elastic2d="bend"/>'))
    m = rb.MjModel.from_xml_path(str(xml))
    assert m.nefm0dof == m.nv and m.nefm0L > m.nv
    dm = K.DeviceModel(lib, m)
    assert dm.size("efm") == 2 and dm.size("ne0") == m.nv and dm.size("ne0lev1") > 1
    b = K.Batch(dm, 1)
    d = rb.MjData(m)
    d.qvel[:] = np.random.default_rng(0).normal(0, .05, m.nv)
    for _ in range(5): rb.mj_step(m, d)
    s = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
    b.set("time", s[None, :1]); b.set("qpos", s[None, 1:1 + m.nq]); b.set("qvel", s[None, 1 + m.nq:1 + m.nq + m.nv])
    b.set("qacc_warmstart", d.qacc_warmstart[None, :])
    rb.mj_forward(m, d); b.forward()
    assert d.efm_active == 1 and d.nefmK == 0 and d.nefmdof == 0
    ref = np.asarray(d.efm_c).ravel()
    assert np.array_equal(b.get("efm_c")[0][:ref.size], ref)
    assert _fields_exact(b, d, ["qfrc_smooth", "qacc_smooth", "qacc"]) == []
    maxcon, seen = _free_run(rb, lib, m, pre=60, nstep=60, csr=1)
    return maxcon


@pytest.mark.parametrize("equality", ["false", "true"])
def test_implicit_effective_metric_bending_only(rb, hostsim_lib, tmp_path, equality):
    assert _bending_only_metric(rb, hostsim_lib, tmp_path, equality) > 10


def _vertex_constraints(rb, lib, tmp_path, option, elastic):
    """mjEQ_FLEXVERT (edge equality "vert"; mj_flex engine_core_smooth.c:745-918, mj_instantiateEquality :1013): per vertex the
    two invariants of the Cauchy strain of its mass-weighted edge fan as residuals, their Jacobian rows over the dofs of the
    vertex and its neighbours.  flexvert_length / flexvert_J field by field, then free-running steps on the geoms."""
    xml = tmp_path / "vert.xml"
    xml.write_text(shell_xml("13 13 1", SHELL_GEOMS, option=option,
                             body=f'<edge equality="vert" damping="1"/><contact selfcollide="none"/>{elastic}'))
    m = rb.MjModel.from_xml_path(str(xml))
    assert m.nv > 128 and m.neq == 1 and m.eq_type[0] == 5
    dm = K.DeviceModel(lib, m)
    assert dm.size("csr") == 1
    b = K.Batch(dm, 1)
    d = rb.MjData(m)
    d.qvel[:] = np.random.default_rng(1).normal(0, .05, m.nv)
    for _ in range(5): rb.mj_step(m, d)
    s = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
    b.set("time", s[None, :1]); b.set("qpos", s[None, 1:1 + m.nq]); b.set("qvel", s[None, 1 + m.nq:1 + m.nq + m.nv])
    b.set("qacc_warmstart", d.qacc_warmstart[None, :])
    rb.mj_forward(m, d); b.forward()
    for name in ("flexvert_length", "flexvert_J"):
        ref = np.asarray(getattr(d, name)).ravel()
        assert ref.size and np.abs(ref).max() > 0
        assert np.array_equal(b.get(name)[0][:ref.size], ref), name
    assert d.ne == 2*m.nflexvert
    assert _fields_exact(b, d, ["qfrc_smooth", "qacc_smooth", "qacc"]) == []
    maxcon, seen = _free_run(rb, lib, m, pre=60, nstep=60, csr=1)
    return maxcon


def test_flex_vertex_equality_constraints(rb, hostsim_lib, tmp_path):
    assert _vertex_constraints(rb, hostsim_lib, tmp_path, 'solver="CG" tolerance="1e-6" timestep=".001" integrator="Euler"',
                               '<elasticity young="3e4" poisson="0" thickness="1e-2" elastic2d="bend"/>') > 10


def test_flex_vertex_equality_constraints_in_the_bending_metric(rb, hostsim_lib, tmp_path):
    assert _vertex_constraints(rb, hostsim_lib, tmp_path, EFM_OPTION,
                               '<elasticity young="3e4" poisson="0" thickness="1e-2" damping="1e-3" elastic2d="bend"/>') > 10


def _newton_beyond_128(rb, lib, tmp_path, which, cone):
    """Newton on the explicit-index rows (mjh_newtonx.h; MakeHessian / FactorizeHessian / HessianIncremental / HessianCone,
    engine_solver.c:2057-2340 on the reference's sparse matrices): H = J' D J + M, the reverse Cholesky factor in the
    order of mju_cholFactorSymbolic's tree walks, mju_cholSolveSparse, one mju_cholUpdateSparse per row that changes zone
    (per cone row with elliptic cones).  Free-running steps through contact: states, counts and Newton iteration counts."""
    xml = tmp_path / "newton.xml"
    option = f'solver="Newton" cone="{cone}" tolerance="1e-8" timestep=".001" integrator="Euler"'
    if which == "shell":
        xml.write_text(shell_xml("9 9 1", SHELL_GEOMS, option=option))
        pre, nstep = 60, 70
    elif which == "equality":
        xml.write_text(shell_xml("8 8 1", SHELL_GEOMS, option=option,
                                 body='<edge equality="true"/><contact selfcollide="none"/>'))
        pre, nstep = 60, 60
    elif which == "condim6":
        # the same scene with condim 6 everywhere (flex-geom and geom-geom contacts; rows 3..5 of a contact are its
        # torsional / rolling rows, on the contact's one column pattern): the cone Hessian's explicit-index branch
        xml.write_text(flex_xml("4 4 4", "0 0 .12", option=option,
                                flex_body='<edge damping="1"/><contact selfcollide="none" condim="6"/><elasticity young="5e4"/>',
                                extra_world='<geom type="sphere" size=".05" pos=".02 .01 .0" condim="6"/>'
                                            '<body pos=".5 0 .045"><freejoint/><geom type="box" size=".04 .04 .04" condim="6"/></body>'
                                            '<body pos="-.5 0 .3"><joint type="hinge" axis="0 1 0" range="-20 20"/><geom type="capsule" fromto="0 0 0 .2 0 0" size=".02" condim="4"/></body>'))
        pre, nstep = 100, 60
    else:
        # a solid flex on a sphere next to a free box and a hinged pendulum: several islands, one of them the flex
        xml.write_text(flex_xml("4 4 4", "0 0 .12", option=option,
                                extra_world='<geom type="sphere" size=".05" pos=".02 .01 .0"/>'
                                            '<body pos=".5 0 .045"><freejoint/><geom type="box" size=".04 .04 .04"/></body>'
                                            '<body pos="-.5 0 .3"><joint type="hinge" axis="0 1 0" range="-20 20"/><geom type="capsule" fromto="0 0 0 .2 0 0" size=".02"/></body>'))
        pre, nstep = 100, 60
    m = rb.MjModel.from_xml_path(str(xml))
    assert m.nv > 128
    dm = K.DeviceModel(lib, m)
    assert dm.size("csr") == 1 and dm.size("xn") == 1
    maxcon, seen = _free_run(rb, lib, m, pre=pre, nstep=nstep, csr=1)
    return maxcon


@pytest.mark.parametrize("which,cone", [("shell", "pyramidal"), ("shell", "elliptic"), ("equality", "pyramidal"),
                                        ("solid", "pyramidal"), ("solid", "elliptic"),
                                        ("condim6", "elliptic"), ("condim6", "pyramidal")])
def test_newton_beyond_128_dofs(rb, hostsim_lib, tmp_path, which, cone):
    assert _newton_beyond_128(rb, hostsim_lib, tmp_path, which, cone) > 3


def test_unsupported_flex_features_are_named(rb, hostsim_lib, tmp_path):
    xml = tmp_path / "eq.xml"
    xml.write_text(flex_xml("4 4 1", "0 0 1", flex_attr='dim="2"', flex_body='<edge equality="false"/><contact internal="true"/>'))
    m = rb.MjModel.from_xml_path(str(xml))
    with pytest.raises(K.MjhipError, match="flex internal collisions"):
        K.DeviceModel(hostsim_lib, m)


def test_multiwave_variant_is_default_for_flex_and_bit_identical(hostsim_lib):
    """Multi-wavefront workgroups (mjh_modes.h: wn + wq; MJH_MW wavefronts per environment): the default mapping of a
    flex model at launches of at most one workgroup per CU (BASELINE config 5).  Wave 0 runs the step, the tree / flex
    passes run workgroup-wide; the trajectory has to be bit-identical to the one-wavefront mapping's (the emulation
    runs 64 x MJH_MW fibers per environment; MJH_HOSTSIM_REVERSE=1 is the race detector)."""
    mm = K.MjbModel(hostsim_lib, os.path.join(GOLDEN, "jelly.mjb"))
    dm = K.DeviceModel(hostsim_lib, mm)
    nenv, T = 2, 10
    b = K.Batch(dm, nenv)
    assert b.kernel_variant() == "multiwave" and b.kernel_name().endswith("wn")
    b.reset()
    s0 = np.zeros((nenv, dm.nstate))
    s0[:, 1:1 + dm.nq] = b.get("qpos")[0]
    s0[:, 1 + dm.nq:1 + dm.nq + dm.nv] = np.random.default_rng(5).normal(0, 0.1, size=(nenv, dm.nv))
    ctrl = np.zeros((nenv, T, dm.nu))
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    g = K.Batch(dm, nenv)
    g.set_variant("generic")
    assert g.kernel_variant() == "generic"
    ref = g.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    assert np.array_equal(out, ref)
    # a model without flexes keeps its one-wavefront mapping
    hm = K.MjbModel(hostsim_lib, os.path.join(GOLDEN, "humanoid.mjb"))
    assert K.Batch(K.DeviceModel(hostsim_lib, hm), 2).kernel_variant() != "multiwave"


INTERP_XML = """
<mujoco>
  <option solver="CG" tolerance="1e-6" timestep=".001" integrator="Euler"/>
  <size memory="10M"/>
  <worldbody>
    <geom name="floor" type="plane" size="0 0 .05"/>
    <geom type="box" pos=".25 0 .05" size=".1 .3 .05"/>
    <geom type="box" pos="0 0 .04" size=".3 .3 .01" euler="0 12 0"/>
    <body mocap="true" pos="-.13 .03 .13" zaxis=".5 0 1"><geom type="capsule" size=".03 .05" condim="1"/></body>
    <flexcomp type="grid" count="COUNT" spacing=".05 .05 .05" pos="0 0 ZPOS" dim="3" radius=".002" mass="2" name="soft" dof="DOF">
      ELASTICITY
      <contact selfcollide="none" internal="false"/>
    </flexcomp>
  </worldbody>
</mujoco>"""


def _interpolated(rb, lib, tmp_path, dof, kind=None, count="5 5 5", pre=0, nstep=200):
    """an interpolated flex (model/flex/trilinear.xml / quadratic.xml's kind: 8 or 27 node bodies carry 125 vertices) falling
    onto a capsule, a tilted plate, a box and the floor: vertex interpolation (mj_flex), corotational cell elasticity with
    the rotation from mju_mat2Rot (mj_flexPassiveInterp), contact rows spread over the cell's nodes (mj_vertBodyWeight),
    more than fifty candidates against the two boxes of the world body (thinned in the order of mj_collideTree's walk
    over the body's and the flex's hierarchies).  trilinear: dense rows (24 dofs); quadratic: mask rows (81 dofs)."""
    xml = tmp_path / "interp.xml"
    tri = dof == "trilinear"
    xml.write_text(INTERP_XML.replace("COUNT", count).replace("DOF", dof).replace("ZPOS", ".17" if tri else ".21")
                   .replace("ELASTICITY", '<elasticity young="1e4" poisson="0.2" damping="0.003"/>' if tri else
                            '<elasticity young="3e3" poisson="0.1" damping="0.0005"/>'))
    m = rb.MjModel.from_xml_path(str(xml), kind=kind)
    assert m.nv == (24 if dof == "trilinear" else 81)
    dm = K.DeviceModel(lib, m)
    assert dm.size("csr") == 0 and dm.size("sparse") == (0 if dof == "trilinear" else 1)
    maxcon, seen = _free_run(rb, lib, m, pre, nstep)
    return maxcon, seen


def test_trilinear_flex_free_run(rb, hostsim_lib, tmp_path):
    maxcon, seen = _interpolated(rb, hostsim_lib, tmp_path, "trilinear")
    assert maxcon == 50 and seen >= {1, 2, 3}, (maxcon, seen)


def test_quadratic_flex_free_run(rb, hostsim_lib, tmp_path):
    maxcon, seen = _interpolated(rb, hostsim_lib, tmp_path, "quadratic", pre=100, nstep=120)
    assert maxcon == 50 and seen >= {1, 2}, (maxcon, seen)


FLEXFLEX_XML = """
<mujoco>
  <option solver="CG" tolerance="1e-6" timestep=".001" integrator="Euler"/>
  <size memory="10M"/>
  <worldbody>
    <geom name="floor" type="plane" size="0 0 .05"/>
    <flexcomp type="grid" count="4 4 3" spacing=".05 .05 .05" pos="0 0 .07" dim="3" radius=".005" mass="1" name="a"DOF>
      BODY
    </flexcomp>
    <flexcomp type="grid" count="3 3 3" spacing=".05 .05 .05" pos=".03 .02 .24" dim="3" radius=".005" mass="1" name="b"DOF>
      BODY
    </flexcomp>
  </worldbody>
</mujoco>"""


def _flex_on_flex(rb, lib, tmp_path, dof, nstep=100, kind=None):
    """one flex dropped onto another (mj_collideElems over the element pairs whose boxes overlap, GJK / EPA between two
    tetrahedra; flat faces meet, so more than fifty contacts are thinned in the order of the walk over the two flexes'
    hierarchies, then sorted by (element, element)); rows carry the four corners of both elements (standard flexes, 225
    dofs: explicit-index rows) or the eight nodes of both cells (trilinear, 48 dofs: dense rows)"""
    xml = tmp_path / "ff.xml"
    body = ('<edge damping="1"/><contact selfcollide="none"/><elasticity young="5e4"/>' if not dof else
            '<contact selfcollide="none"/><elasticity young="1e4" damping="0.003"/>')
    xml.write_text(FLEXFLEX_XML.replace("DOF", f' dof="{dof}"' if dof else "").replace("BODY", body))
    m = rb.MjModel.from_xml_path(str(xml), kind=kind)
    d = rb.MjData(m)
    first = None
    for t in range(300):
        rb.mj_step(m, d)
        if d.ncon and (np.asarray(d.contact[:d.ncon]["flex"])[:, 0] >= 0).any(): first = t; break
    assert first is not None
    dm = K.DeviceModel(lib, m)
    assert dm.size("csr") == (0 if dof else 1)
    b = K.Batch(dm, 1)
    rb.mj_resetData(m, d)
    for _ in range(first - 5): rb.mj_step(m, d)
    s = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
    b.set("time", s[None, :1]); b.set("qpos", s[None, 1:1 + m.nq]); b.set("qvel", s[None, 1 + m.nq:]); b.set("qacc_warmstart", d.qacc_warmstart[None, :])
    most = 0
    for t in range(nstep):
        b.step(); rb.mj_step(m, d)
        c = b.get("counts")[0]
        assert (c[0], c[1], c[5]) == (d.ncon, d.nefc, d.solver_niter[0]), (t, c[:6], d.ncon, d.nefc, d.solver_niter[0])
        assert not b.get("warning")[0].any()
        assert np.array_equal(b.get("qpos")[0], d.qpos) and np.array_equal(b.get("qvel")[0], d.qvel), t
        if d.ncon:
            con = d.contact[:d.ncon]
            ff = np.asarray(con["flex"])[:, 0] >= 0
            most = max(most, int(ff.sum()))
            cf = b.get("con_flex")[0][:6*d.ncon].reshape(-1, 6)
            assert np.array_equal(cf[:, [3, 0, 4, 1]], np.c_[np.asarray(con["flex"]), np.asarray(con["elem"])]), t
    return most


def test_flex_on_flex_standard(rb, hostsim_lib, tmp_path):
    assert _flex_on_flex(rb, hostsim_lib, tmp_path, "") == 50


def test_flex_on_flex_trilinear(rb, hostsim_lib, tmp_path):
    assert _flex_on_flex(rb, hostsim_lib, tmp_path, "trilinear") == 50


REFERENCE_INTERP = {"trilinear": (330, 120, 24), "quadratic": (990, 60, 81), "sphere_trilinear": (1340, 60, 48)}


def _reference_interp_model(rb, lib, name, kind=None, nstep=None):
    """model/flex/{trilinear, quadratic, sphere_trilinear}.xml as shipped (tests/golden/*.mjb, tools/make_golden.py): 512 / 594
    interpolated vertices on 8 / 27 / 2 x 8 node bodies.  The oracle runs `pre` steps from reset (free fall, first contacts),
    the kernels take over its state and both run on: the capsule under the trilinear / quadratic block; the two spheres on the
    tilted plate and the box, where more than fifty contacts of the world body's two boxes are thinned in walk order."""
    pre, n, nv = REFERENCE_INTERP[name]
    m = rb.MjModel.from_binary_path(os.path.join(GOLDEN, name + ".mjb"), kind=kind)
    assert m.nv == nv
    maxcon, seen = _free_run(rb, lib, m, pre, nstep or n)
    assert maxcon > 0
    return maxcon


@pytest.mark.parametrize("name", sorted(REFERENCE_INTERP))
def test_reference_interpolated_flex_models(rb, hostsim_lib, name):
    maxcon = _reference_interp_model(rb, hostsim_lib, name)
    if name == "sphere_trilinear": assert maxcon > 50
