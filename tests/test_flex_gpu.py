"""Flexes on the device: the cases of tests/test_flex_hostsim.py through the HIP library (free motion against the reference
linked with the kernels' libm routines is not needed here: a flex model's kinematics only adds and multiplies)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
import test_flex_hostsim as fh

pytestmark = pytest.mark.gpu


def test_jelly_free_motion_bit_exact_on_gpu(rb, hip_lib):
    m = rb.MjModel.from_binary_path(os.path.join(GOLDEN, "jelly.mjb"))
    fh._free_motion(rb, hip_lib, m)


def test_shell_bending_on_gpu(rb, hip_lib, tmp_path):
    xml = tmp_path / "shell.xml"
    xml.write_text(fh.flex_xml("6 6 1", "0 0 1", flex_attr='dim="2"',
                               flex_body='<edge equality="false" damping="10"/><contact contype="0" conaffinity="0"/>'
                                         '<elasticity young="3e5" poisson="0" thickness="1e-2" damping="1e-3" elastic2d="both"/><pin id="0 5 30 35"/>'))
    fh._free_motion(rb, hip_lib, rb.MjModel.from_xml_path(str(xml)))


def test_flex_contact_lists_exact_on_gpu(rb, hip_lib, tmp_path):
    fh._contact_lists(rb, hip_lib, tmp_path)


def test_flex_on_floor_keeps_fifty_contacts_on_gpu(rb, hip_lib, tmp_path):
    xml = tmp_path / "floor.xml"
    xml.write_text(fh.flex_xml("9 9 2", "0 0 .035"))
    maxcon, kinds = fh._resync_steps(rb, hip_lib, rb.MjModel.from_xml_path(str(xml)), pre=30, nstep=40)
    assert maxcon == 50 and kinds == {"vert"}


def test_jelly_on_the_capsule_on_gpu(rb, hip_lib):
    m = rb.MjModel.from_binary_path(os.path.join(GOLDEN, "jelly.mjb"))
    maxcon, kinds = fh._resync_steps(rb, hip_lib, m, pre=385, nstep=30)
    assert maxcon >= 20 and "elem" in kinds


def test_flex_against_an_actuated_body_on_gpu(rb, hip_lib, tmp_path):
    fh._actuated_body(rb, hip_lib, tmp_path)


def test_flex_island_next_to_rigid_islands_on_gpu(rb, hip_lib, tmp_path):
    fh._multi_island(rb, hip_lib, tmp_path)

def test_two_flexes_are_two_islands_of_sliders_on_gpu(rb, hip_lib, tmp_path):
    fh._two_flexes(rb, hip_lib, tmp_path)


def test_flex_edge_equality_constraints_on_gpu(rb, hip_lib, tmp_path):
    """mjEQ_FLEX rows (one per non-rigid edge) in front of the contact rows: bit-exact incl. CG iteration counts"""
    maxcon, kinds = fh._edge_equality(rb, hip_lib, tmp_path, nstep=60)
    assert maxcon > 0


def test_shell_flex_on_sphere_box_capsule_cylinder_on_gpu(rb, hip_lib, tmp_path):
    """triangle elements against sphere / box / capsule (closed forms) and cylinder (GJK / EPA + fixNormal), explicit-index rows"""
    assert fh._shell_on_geoms(rb, hip_lib, tmp_path) > 40


def test_shell_flex_on_geoms_dense_rows_on_gpu(rb, hip_lib, tmp_path):
    fh._shell_on_geoms(rb, hip_lib, tmp_path, count="4 4 1", csr=0, geoms=fh.SHELL_GEOMS.replace(".06 .06 .15", ".02 .02 .15"))


def test_shell_flex_on_ellipsoids_on_gpu(rb, hip_lib, tmp_path):
    """triangle elements against ellipsoids: GJK / EPA + the ellipsoid case of mjc_fixNormal (round 6)"""
    assert fh._shell_on_ellipsoids(rb, hip_lib, tmp_path) > 10


def test_line_flex_on_cylinder_and_ellipsoid_on_gpu(rb, hip_lib, tmp_path):
    fh._line_on_cylinder(rb, hip_lib, tmp_path)


def test_shell_flex_self_collision_on_gpu(rb, hip_lib, tmp_path):
    """flex : flex contacts of a cloth folding over a bar (sweep-and-prune order, two-sided weighted rows), bit for bit"""
    assert fh._self_collision(rb, hip_lib, tmp_path, "auto", nstep=100) == 50
    assert fh._self_collision(rb, hip_lib, tmp_path, "narrow", nstep=60) > 10
    assert fh._self_collision(rb, hip_lib, tmp_path, "bvh", nstep=100) == 50        # (the hierarchy against itself: walk order)


def test_flex_vertices_on_articulated_bodies_on_gpu(rb, hip_lib, tmp_path):
    """radial sliders under a free body (softbox.xml's kind) and a cloth riding on a hinged pole"""
    fh._articulated_vertices(rb, hip_lib, tmp_path, "radial")
    fh._articulated_vertices(rb, hip_lib, tmp_path, "hinge")


@pytest.mark.parametrize("solver,equality", [("Newton", True), ("CG", False)])
def test_flex_on_mask_rows_on_gpu(rb, hip_lib, tmp_path, solver, equality):
    fh._flex_on_mask_path(rb, hip_lib, tmp_path, solver, equality)


def test_connect_and_weld_rows_next_to_flex_edge_constraints_on_gpu(rb, hip_lib, tmp_path):
    fh._hanging_cloth(rb, hip_lib, tmp_path)


def test_line_flex_capsule_colliders_on_gpu(rb, hip_lib, tmp_path):
    """against the reference linked with the kernels' atan2 / sin / cos (the capsule frames call them): bit for bit"""
    assert 2 in fh._cable(rb, hip_lib, tmp_path, 30, 150, kind="devmath")
    fh._cable(rb, hip_lib, tmp_path, 50, 110, kind="devmath")


@pytest.mark.parametrize("which", ["solid", "shell"])
def test_implicit_effective_metric_on_gpu(rb, hip_lib, tmp_path, which):
    """mj_flexCG: K assembly, block factors, shift, PCG for qacc_smooth and the CG solve in the metric M + K, bit for bit"""
    fh._effective_metric(rb, hip_lib, tmp_path, which)


@pytest.mark.parametrize("equality", ["false", "true"])
def test_implicit_effective_metric_bending_only_on_gpu(rb, hip_lib, tmp_path, equality):
    """bending stiffness only: the stencil operator and the level-scheduled solve with mj_setConst's constant factor"""
    assert fh._bending_only_metric(rb, hip_lib, tmp_path, equality) > 10


def test_flex_vertex_equality_constraints_on_gpu(rb, hip_lib, tmp_path):
    """mjEQ_FLEXVERT: flexvert_length / flexvert_J field by field, then through contact -- explicit and in the bending metric"""
    assert fh._vertex_constraints(rb, hip_lib, tmp_path, 'solver="CG" tolerance="1e-6" timestep=".001" integrator="Euler"',
                                  '<elasticity young="3e4" poisson="0" thickness="1e-2" elastic2d="bend"/>') > 10
    assert fh._vertex_constraints(rb, hip_lib, tmp_path, fh.EFM_OPTION,
                                  '<elasticity young="3e4" poisson="0" thickness="1e-2" damping="1e-3" elastic2d="bend"/>') > 10


@pytest.mark.parametrize("which,cone", [("shell", "pyramidal"), ("shell", "elliptic"), ("equality", "pyramidal"), ("solid", "elliptic"),
                                        ("condim6", "elliptic")])
def test_newton_beyond_128_dofs_on_gpu(rb, hip_lib, tmp_path, which, cone):
    """Newton on the explicit-index rows (mjh_newtonx.h): factorisation in the reference's visiting order, sparse solves,
    rank-one updates, cone Hessians; states, counts and Newton iteration counts identical through contact"""
    assert fh._newton_beyond_128(rb, hip_lib, tmp_path, which, cone) > 3


@pytest.mark.parametrize("dof", ["trilinear", "quadratic"])
def test_interpolated_flex_on_gpu(rb, hip_lib, tmp_path, dof):
    """node bodies, interpolated vertices, corotational cells (mju_mat2Rot calls sin / cos: against the reference linked with
    the kernels' routines), node-weighted contact rows, thinning in walk order -- bit for bit"""
    if dof == "trilinear": assert fh._interpolated(rb, hip_lib, tmp_path, dof, kind="devmath")[0] == 50
    else: assert fh._interpolated(rb, hip_lib, tmp_path, dof, kind="devmath", pre=100, nstep=120)[0] == 50


@pytest.mark.parametrize("name", sorted(fh.REFERENCE_INTERP))
def test_reference_interpolated_flex_models_on_gpu(rb, hip_lib, name):
    """the reference's own interpolated flex models, several hundred steps through their contact phase, bit for bit against
    the reference linked with the kernels' sin / cos"""
    maxcon = fh._reference_interp_model(rb, hip_lib, name, kind="devmath", nstep=400)
    if name == "sphere_trilinear": assert maxcon > 50


def test_flex_on_flex_on_gpu(rb, hip_lib, tmp_path):
    """element : element contacts between two flexes, more than fifty thinned in the order of the walk over both hierarchies"""
    assert fh._flex_on_flex(rb, hip_lib, tmp_path, "") == 50
    assert fh._flex_on_flex(rb, hip_lib, tmp_path, "trilinear", kind="devmath") == 50


def test_jelly_batch_of_64_on_gpu(rb, hip_lib):
    """a BATCH of flex environments (64 x jelly.xml with different vertex velocities, 400 steps from the reset state: the
    fall and the first ~60 steps on the capsule): two of the environments bit for bit, every step, against the reference
    stepped from the same state; all environments distinct; the launch is at most one workgroup per CU, so this runs
    the multi-wavefront mapping (8 wavefronts per environment)"""
    from mujoco_amd import _capi as K
    m = rb.MjModel.from_binary_path(os.path.join(GOLDEN, "jelly.mjb"))
    dm = K.DeviceModel(hip_lib, m)
    nenv, T = 64, 400
    b = K.Batch(dm, nenv)
    assert b.kernel_variant() == "multiwave"
    b.reset()
    rng = np.random.default_rng(11)
    s0 = np.zeros((nenv, dm.nstate))
    s0[:, 1:1 + dm.nq] = b.get("qpos")[0]
    s0[:, 1 + dm.nq:1 + dm.nq + dm.nv] = rng.normal(0, 0.1, size=(nenv, dm.nv))
    ctrl = np.zeros((nenv, T, dm.nu))
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    assert b.get("warning").sum() == 0
    assert len({out[e, -1].tobytes() for e in range(nenv)}) == nenv
    d = rb.MjData(m)
    spec = rb.mjSTATE_FULLPHYSICS
    ncon_seen = 0
    for e in (0, 63):
        rb.mj_resetData(m, d)
        rb.mj_setState(m, d, s0[e], spec)
        for t in range(T):
            rb.mj_step(m, d)
            assert np.array_equal(out[e, t], rb.mj_getState(m, d, spec)), (e, t)
        ncon_seen = max(ncon_seen, int(d.ncon))
    print("jelly batch: contacts at the end of the sampled rollouts up to", ncon_seen)
    assert ncon_seen > 0


def test_trilinear_batch_of_128_on_gpu(rb, hip_lib):
    """a BATCH of interpolated-flex environments (128 x trilinear.xml, different node velocities, 450 steps from the reset state:
    the fall and the first ~100 steps on the capsule): three of the environments bit for bit, every step, against the
    reference (linked with the kernels' sin / cos) stepped from the same state; all environments distinct"""
    from mujoco_amd import _capi as K
    m = rb.MjModel.from_binary_path(os.path.join(GOLDEN, "trilinear.mjb"), kind="devmath")
    dm = K.DeviceModel(hip_lib, m)
    nenv, T = 128, 450
    b = K.Batch(dm, nenv)
    b.reset()
    rng = np.random.default_rng(5)
    s0 = np.zeros((nenv, dm.nstate))
    s0[:, 1:1 + dm.nq] = b.get("qpos")[0]
    s0[:, 1 + dm.nq:1 + dm.nq + dm.nv] = rng.normal(0, 0.05, size=(nenv, dm.nv))
    d = rb.MjData(m)
    mocap = (np.asarray(d.mocap_pos).reshape(1, -1), np.asarray(d.mocap_quat).reshape(1, -1))
    b.set("mocap_pos", np.repeat(mocap[0], nenv, 0)); b.set("mocap_quat", np.repeat(mocap[1], nenv, 0))
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, np.zeros((nenv, T, dm.nu)))
    assert b.get("warning").sum() == 0
    assert len({out[e, -1].tobytes() for e in range(nenv)}) == nenv
    spec = rb.mjSTATE_FULLPHYSICS
    ncon_seen = 0
    for e in (0, 77, 127):
        rb.mj_resetData(m, d)
        rb.mj_setState(m, d, s0[e], spec)
        for t in range(T):
            rb.mj_step(m, d)
            assert np.array_equal(out[e, t], rb.mj_getState(m, d, spec)), (e, t)
            ncon_seen = max(ncon_seen, int(d.ncon))
    assert ncon_seen > 0
