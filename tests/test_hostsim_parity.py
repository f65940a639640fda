"""CPU: the kernel sources, compiled for the host wavefront emulation (tests/hostsim), against the
oracle.  Same C ABI, same kernels as the GPU build; only the lane scheduler differs.  Bit-exact
agreement is asserted (tol=0): the kernels follow the reference's operation order."""
import os
import subprocess
import sys

import numpy as np
import pytest

from mujoco_amd import _capi as K
from conftest import GOLDEN, HOSTSIM_LIB, ROOT, contact_rich_states, humanoid_pgs_oracle, many_constraint_states
from parity_utils import CYL_XML, EQ_XML, IMPL_XML, CONDIM_XML, ACT_XML, SENSOR_XML, BOX_XML, BOXBOX_XML, CAPBOX_XML, MOCAP_XML, PAIR_XML, FLUID_XML, ELLIPSOID_FLUID_XML, CAMERA_XML, ISLANDS_XML, TENDON_XML, WRAP_XML, ACT_GROUP_XML, MUSCLE_XML, SITE_ACT_XML, BALL_ACT_XML, SURFACEVEL_XML, ADHESION_XML, condim_scene_state, chain_xml, many_spheres_xml, check_forward, oracle_rollout, relerr


@pytest.fixture(scope="module")
def setup(rb, hostsim_lib):
    m = humanoid_pgs_oracle(rb)
    dm = K.DeviceModel(hostsim_lib, m)
    return m, dm


def test_static_tables(setup):
    m, dm = setup
    assert dm.nq == 28 and dm.nv == 27 and dm.nstate == 56
    assert dm.npair > 100          # static candidate pairs
    assert dm.nlevel == 7


def test_forward_bit_exact(rb, setup):
    m, dm = setup
    states = contact_rich_states(rb, m, 6, seed=3)
    b = K.Batch(dm, len(states))
    worst = check_forward(rb, m, b, states, tol=0.0)
    assert worst == 0.0


def test_pgs_two_constraints_per_lane_bit_exact(rb, setup):
    """64 < nefc <= 128: the register-resident PGS with two constraints per lane (solve_pgs_wide) --
    every field, force, state and the iteration count against the oracle, bit for bit; and a short
    rollout through such states"""
    m, dm = setup
    states, nefcs = many_constraint_states(rb, m, 8)
    assert len(states) == 8 and min(nefcs) > 64 and max(nefcs) > 96
    b = K.Batch(dm, len(states))
    assert check_forward(rb, m, b, states, tol=0.0) == 0.0
    assert np.array_equal(b.get("counts")[:, 1], nefcs)
    s0 = np.stack([np.concatenate([[s["time"]], s["qpos"], s["qvel"]]) for s in states])
    ws = np.stack([s["qacc_warmstart"] for s in states])
    ctrl = np.repeat(np.stack([s["ctrl"] for s in states])[:, None], 6, axis=1)
    ref, ints = oracle_rollout(rb, m, s0, ctrl, ws)
    out = b.rollout_host(6, K.mjSTATE_CTRL, s0, ws, ctrl)
    assert np.array_equal(out, ref)
    c = b.get("counts")
    assert np.array_equal(c[:, 0], ints[:, -1, 0]) and np.array_equal(c[:, 1], ints[:, -1, 1]) and np.array_equal(c[:, 5], ints[:, -1, 2])


@pytest.mark.parametrize("budget", [4096, 10240, 20480, 65536])
def test_forward_bit_exact_lds_resident(rb, setup, budget):
    """the same stages run on the LDS residency plan (what step/rollout use), every field written
    back after each stage: bit-exact again, for plans from 'almost nothing fits' to 'everything fits'"""
    m, dm = setup
    states = contact_rich_states(rb, m, 4, seed=7)
    b = K.Batch(dm, len(states))
    left = b.plan_lds(budget)
    assert left >= 0 and "LDS plan" in b.lds_report()
    worst = check_forward(rb, m, b, states, tol=0.0, lds=True)
    assert worst == 0.0


@pytest.mark.parametrize("budget", [0, 6144, 12288, 20480])
def test_rollout_bit_exact_any_lds_budget(setup, golden, budget):
    m, dm = setup
    fx = golden("humanoid")
    b = K.Batch(dm, 3)
    b.plan_lds(budget)
    out = b.rollout_host(12, K.mjSTATE_CTRL, fx["state0"][3:6], None, fx["ctrl"][3:6, :12])
    assert np.array_equal(out, fx["state"][3:6, :12])
    # closed-loop stepping continues from the state the rollout kernel exported
    b2 = K.Batch(dm, 3)
    b2.plan_lds(budget)
    b2.set("time", fx["state0"][3:6, :1]); b2.set("qpos", fx["state0"][3:6, 1:29]); b2.set("qvel", fx["state0"][3:6, 29:])
    for t in range(4):
        b2.set("ctrl", fx["ctrl"][3:6, t])
        b2.step(1)
    assert np.array_equal(b2.get("qpos"), fx["state"][3:6, 3, 1:29])
    assert np.array_equal(b2.get("qvel"), fx["state"][3:6, 3, 29:])


@pytest.mark.parametrize("lds", [False, True])
def test_soa_pipeline_forward_bit_exact(rb, setup, lds):
    """SoA-across-environments batch: lane-per-env smooth kernels + wave-per-env constraint kernel
    (on its own LDS plan when lds=True) + lane-per-env tail, every field against the oracle"""
    m, dm = setup
    states = contact_rich_states(rb, m, 3, seed=9)
    b = K.Batch(dm, len(states), layout="soa")
    worst = check_forward(rb, m, b, states, tol=0.0, lds=lds)
    assert worst == 0.0


def test_soa_pipeline_rollout_bit_exact(setup, golden):
    m, dm = setup
    fx = golden("humanoid")
    b = K.Batch(dm, 3, layout="soa")
    out = b.rollout_host(15, K.mjSTATE_CTRL, fx["state0"][2:5], None, fx["ctrl"][2:5, :15])
    assert np.array_equal(out, fx["state"][2:5, :15])
    b.plan_lds(0)
    out = b.rollout_host(6, K.mjSTATE_CTRL, fx["state0"][2:5], None, fx["ctrl"][2:5, :6])
    assert np.array_equal(out, fx["state"][2:5, :6])


@pytest.mark.parametrize("layout", ["aos", "soa"])
def test_rk4_rollout_bit_exact(rb, hostsim_lib, golden, layout):
    """mj_RungeKutta(N=4) (engine_forward.c:1486-1587): four forward evaluations per step, positions
    combined on the manifold -- against the oracle stepping with opt.integrator = mjINT_RK4"""
    m = humanoid_pgs_oracle(rb)
    m.opt.integrator = 1
    dm = K.DeviceModel(hostsim_lib, m)
    fx = golden("humanoid")
    n, T = 2, 6
    s0, ctrl = fx["state0"][4:4 + n], fx["ctrl"][4:4 + n, :T]
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dm, n, layout=layout)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    assert np.array_equal(out, ref)
    assert np.array_equal(b.get("counts")[:, 1], ints[:, -1, 1])      # nefc of the last evaluation


@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("cone", [0, 1])
def test_newton_solver_bit_exact(rb, hostsim_lib, golden, layout, cone):
    """the reference's DEFAULT solver (mjSOL_NEWTON, engine_solver.c:2344-2563) on humanoid, pyramidal and
    elliptic cones: an operation-for-operation restatement (mjh_newton.h: MakeHessian once, rank-one
    Cholesky updates per state change, the reference's summation orders), so whole trajectories and the
    Newton iteration count of EVERY step equal the oracle's bit for bit"""
    m = rb.MjModel.from_binary_path(os.path.join(GOLDEN, "humanoid.mjb"))
    assert m.opt.solver == 2
    m.opt.cone = cone
    dm = K.DeviceModel(hostsim_lib, m)
    fx = golden("humanoid")
    n, T = 4, 60
    s0, ctrl = fx["state0"][:n], fx["ctrl"][:n, :T]
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dm, n, layout=layout)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    assert np.array_equal(out, ref)
    c = b.get("counts")
    assert np.array_equal(c[:, 0], ints[:, -1, 0]) and np.array_equal(c[:, 1], ints[:, -1, 1])
    assert np.array_equal(c[:, 5], ints[:, -1, 2])                # Newton iterations of the last step
    assert ints[..., 2].max() >= 4
    # every step's iteration count: single steps from the oracle's own (state, warm start)
    d = rb.MjData(m)
    b1 = K.Batch(dm, 1, layout=layout)
    rb.mj_setState(m, d, s0[1], rb.mjSTATE_FULLPHYSICS)
    for t in range(40):
        st = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
        ws = np.array(d.qacc_warmstart)[None]
        d.ctrl[:] = ctrl[1, t]
        rb.mj_step(m, d)
        o = b1.rollout_host(1, K.mjSTATE_CTRL, st, ws, ctrl[1:2, t:t + 1])
        assert b1.get("counts")[0, 5] == d.solver_niter[0], t
        assert np.array_equal(o[0, 0], rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)), t
    # one forward pass from contact-rich keyframes: constraint forces and accelerations, exactly
    states = contact_rich_states(rb, m, 4, seed=2)
    b2 = K.Batch(dm, len(states), layout=layout)
    from parity_utils import load_states
    load_states(b2, states)
    b2.forward()
    for e, st in enumerate(states):
        d.qpos[:] = st["qpos"]; d.qvel[:] = st["qvel"]; d.qacc_warmstart[:] = st["qacc_warmstart"]; d.ctrl[:] = st["ctrl"]
        rb.mj_forward(m, d)
        assert b2.get("counts")[e, 1] == d.nefc
        for f in ["qacc", "qfrc_constraint"]:
            assert np.array_equal(b2.get(f)[e], np.asarray(getattr(d, f))), (e, f)
        if d.nefc:
            assert np.array_equal(b2.get("efc_force")[e][:d.nefc], np.array(d.efc_force))


@pytest.mark.parametrize("cone", [0, 1])
def test_cg_solver_bit_exact(rb, hostsim_lib, golden, cone):
    """mjSOL_CG (mj_solPrimal without the Hessian: M^-1-preconditioned gradient, Hager-Zhang direction,
    engine_solver.c:2489-2521): the same restatement, bit-exact over a rollout including the (up to 100)
    iterations of every step"""
    m = rb.MjModel.from_binary_path(os.path.join(GOLDEN, "humanoid.mjb"))
    m.opt.solver = 1
    m.opt.cone = cone
    dm = K.DeviceModel(hostsim_lib, m)
    fx = golden("humanoid")
    n, T = 3, 40
    s0, ctrl = fx["state0"][:n], fx["ctrl"][:n, :T]
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dm, n)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    assert np.array_equal(out, ref)
    assert np.array_equal(b.get("counts")[:, 5], ints[:, -1, 2])


def test_pgs_residual_mode_tolerance_parity(rb, setup, golden):
    """opt-in residual-update PGS sweep (include/mjhip.h: mjhip_batch_set_pgs_mode): tolerance parity -- forces and
    accelerations to rounding, every next state within 1e-6, contact / constraint counts exact; the default mode of the
    same batch stays bit-exact"""
    from parity_utils import pgs_residual_parity
    m, dm = setup
    states = contact_rich_states(rb, m, 6, seed=11)
    worst_f, worst_q, worst_s, dn, nmax = pgs_residual_parity(rb, K, m, dm, states, T=4)
    print("pgs residual: force", worst_f, "qacc", worst_q, "state", worst_s, "max |delta niter|", dn, "max nefc", nmax)
    assert nmax > 8
    assert worst_s <= 1e-6 and worst_q <= 1e-6
    fx = golden("humanoid")
    n, T = 3, 12
    b = K.Batch(dm, n)
    b.set_pgs_mode(1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, fx["state0"][:n], None, fx["ctrl"][:n, :T])
    assert not np.array_equal(out, fx["state"][:n, :T]) or True      # (rounding differs; it may still coincide)
    assert np.max(np.abs(out[:, :6] - fx["state"][:n, :6]) / np.maximum(1.0, np.abs(fx["state"][:n, :6]))) <= 1e-6
    b.set_pgs_mode(0)
    assert np.array_equal(b.rollout_host(T, K.mjSTATE_CTRL, fx["state0"][:n], None, fx["ctrl"][:n, :T]), fx["state"][:n, :T])


def test_pgs_residual_mode_two_constraints_per_lane(rb, setup):
    """the residual-update sweep beyond 64 rows (solve_pgs_resid_wide: the settled humanoid's stragglers): tolerance parity"""
    from parity_utils import pgs_residual_parity
    m, dm = setup
    states, nefcs = many_constraint_states(rb, m, 6)
    assert min(nefcs) > 64
    worst_f, worst_q, worst_s, dn, nmax = pgs_residual_parity(rb, K, m, dm, states, T=3)
    print("pgs residual wide: force", worst_f, "qacc", worst_q, "state", worst_s, "max |delta niter|", dn, "max nefc", nmax)
    assert nmax > 64 and worst_s <= 1e-6 and worst_q <= 1e-6


def test_generic_pgs_path_bit_exact(rb, hostsim_lib, golden):
    """opt.iterations above the precomputed visitation-order table (128) takes the generic PGS sweep
    (LDS/HBM-resident iterate, in-kernel PCG32 shuffle) instead of the register-resident one"""
    m = humanoid_pgs_oracle(rb)
    m.opt.iterations = 150
    dm = K.DeviceModel(hostsim_lib, m)
    fx = golden("humanoid")
    n, T = 3, 8
    s0, ctrl = fx["state0"][1:1 + n], fx["ctrl"][1:1 + n, :T]
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dm, n)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    assert np.array_equal(out, ref)
    assert np.array_equal(b.get("counts")[:, 5], ints[:, -1, 2])


def test_rollout_bit_exact_vs_golden(setup, golden):
    m, dm = setup
    fx = golden("humanoid")
    n, T = 4, 25
    b = K.Batch(dm, n)
    out = b.rollout_host(T, K.mjSTATE_CTRL, fx["state0"][:n], None, fx["ctrl"][:n, :T])
    assert np.array_equal(out, fx["state"][:n, :T])


def test_lane_order_independence(golden):
    """race detector: the emulation run with lanes scheduled 63..0 gives identical bits"""
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "from mujoco_amd import _capi as K\n"
        "lib = K.Lib(%r)\n"
        "m = K.MjbModel(lib, %r); m.set_option('solver', 0)\n"
        "dm = K.DeviceModel(lib, m)\n"
        "fx = np.load(%r)\n"
        "b = K.Batch(dm, 2)\n"
        "out = b.rollout_host(12, K.mjSTATE_CTRL, fx['state0'][1:3], None, fx['ctrl'][1:3, :12])\n"
        "assert np.array_equal(out, fx['state'][1:3, :12]), 'reverse lane order changed the result'\n"
    ) % (ROOT, HOSTSIM_LIB, os.path.join(GOLDEN, "humanoid.mjb"), os.path.join(GOLDEN, "humanoid_traj.npz"))
    env = dict(os.environ, MJH_HOSTSIM_REVERSE="1")
    subprocess.run([sys.executable, "-c", code], check=True, env=env)


def test_warning_freezes_trajectory(rb, setup):
    """rollout.cc:135-155: after a warning the rest of the trajectory is back-filled"""
    m, dm = setup
    d = rb.MjData(m)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    s0[0, 1 + 28 + 3] = 1e12     # bad qvel -> mjWARN_BADQVEL, auto reset
    b = K.Batch(dm, 1)
    out = b.rollout_host(5, K.mjSTATE_CTRL, s0, None, np.zeros((1, 5, 21)))
    ref, _ = oracle_rollout(rb, m, s0, np.zeros((1, 1, 21)))
    assert np.array_equal(out[0, 0], ref[0, 0])
    for t in range(1, 5):
        assert np.array_equal(out[0, t], out[0, 0])
    assert b.get("warning")[0, 4] == 1


def test_capacity_overflow_raises_warning(rb, hostsim_lib, golden):
    m = humanoid_pgs_oracle(rb)
    dm = K.DeviceModel(hostsim_lib, m, 2, 8)     # absurdly small capacities
    d = rb.MjData(m)
    rb.mj_resetDataKeyframe(m, d, 2)
    b = K.Batch(dm, 1)
    b.set("qpos", np.array(d.qpos)[None])
    b.forward()
    w = b.get("warning")[0]
    assert w[1] >= 1 or w[2] >= 1    # CONTACTFULL or CNSTRFULL


def test_unsupported_models_are_rejected(rb, hostsim_lib):
    m = rb.MjModel.from_binary_path(os.path.join(GOLDEN, "humanoid.mjb"))
    m.opt.noslip_iterations = 3            # accepted on the dense constraint path since round 6 ...
    K.DeviceModel(hostsim_lib, m)
    m.opt.jacobian = 1                     # ... not on the sparse one (mjJAC_SPARSE)
    with pytest.raises(K.MjhipError, match="noslip"):
        K.DeviceModel(hostsim_lib, m)
    sc = rb.MjModel.from_binary_path(os.path.join(GOLDEN, "slider_crank.mjb"))
    sc.opt.solver = 0
    sc.opt.integrator = 2      # mjINT_IMPLICIT: accepted since round 6
    K.DeviceModel(hostsim_lib, sc)
    sc.opt.enableflags |= 1 << 4      # mjENBL_SLEEP (out of scope: SURVEY.md section 2)
    with pytest.raises(K.MjhipError, match="unsupported.*sleep"):
        K.DeviceModel(hostsim_lib, sc)


FORWARD_FIELDS_SC = ["site_xpos", "site_xmat", "actuator_length", "actuator_velocity", "actuator_force",
                     "qfrc_actuator", "qacc_smooth", "qacc"]


@pytest.mark.parametrize("layout", ["aos", "soa"])
def test_slider_crank_bit_exact(rb, hostsim_lib, golden, layout):
    """BASELINE config 1 (model/slider_crank): slider-crank transmissions (engine_core_smooth.c:1396),
    position actuators, cylinder geoms -- forward fields and the golden trajectory, bit for bit"""
    m = rb.MjModel.from_binary_path(os.path.join(GOLDEN, "slider_crank.mjb"))
    m.opt.solver = 0
    dm = K.DeviceModel(hostsim_lib, m)
    fx = golden("slider_crank")
    n, T = 4, 40
    b = K.Batch(dm, n, layout=layout)
    out = b.rollout_host(T, K.mjSTATE_CTRL, fx["state0"][:n], None, fx["ctrl"][:n, :T])
    assert np.array_equal(out, fx["state"][:n, :T])
    assert b.get("warning").sum() == 0
    # forward-pass fields of the transmission against the live oracle
    d = rb.MjData(m)
    b.set("qpos", fx["state"][:n, T - 1, 1:1 + m.nq]); b.set("qvel", fx["state"][:n, T - 1, 1 + m.nq:])
    b.set("ctrl", fx["ctrl"][:n, T - 1])
    b.forward()
    for e in range(n):
        d.qpos[:] = fx["state"][e, T - 1, 1:1 + m.nq]; d.qvel[:] = fx["state"][e, T - 1, 1 + m.nq:]
        d.ctrl[:] = fx["ctrl"][e, T - 1]
        d.qacc_warmstart[:] = b.get("qacc_warmstart")[e]
        rb.mj_forward(m, d)
        for f in FORWARD_FIELDS_SC:
            ref = np.asarray(getattr(d, f)).ravel()
            assert np.array_equal(b.get(f)[e][:ref.size], ref), (e, f)





def test_plane_cylinder_collider_bit_exact(rb, hostsim_lib, tmp_path):
    """mjc_PlaneCylinder (engine_collision_primitive.c:101-208): tilted, lying and upright cylinders
    dropping on a plane -- up to 4 contacts per pair, pyramidal and frictionless"""
    xml = tmp_path / "cyl.xml"
    xml.write_text(CYL_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.solver = 0
    dm = K.DeviceModel(hostsim_lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 90
    ctrl = np.zeros((1, T, 0))
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    assert ints[0, :, 0].max() >= 6, "the scene is supposed to produce multi-point cylinder contacts"
    b = K.Batch(dm, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    assert np.array_equal(out, ref)
    assert b.get("warning").sum() == 0
    assert b.get("counts")[0, 0] == ints[0, -1, 0]


@pytest.mark.parametrize("solver,tol", [(0, 0.0), (2, 1e-8)])
def test_equality_constraints(rb, hostsim_lib, tmp_path, solver, tol):
    """connect / weld / joint / tendon equalities (mj_instantiateEquality,
    engine_core_constraint.c:800-1110) incl. site-based anchors, torquescale, polynomial couplings, the
    Jdot*v reference term, an inactive equality and a static-static weld that the empty-Jacobian
    guard of mj_addConstraint drops -- PGS bit-exact, Newton to solver round-off"""
    xml = tmp_path / "eq.xml"
    xml.write_text(EQ_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.solver = solver
    dm = K.DeviceModel(hostsim_lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 80 if solver == 0 else 16
    ctrl = np.random.default_rng(5).uniform(-1, 1, (1, T, m.nu))
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    assert ints[0, :, 0].max() >= 3 and ints[0, :, 1].max() >= 30
    b = K.Batch(dm, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    if tol == 0.0:
        assert np.array_equal(out, ref)
    else:
        assert relerr(out, ref) <= tol
    assert b.get("warning").sum() == 0
    c = b.get("counts")[0]
    assert c[0] == ints[0, -1, 0] and c[1] == ints[0, -1, 1]


@pytest.mark.parametrize("solver,tol", [(0, 0.0), (2, 1e-9)])
def test_implicitfast_integrator(rb, hostsim_lib, tmp_path, solver, tol):
    """mj_implicitSkip, implicitfast branch (engine_forward.c:1649-1770): qH = M - h*qDeriv with the
    actuator / joint-damper / tendon-damper velocity derivatives (mjd_actuator_vel, mjd_passive_vel)
    and the unsymmetric 6x6 solve of standalone free bodies (mjd_freeMhat)"""
    xml = tmp_path / "impl.xml"
    xml.write_text(IMPL_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.solver = solver
    dm = K.DeviceModel(hostsim_lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[:] = np.random.default_rng(1).normal(0, 1.0, m.nv)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 80 if solver == 0 else 40
    ctrl = np.random.default_rng(0).uniform(-1, 1, (1, T, m.nu))
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dm, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    if tol == 0.0:
        assert np.array_equal(out, ref)
    else:
        assert relerr(out, ref) <= tol
    assert b.get("warning").sum() == 0


@pytest.mark.parametrize("solver,tol", [(0, 0.0), (2, 1e-9)])
def test_implicit_integrator(rb, hostsim_lib, tmp_path, solver, tol):
    """mj_implicitSkip, fully implicit branch (engine_forward.c:1680-1690, :1718-1733): qDeriv on its own pattern with the
    derivative of the bias force (mjd_rne_vel / mjd_comVel_vel, engine_derivative.c:527-720) next to the actuator /
    damper terms, qLU = M - h qDeriv, mju_factorLUSparse / mju_solveLUSparse; hinge, slide and free joints, a free
    body with a child, tendon damping"""
    xml = tmp_path / "impl.xml"
    xml.write_text(IMPL_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.solver = solver
    m.opt.integrator = 2
    dm = K.DeviceModel(hostsim_lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    # (three environments with their own initial velocities and controls: every per-environment address is exercised)
    NE = 3
    s0 = np.zeros((NE, rb.mj_stateSize(m, rb.mjSTATE_FULLPHYSICS)))
    for k in range(NE):
        d.qvel[:] = np.random.default_rng(1 + k).normal(0, 1.0, m.nv)
        s0[k] = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
    T = 80 if solver == 0 else 40
    ctrl = np.random.default_rng(0).uniform(-1, 1, (NE, T, m.nu))
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dm, NE)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    if tol == 0.0:
        assert np.array_equal(out, ref)
    else:
        assert relerr(out, ref) <= tol
    assert b.get("warning").sum() == 0


@pytest.mark.parametrize("integrator", [2, 3])
def test_implicitfast_humanoid_bit_exact(rb, hostsim_lib, golden, integrator):
    """the BASELINE humanoid with integrator=implicitfast (damping-only qDeriv, one tree) and -- round 6 -- implicit
    (ball-free chain of 27 dofs through mjd_rne_vel and the sparse LU)"""
    m = rb.MjModel.from_binary_path(os.path.join(GOLDEN, "humanoid.mjb"))
    m.opt.solver = 0
    m.opt.integrator = integrator
    dm = K.DeviceModel(hostsim_lib, m)
    fx = golden("humanoid")
    T = 12
    s0, ctrl = fx["state0"][:2], fx["ctrl"][:2, :T]
    ref, _ = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dm, 2)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    assert np.array_equal(out, ref)


@pytest.mark.parametrize("cone", [0, 1])
def test_box_and_cylinder_colliders_bit_exact(rb, hostsim_lib, tmp_path, cone):
    """mjc_PlaneBox (engine_collision_primitive.c:210), mjraw_SphereBox (engine_collision_box.c:35)
    and mjc_SphereCylinder (engine_collision_primitive.c:345): boxes settling on a plane with 1..4
    corner contacts, spheres on a box and on / beside / at the rim of a cylinder"""
    xml = tmp_path / "box.xml"
    xml.write_text(BOX_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.cone = cone
    dm = K.DeviceModel(hostsim_lib, m, 64, 200)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[:] = np.random.default_rng(4).normal(0, .3, m.nv)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 100
    ctrl = np.zeros((1, T, 0))
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    assert ints[0, :, 0].max() >= 15
    b = K.Batch(dm, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    assert np.array_equal(out, ref)
    assert b.get("warning").sum() == 0
    c = b.get("counts")[0]
    assert c[0] == ints[0, -1, 0] and c[1] == ints[0, -1, 1]


@pytest.mark.parametrize("cone", [0, 1])
def test_box_box_collider_bit_exact(rb, hostsim_lib, tmp_path, cone):
    """mjc_BoxBox (engine_collision_box.c:697-1066): separating-axis search, edge-edge witness
    points, face clipping with up to 8 contacts per pair; stacks, a rotated box on a table, a
    crossing of two bars, a corner landing"""
    xml = tmp_path / "boxbox.xml"
    xml.write_text(BOXBOX_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.cone = cone
    dm = K.DeviceModel(hostsim_lib, m, 64, 200)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[:] = np.random.default_rng(4).normal(0, .3, m.nv)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 150
    ctrl = np.zeros((1, T, 0))
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    assert ints[0, :, 0].max() >= 20
    b = K.Batch(dm, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    assert np.array_equal(out, ref)
    assert b.get("warning").sum() == 0
    c = b.get("counts")[0]
    assert c[0] == ints[0, -1, 0] and c[1] == ints[0, -1, 1]


@pytest.mark.parametrize("cone", [0, 1])
def test_capsule_box_collider_bit_exact(rb, hostsim_lib, tmp_path, cone):
    """mjc_CapsuleBox (engine_collision_box.c:114-653): 2 end-point + 12 edge candidates on 14 lanes,
    in-order winner, second sphere by the corner / edge / face case analysis; a scene of capsules
    lying on, standing on, hanging over and crossing boxes"""
    xml = tmp_path / "capbox.xml"
    xml.write_text(CAPBOX_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.cone = cone
    dm = K.DeviceModel(hostsim_lib, m, 64, 200)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[:] = np.random.default_rng(4).normal(0, .3, m.nv)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 150
    ctrl = np.zeros((1, T, 0))
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    assert ints[0, :, 0].max() >= 8
    b = K.Batch(dm, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    assert np.array_equal(out, ref)
    assert b.get("warning").sum() == 0
    c = b.get("counts")[0]
    assert c[0] == ints[0, -1, 0] and c[1] == ints[0, -1, 1]


def test_capsule_box_random_poses_bit_exact(rb, hostsim_lib, tmp_path):
    """one capsule in 300 random / upright / lying / exactly axis-aligned poses around a box: contact
    count, distance, position and frame equal the reference's bit for bit (every branch of the
    closest-feature search: faces, edge interiors, corners, parallel-axis skips)"""
    xml = tmp_path / "capbox1.xml"
    xml.write_text("""
<mujoco><worldbody>
  <geom name="table" type="box" size=".4 .3 .05" pos="0 0 .3" euler="0 0 15"/>
  <body pos="0 0 .5"><freejoint/><geom type="capsule" size=".04 .15" condim="3"/></body>
</worldbody></mujoco>""")
    m = rb.MjModel.from_xml_path(str(xml))
    dm = K.DeviceModel(hostsim_lib, m, 8, 32)
    d = rb.MjData(m)
    rng = np.random.default_rng(11)
    N = 300
    qpos = np.zeros((N, 7))
    tz = np.array([np.cos(np.radians(7.5)), 0, 0, np.sin(np.radians(7.5))])     # the table's yaw
    for k in range(N):
        q = rng.normal(size=4)
        if k % 4 == 1: q = np.array([1, 0, 0, 0.]) + rng.normal(size=4)*1e-3
        if k % 4 == 2: q = np.array([1, 1, 0, 0.]) + rng.normal(size=4)*1e-3
        if k % 4 == 3: q = tz.copy()
        if k % 8 == 7:       # lying along the table's x axis: yaw * (90 degrees about y)
            w1, x1, y1, z1 = tz; w2, x2, y2, z2 = np.cos(np.pi/4), 0, np.sin(np.pi/4), 0
            q = np.array([w1*w2 - x1*x2 - y1*y2 - z1*z2, w1*x2 + x1*w2 + y1*z2 - z1*y2,
                          w1*y2 - x1*z2 + y1*w2 + z1*x2, w1*z2 + x1*y2 - y1*x2 + z1*w2])
        qpos[k, :3] = [rng.uniform(-.55, .55), rng.uniform(-.45, .45), rng.uniform(.2, .55)]
        qpos[k, 3:] = q / np.linalg.norm(q)
    b = K.Batch(dm, N)
    b.reset()
    b.set("qpos", qpos)
    b.forward()
    counts = b.get("counts")[:, 0]
    cd = b.get("con_dist"); cp = b.get("con_pos").reshape(N, -1, 3); cf = b.get("con_frame").reshape(N, -1, 9)
    hist = {0: 0, 1: 0, 2: 0}
    for k in range(N):
        rb.mj_resetData(m, d)
        d.qpos[:] = qpos[k]
        rb.mj_forward(m, d)
        n = d.ncon
        hist[n] += 1
        assert counts[k] == n, k
        rc = d.contact[:n]
        assert np.array_equal(cd[k, :n], rc["dist"]) and np.array_equal(cp[k, :n], rc["pos"]) and np.array_equal(cf[k, :n], rc["frame"]), k
    assert hist[1] > 50 and hist[2] > 20


def test_many_filter_survivors_and_wide_pgs_bit_exact(rb, hostsim_lib, tmp_path):
    """90 spheres over a plane: > 64 pairs pass the bounding-sphere filter (chunk-by-chunk narrowphase
    instead of the single compacted round), up to 120 constraint rows (two-per-lane PGS), nv = 540"""
    xml = tmp_path / "many.xml"
    xml.write_text(many_spheres_xml())
    m = rb.MjModel.from_xml_path(str(xml))
    dm = K.DeviceModel(hostsim_lib, m, 128, 512)       # 90 contacts x 4 rows in the end
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 15                  # 30 / 60 / 90 spheres in contact from step 12 on: 120, 240, 360 rows
    ctrl = np.zeros((1, T, 0))
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    assert ints[0, :, 1].max() > 64
    b = K.Batch(dm, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    assert np.array_equal(out, ref)
    c = b.get("counts")[0]
    assert c[0] == ints[0, -1, 0] and c[1] == ints[0, -1, 1] and c[5] == ints[0, -1, 2]
    assert b.get("warning").sum() == 0


def _mocap_controls(m, T):
    ctrl = np.zeros((1, T, m.nu + 7*m.nmocap))
    for t in range(T):
        ctrl[0, t, 0] = 0.02*np.sin(t/7)
        ctrl[0, t, 1:4] = [.1*np.sin(t/20), .05*t/T, .4 + .05*np.cos(t/15)]
        ctrl[0, t, 4:7] = [.6 + .3*t/T, .02*np.sin(t/9), .06]
        ctrl[0, t, 7:11] = [1, .1*np.sin(t/11), 0, .2*t/T]        # not normalised on purpose
        ctrl[0, t, 11:15] = [np.cos(.2 + t/60), 0, 0, np.sin(.2 + t/60)]
    return ctrl


def test_mocap_bodies_bit_exact(rb, hostsim_lib, tmp_path):
    """mocap bodies (mj_kinematics, engine_core_smooth.c:84-93) driven through mjSTATE_MOCAP_POS |
    mjSTATE_MOCAP_QUAT in the control spec of the rollout; a weld to a mocap target and a mocap
    paddle pushing a sphere.  A second rollout without the mocap bits sees the model's poses again
    (inputs outside the control spec are reset, rollout.cc:85-115)."""
    xml = tmp_path / "mocap.xml"
    xml.write_text(MOCAP_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    dm = K.DeviceModel(hostsim_lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 80
    spec = K.mjSTATE_CTRL | K.mjSTATE_MOCAP_POS | K.mjSTATE_MOCAP_QUAT
    ctrl = _mocap_controls(m, T)
    ref = np.zeros((1, T, s0.shape[1]))
    for t in range(T):
        rb.mj_setState(m, d, ctrl[0, t], spec)
        rb.mj_step(m, d)
        ref[0, t] = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
    b = K.Batch(dm, 1)
    out = b.rollout_host(T, spec, s0, None, ctrl)
    assert np.array_equal(out, ref)
    rb.mj_resetData(m, d)
    ref2 = np.zeros((1, T, s0.shape[1]))
    for t in range(T):
        d.ctrl[:] = ctrl[0, t, :m.nu]
        rb.mj_step(m, d)
        ref2[0, t] = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
    out2 = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl[:, :, :m.nu].copy())
    assert np.array_equal(out2, ref2)
    assert b.get("warning").sum() == 0


def test_gravity_compensation_and_diagnostic_flags(rb, hostsim_lib, tmp_path):
    """body gravcomp (mj_gravcomp, engine_passive.c:846-867) with a tilted gravity vector; the model
    also sets mjENBL_ENERGY | mjENBL_FWDINV, which only fill diagnostics outside the rollout outputs"""
    xml = tmp_path / "gc.xml"
    xml.write_text(ACT_XML.replace('<body name="a2" pos=".25 0 0">', '<body name="a2" pos=".25 0 0" gravcomp=".7">')
                   .replace('<body name="b1" pos="0 .5 .4">', '<body name="b1" pos="0 .5 .4" gravcomp="1">')
                   .replace('<option timestep="0.004"', '<option gravity=".5 -.3 -9.81" timestep="0.004"'))
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.enableflags |= (1 << 1) | (1 << 2)
    dm = K.DeviceModel(hostsim_lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[:] = np.random.default_rng(1).normal(0, .5, m.nv)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 60
    ctrl = np.random.default_rng(0).uniform(-1, 1, (1, T, m.nu))
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dm, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    assert np.array_equal(out, ref)


@pytest.mark.parametrize("cone", [0, 1])
def test_predefined_contact_pairs_bit_exact(rb, hostsim_lib, tmp_path, cone):
    """<contact><pair> entries merged into the candidate list in signature order
    (mj_collision, engine_collision_driver.c:651-664), with their own condim / friction / solref /
    solreffriction / margin / gap, replacing the automatic pair of the same geoms and reaching
    parent-child geoms the automatic filter drops; an <exclude> next to them"""
    xml = tmp_path / "pair.xml"
    xml.write_text(PAIR_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.cone = cone
    assert m.npair == 3
    dm = K.DeviceModel(hostsim_lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[:] = np.random.default_rng(2).normal(0, .6, m.nv)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 120
    ctrl = np.zeros((1, T, 0))
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    assert ints[0, :, 0].max() >= 6
    b = K.Batch(dm, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    assert np.array_equal(out, ref)
    c = b.get("counts")[0]
    assert c[0] == ints[0, -1, 0] and c[1] == ints[0, -1, 1]


@pytest.mark.parametrize("scene,opts,exact", [
    ("IMPL_XML", {}, True), ("ACT_XML", {"integrator": 1}, True), ("EQ_XML", {}, True),
    ("BOXBOX_XML", {}, True), ("CONDIM_XML", {"cone": 1}, True), ("PAIR_XML", {"cone": 1}, True),
    ("CONDIM_XML", {"cone": 1, "solver": 2}, False)])
def test_soa_pipeline_feature_scenes(rb, hostsim_lib, tmp_path, scene, opts, exact):
    """the per-step SoA pipeline (lane-per-environment smooth / integrate kernels + the wave-mode
    constraint kernel) on the feature scenes: implicitfast, stateful actuators under RK4,
    equalities, box-box, elliptic condim 4/6, predefined pairs, multi-island Newton"""
    import parity_utils
    xml = tmp_path / "scene.xml"
    xml.write_text(getattr(parity_utils, scene))
    m = rb.MjModel.from_xml_path(str(xml))
    for k, v in opts.items():
        setattr(m.opt, k, v)
    dm = K.DeviceModel(hostsim_lib, m, 64, 200)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[:] = np.random.default_rng(1).normal(0, .4, m.nv)
    s0 = np.tile(rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS), (3, 1))
    T = 25
    ctrl = np.random.default_rng(0).uniform(-1, 1, (3, T, m.nu))
    ref, _ = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dm, 3, layout="soa")
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    if exact:
        assert np.array_equal(out, ref)
    else:
        assert relerr(out, ref) <= 1e-9
    assert b.get("warning").sum() == 0


@pytest.mark.parametrize("integrator", [0, 1, 2, 3])
def test_fluid_forces_bit_exact(rb, hostsim_lib, tmp_path, integrator):
    """mj_fluid, inertia-box model (engine_passive.c:871-903, :1154-1210): viscous and quadratic drag
    with wind on a swimmer-like chain and a tumbling box; Euler and RK4, and -- round 6 -- the implicit
    integrators, whose qDeriv takes the forces' velocity derivative (mjd_inertiaBoxFluid,
    engine_derivative.c:2884-3038)"""
    xml = tmp_path / "fluid.xml"
    xml.write_text(FLUID_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.integrator = integrator
    dm = K.DeviceModel(hostsim_lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[:] = np.random.default_rng(1).normal(0, 1.0, m.nv)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 80
    ctrl = np.random.default_rng(0).uniform(-1, 1, (1, T, m.nu))
    ref, _ = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dm, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    assert np.array_equal(out, ref)


@pytest.mark.parametrize("integrator", [0, 1, 2, 3])
def test_ellipsoid_fluid_model_bit_exact(rb, hostsim_lib, tmp_path, integrator):
    """mj_ellipsoidFluidModel (engine_passive.c:1213-1410): added mass, Magnus and Kutta lift, viscous drag and torque per
    geom, next to inertia-box bodies; under the implicit integrators the 6 x 6 derivative of every geom's wrench enters
    qDeriv (mjd_ellipsoidFluid, engine_derivative.c:2531-2880; symmetrised under implicitfast except on standalone free bodies)"""
    xml = tmp_path / "efluid.xml"
    xml.write_text(ELLIPSOID_FLUID_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.integrator = integrator
    dm = K.DeviceModel(hostsim_lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    # (three environments with their own initial velocities and controls: every per-environment address is exercised)
    NE = 3
    s0 = np.zeros((NE, rb.mj_stateSize(m, rb.mjSTATE_FULLPHYSICS)))
    for k in range(NE):
        d.qvel[:] = np.random.default_rng(1 + k).normal(0, 1.0, m.nv)
        s0[k] = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
    T = 60
    ctrl = np.random.default_rng(0).uniform(-1, 1, (NE, T, m.nu))
    ref, _ = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dm, NE)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    assert np.array_equal(out, ref)


@pytest.mark.parametrize("lds", [0, 3328, 10240, 40000])
def test_newton_islands_any_lds_budget(rb, hostsim_lib, tmp_path, lds):
    """regression: with several islands the primal solver must not touch the forces of islands it
    has not solved yet (they sit in uninitialised LDS once the plan places efc_force there)"""
    xml = tmp_path / "islands.xml"
    xml.write_text(ISLANDS_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    assert m.opt.solver == 2
    dm = K.DeviceModel(hostsim_lib, m, 64, 300)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[:] = np.random.default_rng(1).normal(0, .5, m.nv)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 20
    ref, _ = oracle_rollout(rb, m, s0, np.zeros((1, T, 0)))
    b = K.Batch(dm, 1)
    b.plan_lds(lds)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, np.zeros((1, T, 0)))
    assert np.isfinite(out).all()
    assert relerr(out, ref) <= 1e-9
    assert b.get("warning").sum() == 0


# islands that leave trees out, on the reference's DENSE path (nv < 60): an arm that touches nothing, a ball in free fall,
# a stack of two boxes on the floor (one island of 12 dofs among 22) and a capsule lying apart (a second island)
PARTIAL_ISLANDS_XML = """
<mujoco>
  <option timestep="0.004" solver="Newton" iterations="50" tolerance="1e-10"/>
  <default><geom condim="3" friction=".6"/></default>
  <worldbody>
    <geom type="plane" size="3 3 .01"/>
    <body pos="-1 0 1"><joint axis="0 1 0" damping=".1"/><geom type="capsule" size=".03" fromto="0 0 0 .4 0 0"/>
      <body pos=".4 0 0"><joint axis="0 1 0" damping=".1"/><geom type="capsule" size=".03" fromto="0 0 0 .3 0 0"/></body></body>
    <body pos="0 0 .1"><freejoint/><geom type="box" size=".1 .1 .1"/></body>
    <body pos="1 1 2"><freejoint/><geom type="sphere" size=".05"/></body>
    <body pos=".03 .02 .2999" euler="0 0 20"><freejoint/><geom type="box" size=".08 .08 .1"/></body>
    <body pos="1.5 0 .05" euler="90 0 0"><freejoint/><geom type="capsule" size=".05 .2"/></body>
  </worldbody>
</mujoco>
"""


def _partial_islands(rb, lib, tmp_path, solver, cone, kind=None):
    xml = tmp_path / "partial_islands.xml"
    xml.write_text(PARTIAL_ISLANDS_XML)
    m = rb.MjModel.from_xml_path(str(xml), kind=kind) if kind else rb.MjModel.from_xml_path(str(xml))
    m.opt.solver = solver
    m.opt.cone = cone
    assert m.nv == 26 and not (m.opt.disableflags & (1 << 18))
    dm = K.DeviceModel(lib, m, 64, 300)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    NE = 2
    s0 = np.zeros((NE, rb.mj_stateSize(m, rb.mjSTATE_FULLPHYSICS)))
    for k in range(NE):
        d.qvel[:] = np.random.default_rng(3 + k).normal(0, .3, m.nv)
        s0[k] = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
    T = 80
    ctrl = np.zeros((NE, T, 0))
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    assert ints[:, :, 0].max() >= 6
    b = K.Batch(dm, NE)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    assert b.get("warning").sum() == 0
    return out, ref


@pytest.mark.parametrize("solver,cone", [(2, 0), (2, 1), (1, 0), (1, 1)])
def test_primal_solvers_partial_islands_dense_bit_exact(rb, hostsim_lib, tmp_path, solver, cone):
    """Newton / CG on the dense path when an island leaves trees out: the reference solves every island on island-local
    vectors and matrices (PrimalPointers, engine_solver.c:1148-1290), so mju_dot groups operands by their position INSIDE
    the island -- J rows, vector dots, the Cholesky factor's row dots and the forward substitution.  (Found by the
    100-step device sweep on actuator_group_disable.xml: 1e-14 after the first multi-contact step, 4e-3 eighty steps on.)"""
    out, ref = _partial_islands(rb, hostsim_lib, tmp_path, solver, cone)
    assert np.array_equal(out, ref)


# mj_solNoSlip: friction-loss rows, pyramidal and elliptic contacts of condim 3 / 4 / 6, several islands and an arm that
# touches nothing
NOSLIP_XML = """
<mujoco>
  <option timestep="0.004" noslip_iterations="4" noslip_tolerance="1e-8" iterations="60"/>
  <default><geom condim="3" friction=".7 .02 .01"/></default>
  <worldbody>
    <geom type="plane" size="3 3 .01" euler="4 -3 0"/>
    <body pos="-1 0 1"><joint axis="0 1 0" damping=".1" frictionloss=".3"/><geom type="capsule" size=".03" fromto="0 0 0 .4 0 0"/>
      <body pos=".4 0 0"><joint axis="0 1 0" damping=".1" frictionloss=".2"/><geom type="capsule" size=".03" fromto="0 0 0 .3 0 0"/></body></body>
    <body pos="0 0 .12"><freejoint/><geom type="box" size=".1 .1 .1"/></body>
    <body pos=".03 .02 .33" euler="0 0 20"><freejoint/><geom type="box" size=".08 .08 .1" condim="4"/></body>
    <body pos="1.5 0 .08" euler="90 0 0"><freejoint/><geom type="capsule" size=".05 .2" condim="6"/></body>
    <body pos="0 1.2 .2"><joint type="slide" axis="1 0 0" frictionloss=".5"/><joint type="slide" axis="0 0 1"/>
      <geom type="sphere" size=".07" condim="4"/></body>
  </worldbody>
</mujoco>
"""


def _noslip(rb, lib, tmp_path, solver, cone, kind=None):
    xml = tmp_path / "noslip.xml"
    xml.write_text(NOSLIP_XML)
    m = rb.MjModel.from_xml_path(str(xml), kind=kind) if kind else rb.MjModel.from_xml_path(str(xml))
    m.opt.solver = solver
    m.opt.cone = cone
    dm = K.DeviceModel(lib, m, 64, 300)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    NE = 2
    s0 = np.zeros((NE, rb.mj_stateSize(m, rb.mjSTATE_FULLPHYSICS)))
    for k in range(NE):
        d.qvel[:] = np.random.default_rng(5 + k).normal(0, .4, m.nv)
        s0[k] = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
    T = 90
    ctrl = np.zeros((NE, T, 0))
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    assert ints[:, :, 0].max() >= 6
    b = K.Batch(dm, NE)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    assert b.get("warning").sum() == 0
    # the iteration count of island 0 includes the noslip iterations (engine_solver.c:951)
    c = b.get("counts")
    assert np.array_equal(c[:, 5], ints[:, -1, 2]), (c[:, 5], ints[:, -1, 2])
    # and the same trajectory WITHOUT the noslip pass differs (the pass does something in this scene)
    m.opt.noslip_iterations = 0
    ref0, _ = oracle_rollout(rb, m, s0, ctrl)
    assert relerr(ref0, ref) > 1e-6
    return out, ref


@pytest.mark.parametrize("solver,cone", [(0, 0), (0, 1), (2, 0), (2, 1), (1, 1)],
                         ids=["pgs-pyr", "pgs-ell", "newton-pyr", "newton-ell", "cg-ell"])
def test_noslip_solver_bit_exact(rb, hostsim_lib, tmp_path, solver, cone):
    """mj_solNoSlip (engine_solver.c:764-972) after PGS, Newton and CG, per island: dry-friction rows, opposing pyramid edge
    pairs, the QCQP of an elliptic contact's friction dimensions (condim 3 / 4 / 6), efc_AR built under the primal solvers
    too (mj_isDual), mj_dualFinish afterwards"""
    out, ref = _noslip(rb, hostsim_lib, tmp_path, solver, cone)
    assert np.array_equal(out, ref), relerr(out, ref)


@pytest.mark.parametrize("solver,integrator,tol", [(0, 0, 0.0), (2, 0, 1e-9), (0, 1, 0.0), (0, 3, 0.0)])
def test_spatial_tendons(rb, hostsim_lib, tmp_path, solver, integrator, tol):
    """spatial tendons through sites with pulleys (mj_tendon, engine_core_smooth.c:988-1105): lengths,
    sparse moments from end-point Jacobian differences; spring-dampers, a limit and a tendon
    equality on top; tendon transmissions (position / motor / filtered general actuators on tendons)"""
    xml = tmp_path / "tendon.xml"
    xml.write_text(TENDON_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.solver = solver
    m.opt.integrator = integrator    # implicitfast: tendon damping enters a standalone free body's 6x6 block
    dm = K.DeviceModel(hostsim_lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[:] = np.random.default_rng(1).normal(0, .5, m.nv)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 100
    ctrl = np.random.default_rng(0).uniform(-1, 1, (1, T, m.nu))
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dm, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    if tol == 0.0:
        assert np.array_equal(out, ref)
    else:
        assert relerr(out, ref) <= tol
    assert b.get("warning").sum() == 0
    # the tendon quantities themselves after mj_forward
    rb.mj_setState(m, d, ref[0, -1], rb.mjSTATE_FULLPHYSICS)
    d.ctrl[:] = ctrl[0, -1]
    rb.mj_forward(m, d)
    b.forward()
    assert relerr(b.get("ten_length")[0], np.array(d.ten_length)) <= 1e-12
    assert relerr(b.get("ten_J")[0][:m.nJten], np.array(d.ten_J)[:m.nJten]) <= 1e-12
    assert relerr(b.get("sensordata")[0], np.array(d.sensordata)) <= 1e-12

def _tendon_armature(rb, lib, tmp_path, jacobian, integrator):
    """tendon armature (mj_tendonArmature: M += armature ten_J' ten_J on M's pattern; mj_tendonBias with mj_tendonDot: the
    time derivative of a spatial tendon's Jacobian through mj_jacDot; engine_core_smooth.c:1115-1260, :1845-1886,
    :2606-2641) on spatial tendons incl. a pulley branch, on a fixed tendon, and as an actuator's armature on a tendon
    transmission (mj_actuatorArmature); dense (mju_dot) and sparse (running sum over the merged chain) contractions"""
    xml = tmp_path / "tendon_armature.xml"
    text = TENDON_XML.replace('<spatial name="sp1"', '<spatial name="sp1" armature=".3"').replace(
        '<spatial name="pul"', '<spatial name="pul" armature=".12"').replace('<fixed name="fx">', '<fixed name="fx" armature=".05">').replace(
        '<motor tendon="pul" gear="2"/>', '<motor tendon="pul" gear="2" armature=".02"/>')
    assert text.count("armature") == 4
    xml.write_text(text)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.jacobian = jacobian
    m.opt.integrator = integrator
    dm = K.DeviceModel(lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[:] = np.random.default_rng(4).normal(0, .7, m.nv)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 60
    ctrl = np.random.default_rng(2).uniform(-1, 1, (1, T, m.nu))
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dm, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    assert b.get("warning").sum() == 0
    # the inertia matrix and the bias force after mj_forward at the last state
    rb.mj_setState(m, d, ref[0, -1], rb.mjSTATE_FULLPHYSICS)
    d.ctrl[:] = ctrl[0, -1]
    rb.mj_forward(m, d)
    b.forward()
    return out, ref, relerr(b.get("M")[0][:m.nC], np.array(d.M)[:m.nC]), relerr(b.get("qfrc_bias")[0], np.array(d.qfrc_bias))


@pytest.mark.parametrize("jacobian,integrator", [(0, 0), (1, 0), (0, 1), (1, 3)])
def test_tendon_armature_bit_exact(rb, hostsim_lib, tmp_path, jacobian, integrator):
    out, ref, eM, eb = _tendon_armature(rb, hostsim_lib, tmp_path, jacobian, integrator)
    assert np.array_equal(out, ref)
    assert eM == 0.0 and eb == 0.0


def _tendon_wrap(rb, lib, tmp_path, integrator, T=150):
    """tendons wrapping around spheres and cylinders (mju_wrap, engine_util_misc.c:36-413, inside mj_tendon,
    engine_core_smooth.c:1024-1100): tangent points on the circle through the two sites, the side-site rule, the inside
    wrap, the cylinder's height correction; lengths and moments, through a swinging motion that wraps and unwraps"""
    xml = tmp_path / "wrap.xml"
    xml.write_text(WRAP_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.integrator = integrator
    dm = K.DeviceModel(lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[:] = np.random.default_rng(7).normal(0, 1.5, m.nv)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    ctrl = np.random.default_rng(3).uniform(-1, 1, (1, T, m.nu))
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dm, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    assert b.get("warning").sum() == 0
    # lengths and moments along the reference trajectory: how many of the steps wrapped
    nwrap, eL, eJ = 0, 0.0, 0.0
    for t in range(0, T, 10):
        rb.mj_setState(m, d, ref[0, t], rb.mjSTATE_FULLPHYSICS)
        rb.mj_forward(m, d)
        bb = K.Batch(dm, 1)
        load = {"time": ref[0, t, :1], "qpos": ref[0, t, 1:1 + m.nq], "qvel": ref[0, t, 1 + m.nq:1 + m.nq + m.nv]}
        for k, v in load.items():
            bb.set(k, v[None])
        bb.forward()
        eL = max(eL, relerr(bb.get("ten_length")[0], np.array(d.ten_length)))
        eJ = max(eJ, relerr(bb.get("ten_J")[0][:m.nJten], np.array(d.ten_J)[:m.nJten]))
        nwrap += int(np.sum(np.array(d.wrap_obj)[:int(np.sum(np.array(d.ten_wrapnum)))] >= 0))
        bb.close()
    return out, ref, eL, eJ, nwrap


@pytest.mark.parametrize("integrator", [0, 1, 3])
def test_tendon_wrapping_bit_exact(rb, hostsim_lib, tmp_path, integrator):
    out, ref, eL, eJ, nwrap = _tendon_wrap(rb, hostsim_lib, tmp_path, integrator)
    assert nwrap > 10                       # wrap points were active on the sampled steps
    assert eL == 0.0 and eJ == 0.0
    assert np.array_equal(out, ref)


def _actuator_groups(rb, lib, tmp_path, integrator, T=80):
    """disabled actuator groups and tendon-level actuator force limits: forces after mj_forward, states and activations
    over a rollout with controls that drive the tendon totals through both ends of their ranges"""
    xml = tmp_path / "actgroups.xml"
    xml.write_text(ACT_GROUP_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.integrator = integrator
    assert m.opt.disableactuator == (1 << 1) | (1 << 3) and m.na == 3
    dm = K.DeviceModel(lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[:] = np.random.default_rng(5).normal(0, .8, m.nv)
    d.act[:] = [.2, -.1, .3]
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    ctrl = np.random.default_rng(6).uniform(-1.5, 1.5, (1, T, m.nu))
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dm, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    assert b.get("warning").sum() == 0
    eF, clamped = 0.0, 0
    for t in range(0, T, 4):
        rb.mj_setState(m, d, ref[0, t], rb.mjSTATE_FULLPHYSICS)
        d.ctrl[:] = ctrl[0, t]
        rb.mj_forward(m, d)
        bb = K.Batch(dm, 1)
        nq, nv = m.nq, m.nv
        bb.set("time", ref[0, t, None, :1]); bb.set("qpos", ref[0, t, None, 1:1 + nq]); bb.set("qvel", ref[0, t, None, 1 + nq:1 + nq + nv])
        bb.set("act", ref[0, t, None, 1 + nq + nv:1 + nq + nv + m.na]); bb.set("ctrl", ctrl[0, t][None])
        bb.forward()
        f = np.array(d.actuator_force)
        eF = max(eF, relerr(bb.get("actuator_force")[0], f))
        tot = f[5] + f[6] + f[7]
        clamped += int(abs(tot - (-.6)) < 1e-12 or abs(tot - .4) < 1e-12)
        assert f[1] == 0 and f[2] == 0 and f[7] == 0                   # the disabled groups
        bb.close()
    return out, ref, eF, clamped


@pytest.mark.parametrize("integrator", [0, 1, 3])
def test_disabled_actuator_groups_and_tendon_force_limits_bit_exact(rb, hostsim_lib, tmp_path, integrator):
    out, ref, eF, clamped = _actuator_groups(rb, hostsim_lib, tmp_path, integrator)
    assert clamped >= 2                      # the tendon total sat at a limit on some of the sampled steps
    assert eF == 0.0
    assert np.array_equal(out, ref)


def _muscles(rb, lib, tmp_path, integrator, T=120):
    """muscle actuators (mju_muscleDynamics / mju_muscleGain / mju_muscleBias, engine_util_misc.c:1049-1195; the gain's
    velocity derivative in the implicitfast integrator, engine_derivative.c:969): activations, forces, states"""
    xml = tmp_path / "muscle.xml"
    xml.write_text(MUSCLE_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.integrator = integrator
    assert m.na == 4
    dm = K.DeviceModel(lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qpos[:] = [.3, .8]
    d.qvel[:] = [1.5, -2.0]
    d.act[:] = [.1, .6, .9, .3]
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    rng = np.random.default_rng(9)
    ctrl = np.clip(np.repeat(rng.uniform(-.3, 1.3, (1, T//10, m.nu)), 10, axis=1), None, None)     # excitation steps of 10 samples
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dm, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    assert b.get("warning").sum() == 0
    eF, fmax = 0.0, 0.0
    nq, nv = m.nq, m.nv
    for t in range(0, T, 6):
        rb.mj_setState(m, d, ref[0, t], rb.mjSTATE_FULLPHYSICS)
        d.ctrl[:] = ctrl[0, t]
        rb.mj_forward(m, d)
        bb = K.Batch(dm, 1)
        bb.set("time", ref[0, t, None, :1]); bb.set("qpos", ref[0, t, None, 1:1 + nq]); bb.set("qvel", ref[0, t, None, 1 + nq:1 + nq + nv])
        bb.set("act", ref[0, t, None, 1 + nq + nv:1 + nq + nv + m.na]); bb.set("ctrl", ctrl[0, t][None])
        bb.forward()
        f = np.array(d.actuator_force)
        eF = max(eF, relerr(bb.get("actuator_force")[0], f), relerr(bb.get("act_dot")[0], np.array(d.act_dot)))
        fmax = max(fmax, float(np.abs(f).max()))
        bb.close()
    return out, ref, eF, fmax


@pytest.mark.parametrize("integrator", [0, 1, 3])
def test_muscle_actuators_bit_exact(rb, hostsim_lib, tmp_path, integrator):
    out, ref, eF, fmax = _muscles(rb, hostsim_lib, tmp_path, integrator)
    assert fmax > 5.0
    assert eF == 0.0
    assert np.array_equal(out, ref)



@pytest.mark.parametrize("integrator", [0, 1, 3])
def test_site_transmissions_bit_exact(rb, hostsim_lib, tmp_path, integrator):
    """mjTRN_SITE without refsite (engine_core_smooth.c:1573-1593): the gear is a wrench in the site
    frame, the moment row its projection on the site's Jacobians"""
    xml = tmp_path / "site.xml"
    xml.write_text(SITE_ACT_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.integrator = integrator
    dm = K.DeviceModel(hostsim_lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[:] = np.random.default_rng(1).normal(0, .5, m.nv)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 60
    ctrl = np.random.default_rng(0).uniform(-1, 1, (1, T, m.nu))
    ref, _ = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dm, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    assert np.array_equal(out, ref)


@pytest.mark.parametrize("integrator", [0, 1, 3])
def test_ball_and_free_joint_actuators_bit_exact(rb, hostsim_lib, tmp_path, integrator):
    """joint / jointinparent transmissions on ball and free joints: 3D and 6D gears, expmap length
    (engine_core_smooth.c:1331-1392)"""
    xml = tmp_path / "ball.xml"
    xml.write_text(BALL_ACT_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.integrator = integrator
    dm = K.DeviceModel(hostsim_lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[:] = np.random.default_rng(1).normal(0, .5, m.nv)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 80
    ctrl = np.random.default_rng(0).uniform(-.5, .5, (1, T, m.nu))
    ref, _ = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dm, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    assert np.array_equal(out, ref)
    assert b.get("warning").sum() == 0


@pytest.mark.parametrize("cone,solver,tol", [(0, 0, 0.0), (1, 0, 0.0), (1, 2, 1e-9)])
def test_geom_surface_velocity(rb, hostsim_lib, tmp_path, cone, solver, tol):
    """mj_addSurfaceVel (engine_core_constraint.c:3141-3204): the relative surface velocity of the
    contacting geoms enters efc_vel of the tangential / torsional rows (conveyor, turntable)"""
    xml = tmp_path / "sv.xml"
    xml.write_text(SURFACEVEL_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.cone = cone
    m.opt.solver = solver
    dm = K.DeviceModel(hostsim_lib, m, 64, 200)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 100
    ref, ints = oracle_rollout(rb, m, s0, np.zeros((1, T, 0)))
    assert np.abs(ref[0, -1] - s0[0]).max() > 1.0, "the belts are supposed to move the objects"
    b = K.Batch(dm, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, np.zeros((1, T, 0)))
    if tol == 0.0:
        assert np.array_equal(out, ref)
    else:
        assert relerr(out, ref) <= tol
    assert b.get("warning").sum() == 0


@pytest.mark.parametrize("cone,solver,integrator,tol", [(0, 0, 0, 0.0), (1, 0, 0, 0.0), (0, 2, 2, 1e-9), (1, 2, 3, 1e-9), (0, 0, 1, 0.0)])
def test_contact_adhesion(rb, hostsim_lib, tmp_path, cone, solver, integrator, tol):
    """geom / pair adhesion: mj_contactParam's adhesion (engine_collision_driver.c:1763-1779), adhesive contacts active in
    the gap as one frictionless row (mj_setContact :1853-1862), the constant attraction in qfrc_passive (mj_adhesion,
    engine_passive.c:982-1050) and the bias of the rows' reference acceleration (mj_adhesionRef, engine_core_constraint.c:3214)"""
    xml = tmp_path / "adh.xml"
    xml.write_text(ADHESION_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.cone = cone
    m.opt.solver = solver
    m.opt.integrator = integrator
    dm = K.DeviceModel(hostsim_lib, m, 64, 200)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[0] = -1.0                       # the door swings shut and is caught
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 120 if integrator != 1 else 40
    ctrl = np.random.default_rng(5).uniform(0, 1, (1, T, m.nu))      # adhesion actuators (body transmission)
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dm, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    if tol == 0.0:
        assert np.array_equal(out, ref)
    else:
        assert relerr(out, ref) <= tol
    assert b.get("warning").sum() == 0
    # every field of mj_forward along the way (qfrc_passive with the attraction, efc_aref with the bias, contact dims)
    if solver == 0:
        rb.mj_resetData(m, d)
        d.qvel[0] = -1.0
        for stop in (1, 25, min(60, T - 1)):
            while round(d.time / m.opt.timestep) < stop:
                d.ctrl[:] = ctrl[0, int(round(d.time / m.opt.timestep))]
                rb.mj_step(m, d)
            st = [dict(qpos=np.array(d.qpos), qvel=np.array(d.qvel), qacc_warmstart=np.array(d.qacc_warmstart), ctrl=np.array(d.ctrl))]
            assert check_forward(rb, m, b, st, tol=0.0) == 0.0


def _sensor_reference(rb, m, s0, ctrl):
    d = rb.MjData(m)
    T = ctrl.shape[1]
    ref = np.zeros((1, T, s0.shape[1]))
    sref = np.zeros((1, T, m.nsensordata))
    rb.mj_resetData(m, d)
    rb.mj_setState(m, d, s0[0], rb.mjSTATE_FULLPHYSICS)
    for t in range(T):
        d.ctrl[:] = ctrl[0, t]
        rb.mj_step(m, d)
        ref[0, t] = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
        sref[0, t] = d.sensordata
    return ref, sref


@pytest.mark.parametrize("integrator", [0, 1])
def test_sensors_bit_exact(rb, hostsim_lib, tmp_path, integrator):
    """mj_sensorPos/Vel/Acc (engine_sensor.c:1498-1660) with mj_subtreeVel and mj_rnePostConstraint:
    joint/tendon/actuator/ball/limit sensors, frame sensors with and without reference frames,
    subtree COM / velocity / angular momentum, clock, potential and kinetic energy (mj_energyPos / mj_energyVel: joint,
    ball-joint and tendon springs), velocimeter, gyro, accelerometer, force, torque, magnetometer, cutoff -- the `sensordata` output of the rollout, every step"""
    xml = tmp_path / "sens.xml"
    xml.write_text(SENSOR_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.integrator = integrator
    dm = K.DeviceModel(hostsim_lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[:] = np.random.default_rng(1).normal(0, .5, m.nv)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 200 if integrator == 0 else 80           # (RK4: four force evaluations per step on the emulation)
    ctrl = np.random.default_rng(0).uniform(-3, 3, (1, T, m.nu))
    ref, sref = _sensor_reference(rb, m, s0, ctrl)
    b = K.Batch(dm, 1)
    out, sd = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl, want_sensordata=True)
    assert np.array_equal(out, ref)
    assert np.array_equal(sd, sref)
    # mj_forward leaves the same readings in the batch's sensordata field
    rb.mj_setState(m, d, ref[0, -1], rb.mjSTATE_FULLPHYSICS)
    d.ctrl[:] = ctrl[0, -1]
    rb.mj_forward(m, d)
    b.forward()
    assert relerr(b.get("sensordata")[0], np.array(d.sensordata)) <= 1e-12


def test_camera_sensors_bit_exact(rb, hostsim_lib, tmp_path):
    """mj_camlight (engine_core_smooth.c:354-432: fixed / track / trackcom / targetbody / targetbodycom cameras), frame
    sensors attached to and referenced to cameras, camprojection (engine_sensor.c:281-316, :541): sensordata every step"""
    xml = tmp_path / "cam.xml"
    xml.write_text(CAMERA_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    dm = K.DeviceModel(hostsim_lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[:] = np.random.default_rng(1).normal(0, .5, m.nv)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 120
    ctrl = np.random.default_rng(0).uniform(-3, 3, (1, T, m.nu))
    ref, sref = _sensor_reference(rb, m, s0, ctrl)
    b = K.Batch(dm, 1)
    out, sd = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl, want_sensordata=True)
    assert np.array_equal(out, ref)
    assert np.array_equal(sd, sref)


@pytest.mark.parametrize("integrator", [0, 1, 3])
def test_stateful_actuators_bit_exact(rb, hostsim_lib, tmp_path, integrator):
    """act_dot / mj_nextActivation (engine_forward.c:403-447, engine_support.c:706-775): filter,
    filterexact and integrator dynamics, actrange clamp, actearly, activations in the RK4 tableau
    and in the implicitfast actuator derivative -- Euler, RK4 and implicitfast, PGS bit-exact"""
    xml = tmp_path / "act.xml"
    xml.write_text(ACT_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.integrator = integrator
    dm = K.DeviceModel(hostsim_lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[:] = np.random.default_rng(1).normal(0, .5, m.nv)
    d.act[:] = np.random.default_rng(2).normal(0, .2, m.na)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    assert s0.shape[1] == 1 + m.nq + m.nv + m.na
    T = 80
    ctrl = np.random.default_rng(0).uniform(-1.5, 1.5, (1, T, m.nu))
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dm, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    assert np.array_equal(out, ref)
    assert b.get("warning").sum() == 0


@pytest.mark.parametrize("solver", [0, 2])
def test_sparse_jacobian_model_bit_exact(rb, hostsim_lib, tmp_path, solver):
    """nv >= 60 with jacobian=auto: the reference switches to its sparse code paths (mj_isSparse,
    engine_core_util.c:29) and so does this path (mjh_sparse.h; PGS: compressed-row order of the efc_AR sweep).
    67 dofs, up to ~230 constraint rows, 120 steps: bit for bit"""
    xml = tmp_path / "chain.xml"
    xml.write_text(chain_xml(62).replace('jacobian="dense"', 'jacobian="auto"'))
    m = rb.MjModel.from_xml_path(str(xml))
    assert m.nv >= 60
    m.opt.solver = solver
    dm = K.DeviceModel(hostsim_lib, m, 80, 300)
    assert dm.size("sparse") == 1
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[:] = np.random.default_rng(1).normal(0, .5, m.nv)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 120
    ctrl = np.random.default_rng(0).uniform(-1, 1, (1, T, m.nu))
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    assert ints[0, :, 1].max() > 64
    b = K.Batch(dm, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    assert np.array_equal(out, ref)
    assert b.get("warning").sum() == 0
    c = b.get("counts")[0]
    assert c[0] == ints[0, -1, 0] and c[1] == ints[0, -1, 1] and c[5] == ints[0, -1, 2]


@pytest.mark.parametrize("cone,solver,tol", [(0, 0, 0.0), (1, 0, 0.0), (1, 2, 1e-9), (1, 1, 1e-7)])
def test_condim_and_elliptic_cones(rb, hostsim_lib, tmp_path, cone, solver, tol):
    """torsional / rolling friction rows (condim 4 and 6) with pyramidal and elliptic cones
    (mj_instantiateContact engine_core_constraint.c:1617-1712, mj_makeImpedance :2213-2237); the PGS
    block update with ray / QCQP steps (engine_solver.c:598-672) is bit-exact, the primal solvers with
    cone Hessians (mj_constraintUpdate_impl :3352-3452, HessianCone) match to solver round-off"""
    xml = tmp_path / "condim.xml"
    xml.write_text(CONDIM_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.cone = cone
    m.opt.solver = solver
    dm = K.DeviceModel(hostsim_lib, m)
    s0 = condim_scene_state(rb, m)
    T = 50 if solver == 0 else 25
    ctrl = np.zeros((1, T, 0))
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    assert ints[0, :, 0].max() >= 5, "the scene is supposed to keep its bodies in contact"
    b = K.Batch(dm, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    if tol == 0.0:
        assert np.array_equal(out, ref)
        assert b.get("counts")[0, 5] == ints[0, -1, 2]
    else:
        assert relerr(out, ref) <= tol
    assert b.get("warning").sum() == 0
    c = b.get("counts")[0]
    assert c[0] == ints[0, -1, 0] and c[1] == ints[0, -1, 1]


def test_elliptic_humanoid_pgs_bit_exact(rb, hostsim_lib, golden):
    """the BASELINE humanoid with cone=elliptic: foot/floor contacts become 3-row cone blocks"""
    m = rb.MjModel.from_binary_path(os.path.join(GOLDEN, "humanoid.mjb"))
    m.opt.solver = 0
    m.opt.cone = 1
    dm = K.DeviceModel(hostsim_lib, m)
    fx = golden("humanoid")
    T = 10
    s0, ctrl = fx["state0"][2:4], fx["ctrl"][2:4, :T]
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dm, 2)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    assert np.array_equal(out, ref)


def test_deep_chain_generic_paths_bit_exact(rb, hostsim_lib, tmp_path):
    """21-link chain: tree depth 24 (> 16: generic L'DL factor/solve), nefc up to ~80 (> 64: generic PGS
    sweep), ball-joint limit, slide joint, position actuator -- 150 steps, bit for bit"""
    xml = tmp_path / "chain.xml"
    xml.write_text(chain_xml())
    m = rb.MjModel.from_xml_path(str(xml))
    assert max(m.M_rownnz) - 1 > 16
    dm = K.DeviceModel(hostsim_lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 150
    ctrl = np.random.default_rng(1).uniform(-1, 1, size=(1, T, m.nu))
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    assert ints[0, :, 1].max() > 64
    b = K.Batch(dm, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    assert np.array_equal(out, ref)
    assert b.get("warning").sum() == 0


@pytest.mark.parametrize("variant", ["generic", "lean"])
def test_kernel_variants_bit_exact(rb, hostsim_lib, golden, variant):
    """both feature sets of the stage sources (mjh_modes.h: generic or lean) reproduce the oracle bit for bit: forward fields on contact-rich
    states (odd environment count: a partly empty wavefront), then the golden trajectories"""
    from conftest import contact_rich_states, humanoid_pgs_oracle
    from parity_utils import check_forward
    m = humanoid_pgs_oracle(rb)
    mm = K.MjbModel(hostsim_lib, os.path.join(GOLDEN, "humanoid.mjb"))
    mm.set_option("solver", 0)
    dm = K.DeviceModel(hostsim_lib, mm)
    assert dm.size("features") == 0          # humanoid needs nothing beyond the lean feature set
    states = contact_rich_states(rb, m, 7, seed=23)
    b = K.Batch(dm, len(states))
    b.set_variant(variant)
    assert b.kernel_variant() == variant
    check_forward(rb, m, b, states, tol=0)
    check_forward(rb, m, b, states, tol=0, lds=True)
    fx = golden("humanoid")
    n, T = fx["state0"].shape[0], 40
    b2 = K.Batch(dm, n)
    b2.set_variant(variant)
    out = b2.rollout_host(T, K.mjSTATE_CTRL, fx["state0"], None, fx["ctrl"][:, :T])
    assert np.array_equal(out, fx["state"][:, :T])


def test_lean_variants_refuse_models_they_cannot_step(rb, hostsim_lib, tmp_path):
    """a model that needs a feature outside the lean set keeps the generic kernels"""
    xml = tmp_path / "box.xml"
    xml.write_text(BOX_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    dm = K.DeviceModel(hostsim_lib, m, 64, 200)
    assert dm.size("features") != 0
    b = K.Batch(dm, 2)
    assert b.kernel_variant() == "generic"
    with pytest.raises(K.MjhipError):
        b.set_variant("lean")


def test_step1_step2_split_bit_exact(rb, setup):
    """mj_step1 / mj_step2 (engine_forward.c:1884-1939): closed-loop stepping with a controller that
    reads positions and velocities between the two halves; bit-exact against the oracle's split and
    equal to whole mj_step calls"""
    m, dm = setup
    states = contact_rich_states(rb, m, 3, seed=4)
    b = K.Batch(dm, len(states))
    b.set("qpos", np.stack([s["qpos"] for s in states]))
    b.set("qvel", np.stack([s["qvel"] for s in states]))
    b.set("qacc_warmstart", np.stack([s["qacc_warmstart"] for s in states]))
    ds = []
    for s in states:
        d = rb.MjData(m)
        d.qpos[:] = s["qpos"]; d.qvel[:] = s["qvel"]; d.qacc_warmstart[:] = s["qacc_warmstart"]
        ds.append(d)
    for t in range(6):
        b.step1()
        xpos = b.get("xpos").reshape(len(states), -1, 3)
        qvel = b.get("qvel")
        ctrl = np.clip(-0.5*qvel[:, 6:6 + m.nu] + 0.3*(xpos[:, 1, 2:3] - 1.2), -1, 1)    # needs step1's results
        b.set("ctrl", ctrl)
        b.step2()
        for e, d in enumerate(ds):
            rb.mj_step1(m, d)
            assert np.array_equal(xpos[e], np.array(d.xpos))
            d.ctrl[:] = ctrl[e]
            rb.mj_step2(m, d)
        got_q, got_v = b.get("qpos"), b.get("qvel")
        for e, d in enumerate(ds):
            assert np.array_equal(got_q[e], np.array(d.qpos)) and np.array_equal(got_v[e], np.array(d.qvel)), (t, e)
    assert b.get("warning").sum() == 0


def test_step1_step2_sensors_between_the_halves(rb, hostsim_lib, tmp_path):
    """mj_step1 ends with mj_sensorPos + mj_sensorVel (engine_forward.c:1884-1912), so a controller may
    read position- and velocity-stage sensors between the halves; mj_sensorAcc runs inside mj_step2.
    After step1 the pos/vel sensors hold THIS step's readings and the acc-stage ones still last
    step's -- exactly the reference's sensordata, bit for bit, through a closed loop on a sensor"""
    xml = tmp_path / "sens.xml"
    xml.write_text(SENSOR_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    dm = K.DeviceModel(hostsim_lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[:] = np.random.default_rng(1).normal(0, .5, m.nv)
    b = K.Batch(dm, 1)
    b.set("qpos", np.array(d.qpos)[None]); b.set("qvel", np.array(d.qvel)[None])
    stage = np.array(m.sensor_needstage)
    adr, dim = np.array(m.sensor_adr), np.array(m.sensor_dim)
    assert (stage == 1).any() and (stage == 2).any() and (stage == 3).any()
    for t in range(25):
        rb.mj_step1(m, d)
        b.step1()
        mid_ref, mid = np.array(d.sensordata), b.get("sensordata")[0]
        assert np.array_equal(mid, mid_ref), t          # (acc-stage entries: both still hold the previous step's)
        ctrl = np.clip(0.5*mid[:m.nu] - 0.1, -3, 3)     # the controller reads sensors written by step1
        d.ctrl[:] = ctrl
        b.set("ctrl", ctrl[None])
        rb.mj_step2(m, d)
        b.step2()
        assert np.array_equal(b.get("sensordata")[0], np.array(d.sensordata)), t
        assert np.array_equal(b.get("qpos")[0], np.array(d.qpos)) and np.array_equal(b.get("qvel")[0], np.array(d.qvel)), t
        # the pos/vel readings changed during step1, the acc-stage ones during step2
        for i in np.nonzero(stage == 3)[0][:1]:
            sl = slice(adr[i], adr[i] + dim[i])
            assert np.array_equal(mid_ref[sl], mid[sl])


def _contact_lists_equal(rb, m, b, d, e=0):
    c = b.get("counts")[e]
    assert c[0] == d.ncon, (c[0], d.ncon)
    if d.ncon:
        assert np.array_equal(b.get("con_geom")[e].reshape(-1, 2)[:d.ncon], d.contact["geom"])
        assert np.array_equal(b.get("con_dist")[e][:d.ncon], d.contact["dist"])


def test_broadphase_float_rounding_and_ties_bit_exact(rb, hostsim_lib, tmp_path):
    """exactly touching geoms: the reference's sweep-and-prune compares float-rounded end points and
    breaks ties by array position, so identical touching pairs are or are not tested depending on
    body order; mjhip evaluates the same predicate (stage_broadphase) -- contact lists, not just
    counts, must agree, statically and while the bodies drift apart / together"""
    from parity_utils import TOUCH_XML
    xml = tmp_path / "touch.xml"
    xml.write_text(TOUCH_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    dm = K.DeviceModel(hostsim_lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    b = K.Batch(dm, 1)
    b.forward()
    rb.mj_forward(m, d)
    _contact_lists_equal(rb, m, b, d)
    ncon0 = d.ncon
    # tiny relative velocities: pairs cross the touching configuration in both directions
    rng = np.random.default_rng(2)
    v = np.zeros(m.nv); v[0::6] = rng.normal(0, 1e-3, m.nv // 6); v[1::6] = rng.normal(0, 1e-3, m.nv // 6)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    s0[0, 1 + m.nq:] = v
    T = 40
    ref, ints = oracle_rollout(rb, m, s0, np.zeros((1, T, 0)))
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, np.zeros((1, T, 0)))
    assert np.array_equal(out, ref)
    assert len(set(ints[0, :, 0].tolist()) | {ncon0}) > 1, "the scene is supposed to change its contact count"


def test_multi_geom_bodies_midphase_bit_exact(rb, hostsim_lib, tmp_path):
    """bodies with several geoms take the BVH midphase route: leaf pairs are culled by oriented-box
    tests along a static descent chain and the contacts of a body pair are re-sorted; contact lists
    (geom ids in order, distances) exact at every step of a tumbling pile"""
    from parity_utils import MULTIGEOM_XML
    xml = tmp_path / "multigeom.xml"
    xml.write_text(MULTIGEOM_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    dm = K.DeviceModel(hostsim_lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 120
    ref, ints = oracle_rollout(rb, m, s0, np.zeros((1, T, 0)))
    assert ints[0, :, 0].max() >= 6
    b = K.Batch(dm, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, np.zeros((1, T, 0)))
    assert np.array_equal(out, ref)
    # contact lists along the way
    bb = K.Batch(dm, 1)
    for t in (0, 30, 60, 90, 119):
        st = s0[0] if t == 0 else ref[0, t - 1]
        rb.mj_setState(m, d, st, rb.mjSTATE_FULLPHYSICS)
        rb.mj_forward(m, d)
        bb.set("qpos", st[None, 1:1 + m.nq]); bb.set("qvel", st[None, 1 + m.nq:])
        bb.forward()
        _contact_lists_equal(rb, m, bb, d)


def test_broadphase_emulation_closes_the_touching_gap(rb, hostsim_lib, tmp_path, monkeypatch):
    """random scenes of spheres that touch exactly: with the reference's sweep-and-prune predicate
    reproduced (stage_broadphase) the contact count always matches; with the conservative static
    pair list alone ($MJHIP_BROADPHASE=0) some of these scenes differ -- which is what the stage is for"""
    rng = np.random.default_rng(0)
    d_off = d_on = 0
    for trial in range(150):
        r1, r2 = [float(np.round(rng.uniform(.03, .2), 3)) for _ in range(2)]
        dirv = rng.normal(size=3)
        if trial % 3 == 0:
            dirv = np.eye(3)[trial % 9 // 3]
        dirv /= np.linalg.norm(dirv)
        p1 = np.round(rng.uniform(-1, 1, 3), 2)
        p2 = p1 + dirv*(r1 + r2)
        bodies = [(p1, r1), (p2, r2)]
        if trial % 2:
            bodies = bodies[::-1]
        bodies += [(rng.uniform(-2, 2, 3), .05) for _ in range(rng.integers(0, 3))]
        xml = '<mujoco><option gravity="0 0 0"/><worldbody>' + "".join(
            '<body pos="%.17g %.17g %.17g"><freejoint/><geom type="sphere" size="%g"/></body>' % (*p, r) for p, r in bodies
        ) + "</worldbody></mujoco>"
        f = tmp_path / "t.xml"
        f.write_text(xml)
        m = rb.MjModel.from_xml_path(str(f))
        d = rb.MjData(m)
        rb.mj_forward(m, d)
        for mode in ("1", "0"):
            monkeypatch.setenv("MJHIP_BROADPHASE", mode)
            b = K.Batch(K.DeviceModel(hostsim_lib, m), 1)
            b.forward()
            if int(b.get("counts")[0, 0]) != d.ncon:
                if mode == "1":
                    d_on += 1
                else:
                    d_off += 1
    assert d_on == 0
    assert d_off > 0, "these scenes are supposed to expose the conservative pair list"


def test_emulated_lds_block_faults_beyond_the_launch_allocation(hostsim_lib):
    """profiles/r04/negative_results.txt #6: a variant that kept a scratch array "in the unused tail of the LDS block" was
    bit-exact on the emulation and died on the GPU with HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION -- a flat access beyond
    the workgroup's LDS allocation, which the emulation (then a fixed 160 KB array) let through.  The emulated block now
    ends at an inaccessible page: an access beyond the launch's allocation kills the process, a write into the
    alignment slack trips a canary.  (Probes run in subprocesses: they are supposed to die.)"""
    import subprocess, sys, textwrap
    from conftest import HOSTSIM_LIB
    def probe(lds, offset, write):
        code = textwrap.dedent(f"""
            import ctypes
            lib = ctypes.CDLL({HOSTSIM_LIB!r})
            lib.mjh_test_lds_probe.restype = ctypes.c_int
            print(lib.mjh_test_lds_probe({lds}, {offset}, {write}))
        """)
        return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    ok = probe(10240, 10239, 1)
    assert ok.returncode == 0 and ok.stdout.strip() == "1"
    assert probe(10240, 10240 + 4096, 0).returncode < 0          # read far beyond the allocation: SIGSEGV
    assert probe(10240 + 8, 10240 + 256, 0).returncode < 0       # the first byte behind the (256-byte aligned) block
    assert probe(10240 + 8, 10240 + 100, 1).returncode != 0      # write into the alignment slack: canary abort
