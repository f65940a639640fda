"""GPU (-m gpu): the HIP path through the C ABI against the oracle.

Integer observables -- contact counts, contact geom ids, efc types/ids/addresses, PGS iteration
counts -- must be bit-exact.  Floats: the kernels are compiled without FMA contraction and follow
the reference's operation order, so the only admissible differences come from device libm
(sin/cos/atan2/pow last-ulp); the tolerance written here is the north_star's 1e-6 relative, and the
tests print the much smaller error actually observed."""
import os

import numpy as np
import pytest

from mujoco_amd import _capi as K
import mujoco_amd
from conftest import GOLDEN, contact_rich_states, humanoid_pgs_oracle, many_constraint_states
from parity_utils import CYL_XML, EQ_XML, IMPL_XML, CONDIM_XML, ACT_XML, SENSOR_XML, BOX_XML, BOXBOX_XML, CAPBOX_XML, MOCAP_XML, PAIR_XML, FLUID_XML, ELLIPSOID_FLUID_XML, CAMERA_XML, TENDON_XML, ADHESION_XML, condim_scene_state, chain_xml, many_spheres_xml, check_forward, oracle_rollout, relerr

pytestmark = pytest.mark.gpu
TOL = 1e-6


@pytest.fixture(scope="module")
def dm(hip_lib):
    m = mujoco_amd.MjbModel(hip_lib, os.path.join(GOLDEN, "humanoid.mjb"))
    m.set_option("solver", 0)
    return K.DeviceModel(hip_lib, m)


def test_forward_vs_live_oracle(rb, hip_lib, dm):
    m = humanoid_pgs_oracle(rb)
    states = contact_rich_states(rb, m, 32, seed=11)
    b = K.Batch(dm, len(states))
    worst = check_forward(rb, m, b, states, tol=TOL)
    print("forward: worst relative error over all fields:", worst)


def test_rollout_vs_golden(rb, hip_lib, dm, golden):
    fx = golden("humanoid")
    n, T = fx["state0"].shape[0], fx["ctrl"].shape[1]
    b = K.Batch(dm, n)
    out = b.rollout_host(T, K.mjSTATE_CTRL, fx["state0"], None, fx["ctrl"])
    err_short = relerr(out[:, :10], fx["state"][:, :10])
    print("rollout: rel err first 10 steps", err_short, " full horizon", relerr(out, fx["state"]),
          " bit-exact:", np.array_equal(out, fx["state"]))
    assert err_short <= TOL
    # per-step parity from identical inputs over the whole horizon: 20 restart points per trajectory,
    # one step each from the golden state (cold warm start on both sides), against the live oracle
    s_prev = np.concatenate([fx["state0"][:, None], fx["state"][:, :-1]], axis=1)
    idx = np.linspace(0, T - 1, 20).astype(int)
    s0 = s_prev[:, idx].reshape(-1, s_prev.shape[-1])
    c0 = fx["ctrl"][:, idx].reshape(-1, 1, fx["ctrl"].shape[-1])
    bb = K.Batch(dm, n * 20)
    one = bb.rollout_host(1, K.mjSTATE_CTRL, s0, None, c0)
    ref, ints = oracle_rollout(rb, humanoid_pgs_oracle(rb), s0, c0)
    assert relerr(one, ref) <= TOL
    counts = bb.get("counts")
    assert np.array_equal(counts[:, 0], ints[:, 0, 0]) and np.array_equal(counts[:, 1], ints[:, 0, 1])
    assert np.array_equal(counts[:, 5], ints[:, 0, 2])


@pytest.mark.parametrize("solver", [0, 2, 1], ids=["pgs", "newton", "cg"])
def test_humanoid_rollout_bit_exact_vs_device_sincos_reference(rb, hip_lib, golden, solver):
    """BASELINE config 2's model, whole 120-step rollouts on the device against the compiled reference linked with
    the kernels' own sin / cos (oracle/devmath_shim.cc: liboracle_dm.so; humanoid's step calls no other libm
    function): states bit for bit, contact / row / solver-iteration counts exact at every step -- PGS (the lean
    kernel), Newton and CG (the generic kernel, dense primal path)."""
    fx = golden("humanoid")
    n, T = 8, fx["ctrl"].shape[1]
    m = rb.MjModel.from_binary_path(os.path.join(GOLDEN, "humanoid.mjb"), kind="devmath")
    m.opt.solver = solver
    mm = mujoco_amd.MjbModel(hip_lib, os.path.join(GOLDEN, "humanoid.mjb"))
    mm.set_option("solver", solver)
    dmv = K.DeviceModel(hip_lib, mm)
    s0, ctrl = fx["state0"][:n], fx["ctrl"][:n]
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dmv, n)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    assert b.get("warning").sum() == 0
    nbad = int(np.sum(np.any(out != ref, axis=2)))
    print("solver", solver, "variant", b.kernel_variant(), ": steps not bit-exact", nbad, "of", n*T, " max niter", ints[:, :, 2].max())
    assert nbad == 0
    c = b.get("counts")
    assert np.array_equal(c[:, 0], ints[:, -1, 0]) and np.array_equal(c[:, 1], ints[:, -1, 1]) and np.array_equal(c[:, 5], ints[:, -1, 2])


def test_soa_layout_forward_and_rollout_vs_live_oracle(rb, hip_lib, dm, golden):
    """the SoA-across-environments layout (north_star's coalesced layout: lane-per-environment smooth
    and integrate kernels + the wave-per-environment constraint kernel): every FORWARD field against
    the oracle, and a rollout that equals the environment-major default's bit for bit"""
    m = humanoid_pgs_oracle(rb)
    states = contact_rich_states(rb, m, 70, seed=13)      # > 64: the lane-mode kernels span two wavefronts
    b = K.Batch(dm, len(states), layout="soa")
    worst = check_forward(rb, m, b, states, tol=TOL)
    print("SoA forward worst rel err", worst)
    fx = golden("humanoid")
    n = fx["state0"].shape[0]
    T = 40
    bs = K.Batch(dm, n, layout="soa")
    out_soa = bs.rollout_host(T, K.mjSTATE_CTRL, fx["state0"], None, fx["ctrl"][:, :T])
    ba = K.Batch(dm, n)
    out_aos = ba.rollout_host(T, K.mjSTATE_CTRL, fx["state0"], None, fx["ctrl"][:, :T])
    assert np.array_equal(out_soa, out_aos)
    assert relerr(out_soa[:, :10], fx["state"][:, :10]) <= TOL
    assert np.array_equal(bs.get("counts")[:, :3], ba.get("counts")[:, :3])
    assert bs.get("warning").sum() == 0


def test_pgs_two_constraints_per_lane_vs_live_oracle(rb, hip_lib, dm):
    """64 < nefc <= 128 (solve_pgs_wide: two constraints per lane, AR read from global memory): every
    FORWARD field within tolerance, contact / constraint / PGS iteration counts exact"""
    m = humanoid_pgs_oracle(rb)
    states, nefcs = many_constraint_states(rb, m, 24)
    assert min(nefcs) > 64 and max(nefcs) > 96
    b = K.Batch(dm, len(states))
    worst = check_forward(rb, m, b, states, tol=TOL)
    print("wide PGS forward worst rel err", worst, "nefc", sorted(nefcs))
    assert np.array_equal(b.get("counts")[:, 1], nefcs)


def test_single_step_parity_vs_live_oracle(rb, hip_lib, dm):
    """one mj_step from identical (state, warmstart, ctrl): qpos/qvel within 1e-6 relative, integer
    observables exact (the north_star's parity statement)"""
    m = humanoid_pgs_oracle(rb)
    states = contact_rich_states(rb, m, 64, seed=5)
    n = len(states)
    b = K.Batch(dm, n)
    nstate = 56
    s0 = np.zeros((n, nstate))
    for e, s in enumerate(states):
        s0[e, 0] = s["time"]; s0[e, 1:29] = s["qpos"]; s0[e, 29:] = s["qvel"]
    ws = np.stack([s["qacc_warmstart"] for s in states])
    ctrl = np.stack([s["ctrl"] for s in states])[:, None]
    out = b.rollout_host(1, K.mjSTATE_CTRL, s0, ws, ctrl)[:, 0]
    counts = b.get("counts")
    d = rb.MjData(m)
    worst = 0.0
    for e in range(n):
        rb.mj_resetData(m, d)
        rb.mj_setState(m, d, s0[e], rb.mjSTATE_FULLPHYSICS)
        d.qacc_warmstart[:] = ws[e]
        d.ctrl[:] = ctrl[e, 0]
        rb.mj_step(m, d)
        ref = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
        worst = max(worst, relerr(out[e], ref))
        assert relerr(out[e], ref) <= TOL, (e, relerr(out[e], ref))
        assert counts[e][0] == d.ncon and counts[e][1] == d.nefc and counts[e][5] == d.solver_niter[0]
    print("single step: worst rel err", worst)


def test_rollout_api_matches_serial_oracle(rb, hip_lib):
    """the drop-in `mujoco_amd.rollout.rollout` vs the reference's py_rollout loop (rollout_test.py:976)"""
    from mujoco_amd import rollout
    m = humanoid_pgs_oracle(rb)
    d = rb.MjData(m)
    rng = np.random.default_rng(2)
    nbatch, nstep = 5, 7
    s0 = np.zeros((nbatch, 56))
    for e in range(nbatch):
        rb.mj_resetDataKeyframe(m, d, e % m.nkey)
        s0[e] = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
    ctrl = rng.uniform(-1, 1, size=(nbatch, nstep, m.nu))
    state, sensordata = rollout.rollout(m, d, s0, ctrl)
    ref, _ = oracle_rollout(rb, m, s0, ctrl)
    assert state.shape == (nbatch, nstep, 56) and sensordata.shape == (nbatch, nstep, 0)
    assert relerr(state, ref) <= TOL
    # d holds the last step of the last rollout (rollout.cc:73)
    assert relerr(np.array(d.qpos), ref[-1, -1, 1:29]) <= TOL
    # singleton tiling + nstep inference
    st2, _ = rollout.rollout(m, d, s0[0], ctrl[0:1])
    assert st2.shape == (1, nstep, 56)
    assert relerr(st2[0], ref[0]) <= TOL
    with pytest.raises(ValueError):
        rollout.rollout(m, d, s0[:, :-1], ctrl)


def test_rollout_api_shards_over_real_devices(rb, hip_lib):
    """`mjhip_rollout` on a node with several GPUs: the rollouts of one call are split over the visible
    devices -- one host thread, stream, model / batch cache and lock per device -- and every row comes back from the
    device that owns it.  Checked against the serial reference, with a batch that does not divide by the device count,
    twice (the second call hits the per-device caches).  Skipped on single-GPU boxes; the same code path runs on
    emulated devices in tests/test_rollout_api_cpu.py."""
    ndev = hip_lib.device_count()
    if ndev < 2:
        pytest.skip("needs at least two visible HIP devices")
    from mujoco_amd import rollout
    m = humanoid_pgs_oracle(rb)
    d = rb.MjData(m)
    rng = np.random.default_rng(7)
    nbatch, nstep = 4*ndev + 3, 6
    s0 = np.zeros((nbatch, 56))
    for e in range(nbatch):
        rb.mj_resetDataKeyframe(m, d, e % m.nkey)
        s0[e] = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
        s0[e, 29:] += rng.normal(0, 0.05, size=27)
    ctrl = rng.uniform(-1, 1, size=(nbatch, nstep, m.nu))
    ref, _ = oracle_rollout(rb, m, s0, ctrl)
    for _ in range(2):
        state, _sd = rollout.rollout(m, d, s0, ctrl)
        assert relerr(state, ref) <= TOL
        assert relerr(np.array(d.qpos), ref[-1, -1, 1:29]) <= TOL


def test_rk4_rollout_vs_live_oracle(rb, hip_lib, golden):
    """RK4 integrator (mj_RungeKutta, engine_forward.c:1486-1587) on the GPU against the oracle"""
    m = humanoid_pgs_oracle(rb)
    m.opt.integrator = 1
    mm = mujoco_amd.MjbModel(hip_lib, os.path.join(GOLDEN, "humanoid.mjb"))
    mm.set_option("solver", 0)
    mm.set_option("integrator", 1)
    dmr = K.DeviceModel(hip_lib, mm)
    fx = golden("humanoid")
    n, T = fx["state0"].shape[0], 10
    ref, ints = oracle_rollout(rb, m, fx["state0"], fx["ctrl"][:, :T])
    b = K.Batch(dmr, n)
    out = b.rollout_host(T, K.mjSTATE_CTRL, fx["state0"], None, fx["ctrl"][:, :T])
    print("rk4 rollout rel err", relerr(out, ref))
    assert relerr(out, ref) <= TOL
    assert np.array_equal(b.get("counts")[:, 1], ints[:, -1, 1])


def test_slider_crank_vs_golden(hip_lib, golden):
    """BASELINE config 1: slider-crank transmissions + position actuators (no contacts in this regime)"""
    mm = mujoco_amd.MjbModel(hip_lib, os.path.join(GOLDEN, "slider_crank.mjb"))
    mm.set_option("solver", 0)
    dmc = K.DeviceModel(hip_lib, mm)
    fx = golden("slider_crank")
    n, T = fx["state0"].shape[0], fx["ctrl"].shape[1]
    b = K.Batch(dmc, n)
    out = b.rollout_host(T, K.mjSTATE_CTRL, fx["state0"], None, fx["ctrl"])
    print("slider_crank rollout rel err", relerr(out, fx["state"]))
    assert relerr(out, fx["state"]) <= TOL
    assert b.get("warning").sum() == 0


def test_cylinder_scene_islands_vs_live_oracle(rb, hip_lib, tmp_path):
    """plane-cylinder collider (up to 4 contacts) and per-island PGS on a 4-tree scene"""
    xml = tmp_path / "cyl.xml"
    xml.write_text(CYL_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.solver = 0
    dmc = K.DeviceModel(hip_lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 60
    ctrl = np.zeros((1, T, 0))
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dmc, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    print("cylinder scene rel err", relerr(out, ref), "max ncon", ints[0, :, 0].max())
    assert relerr(out, ref) <= TOL
    c = b.get("counts")[0]
    assert c[0] == ints[0, -1, 0] and c[1] == ints[0, -1, 1] and c[5] == ints[0, -1, 2]


@pytest.mark.parametrize("solver", [0, 2])
def test_equality_constraints_vs_live_oracle(rb, hip_lib, tmp_path, solver):
    """connect / weld / joint / tendon equalities mixed with contacts, limits and friction loss"""
    xml = tmp_path / "eq.xml"
    xml.write_text(EQ_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.solver = solver
    dme = K.DeviceModel(hip_lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 80
    ctrl = np.random.default_rng(5).uniform(-1, 1, (1, T, m.nu))
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dme, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    print("equality scene solver", solver, "rel err", relerr(out, ref), "max nefc", ints[0, :, 1].max())
    assert relerr(out, ref) <= TOL
    assert b.get("warning").sum() == 0
    c = b.get("counts")[0]
    assert c[0] == ints[0, -1, 0] and c[1] == ints[0, -1, 1]


@pytest.mark.parametrize("integrator", [3, 2])
@pytest.mark.parametrize("solver", [0, 2])
def test_implicitfast_vs_live_oracle(rb, hip_lib, tmp_path, solver, integrator):
    """implicitfast: velocity-dependent actuators, tendon damping, standalone free bodies; round 6: the fully
    implicit integrator on the same scene (mjd_rne_vel, sparse LU)"""
    xml = tmp_path / "impl.xml"
    xml.write_text(IMPL_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.solver = solver
    m.opt.integrator = integrator
    dmi = K.DeviceModel(hip_lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    # (three environments with their own initial velocities and controls: every per-environment address is exercised)
    NE = 3
    s0 = np.zeros((NE, rb.mj_stateSize(m, rb.mjSTATE_FULLPHYSICS)))
    for k in range(NE):
        d.qvel[:] = np.random.default_rng(1 + k).normal(0, 1.0, m.nv)
        s0[k] = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
    T = 100
    ctrl = np.random.default_rng(0).uniform(-1, 1, (NE, T, m.nu))
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dmi, NE)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    print("implicitfast scene solver", solver, "rel err", relerr(out, ref), "max nefc", ints[0, :, 1].max())
    assert relerr(out, ref) <= TOL
    assert b.get("warning").sum() == 0


@pytest.mark.parametrize("cone,solver", [(0, 0), (1, 0), (1, 2), (1, 1)])
def test_condim_and_elliptic_cones_vs_live_oracle(rb, hip_lib, tmp_path, cone, solver):
    """condim 4/6 contacts, pyramidal and elliptic cones, PGS / Newton / CG"""
    xml = tmp_path / "condim.xml"
    xml.write_text(CONDIM_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.cone = cone
    m.opt.solver = solver
    dmc = K.DeviceModel(hip_lib, m)
    s0 = condim_scene_state(rb, m)
    T = 60
    ctrl = np.zeros((1, T, 0))
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dmc, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    print("condim scene cone", cone, "solver", solver, "rel err", relerr(out, ref), "max nefc", ints[0, :, 1].max())
    assert relerr(out, ref) <= TOL
    assert b.get("warning").sum() == 0


def test_elliptic_humanoid_vs_live_oracle(rb, hip_lib, golden):
    """humanoid, cone=elliptic, PGS and Newton, the contact-rich golden environments"""
    fx = golden("humanoid")
    n, T = fx["state0"].shape[0], 40
    for solver in (0, 2):
        m = rb.MjModel.from_binary_path(os.path.join(GOLDEN, "humanoid.mjb"))
        m.opt.solver = solver
        m.opt.cone = 1
        dme = K.DeviceModel(hip_lib, m)
        ref, ints = oracle_rollout(rb, m, fx["state0"][:n], fx["ctrl"][:n, :T])
        b = K.Batch(dme, n)
        out = b.rollout_host(T, K.mjSTATE_CTRL, fx["state0"][:n], None, fx["ctrl"][:n, :T])
        print("elliptic humanoid solver", solver, "rel err", relerr(out, ref), "max nefc", ints[:, :, 1].max())
        assert relerr(out, ref) <= TOL
        assert b.get("warning").sum() == 0


@pytest.mark.parametrize("solver", [0, 2])
def test_sparse_jacobian_model_vs_live_oracle(rb, hip_lib, tmp_path, solver):
    """67-dof chain, jacobian=auto (sparse paths in the reference), up to ~230 constraint rows"""
    xml = tmp_path / "chain.xml"
    xml.write_text(chain_xml(62).replace('jacobian="dense"', 'jacobian="auto"'))
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.solver = solver
    dms = K.DeviceModel(hip_lib, m, 80, 300)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[:] = np.random.default_rng(1).normal(0, .5, m.nv)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 170
    ctrl = np.random.default_rng(0).uniform(-1, 1, (1, T, m.nu))
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dms, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    print("sparse-jacobian chain solver", solver, "rel err", relerr(out, ref), "max nefc", ints[0, :, 1].max())
    assert relerr(out, ref) <= TOL
    assert b.get("warning").sum() == 0


@pytest.mark.parametrize("integrator", [0, 1, 3])
def test_stateful_actuators_vs_live_oracle(rb, hip_lib, tmp_path, integrator):
    """filter / filterexact / integrator actuator dynamics under Euler, RK4 and implicitfast"""
    xml = tmp_path / "act.xml"
    xml.write_text(ACT_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.integrator = integrator
    dma = K.DeviceModel(hip_lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[:] = np.random.default_rng(1).normal(0, .5, m.nv)
    d.act[:] = np.random.default_rng(2).normal(0, .2, m.na)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 150
    ctrl = np.random.default_rng(0).uniform(-1.5, 1.5, (1, T, m.nu))
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dma, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    print("stateful actuators integrator", integrator, "rel err", relerr(out, ref))
    assert relerr(out, ref) <= TOL
    assert b.get("warning").sum() == 0


@pytest.mark.parametrize("cone", [0, 1])
def test_box_and_cylinder_colliders_vs_live_oracle(rb, hip_lib, tmp_path, cone):
    """plane-box, sphere-box, sphere-cylinder colliders; up to 19 simultaneous contacts"""
    xml = tmp_path / "box.xml"
    xml.write_text(BOX_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.cone = cone
    dmb = K.DeviceModel(hip_lib, m, 64, 200)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[:] = np.random.default_rng(4).normal(0, .3, m.nv)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 200
    ctrl = np.zeros((1, T, 0))
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dmb, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    print("box scene cone", cone, "rel err", relerr(out, ref), "max ncon", ints[0, :, 0].max())
    assert relerr(out, ref) <= TOL
    assert b.get("warning").sum() == 0


@pytest.mark.parametrize("cone", [0, 1])
def test_box_box_collider_vs_live_oracle(rb, hip_lib, tmp_path, cone):
    """box-box stacks and crossings, up to 29 simultaneous contacts"""
    xml = tmp_path / "boxbox.xml"
    xml.write_text(BOXBOX_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.cone = cone
    dmb = K.DeviceModel(hip_lib, m, 64, 200)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[:] = np.random.default_rng(4).normal(0, .3, m.nv)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 250
    ctrl = np.zeros((1, T, 0))
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dmb, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    print("box-box scene cone", cone, "rel err", relerr(out, ref), "max ncon", ints[0, :, 0].max())
    assert relerr(out, ref) <= TOL
    assert b.get("warning").sum() == 0


@pytest.mark.parametrize("cone", [0, 1])
def test_capsule_box_collider_vs_live_oracle(rb, hip_lib, tmp_path, cone):
    """capsules lying on, standing on, hanging over and crossing boxes (mjc_CapsuleBox)"""
    xml = tmp_path / "capbox.xml"
    xml.write_text(CAPBOX_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.cone = cone
    dmb = K.DeviceModel(hip_lib, m, 64, 200)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[:] = np.random.default_rng(4).normal(0, .3, m.nv)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 250
    ctrl = np.zeros((1, T, 0))
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dmb, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    print("capsule-box scene cone", cone, "rel err", relerr(out, ref), "max ncon", ints[0, :, 0].max())
    assert relerr(out, ref) <= TOL
    assert b.get("warning").sum() == 0
    c = b.get("counts")[0]
    assert c[0] == ints[0, -1, 0] and c[1] == ints[0, -1, 1]


def test_many_filter_survivors_and_wide_pgs_vs_live_oracle(rb, hip_lib, tmp_path):
    """90 spheres over a plane (nv = 540, generic kernel): > 64 filter survivors, up to 120 rows"""
    xml = tmp_path / "many.xml"
    xml.write_text(many_spheres_xml())
    m = rb.MjModel.from_xml_path(str(xml))
    dmb = K.DeviceModel(hip_lib, m, 128, 512)       # 90 contacts x 4 rows in the end
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 40
    ctrl = np.zeros((1, T, 0))
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dmb, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    print("many spheres rel err", relerr(out, ref), "max ncon", ints[0, :, 0].max(), "max nefc", ints[0, :, 1].max())
    assert relerr(out, ref) <= TOL
    c = b.get("counts")[0]
    assert c[0] == ints[0, -1, 0] and c[1] == ints[0, -1, 1]
    assert b.get("warning").sum() == 0


def test_mocap_bodies_vs_live_oracle(rb, hip_lib, tmp_path):
    """mocap poses supplied through the control array (mjSTATE_MOCAP_POS | mjSTATE_MOCAP_QUAT)"""
    from test_hostsim_parity import _mocap_controls
    xml = tmp_path / "mocap.xml"
    xml.write_text(MOCAP_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    dmm = K.DeviceModel(hip_lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 200
    spec = K.mjSTATE_CTRL | K.mjSTATE_MOCAP_POS | K.mjSTATE_MOCAP_QUAT
    ctrl = _mocap_controls(m, T)
    ref = np.zeros((1, T, s0.shape[1]))
    for t in range(T):
        rb.mj_setState(m, d, ctrl[0, t], spec)
        rb.mj_step(m, d)
        ref[0, t] = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
    b = K.Batch(dmm, 1)
    out = b.rollout_host(T, spec, s0, None, ctrl)
    print("mocap scene rel err", relerr(out, ref))
    assert relerr(out, ref) <= TOL
    assert b.get("warning").sum() == 0


@pytest.mark.parametrize("cone", [0, 1])
def test_predefined_contact_pairs_vs_live_oracle(rb, hip_lib, tmp_path, cone):
    """predefined contact pairs (own parameters, solreffriction) and an exclude"""
    xml = tmp_path / "pair.xml"
    xml.write_text(PAIR_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.cone = cone
    dmp = K.DeviceModel(hip_lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[:] = np.random.default_rng(2).normal(0, .6, m.nv)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 200
    ctrl = np.zeros((1, T, 0))
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dmp, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    print("pair scene cone", cone, "rel err", relerr(out, ref), "max ncon", ints[0, :, 0].max())
    assert relerr(out, ref) <= TOL


@pytest.mark.parametrize("cone,solver,integrator", [(0, 0, 0), (1, 0, 0), (0, 2, 2), (1, 2, 3)])
def test_contact_adhesion_vs_live_oracle(rb, hip_lib, tmp_path, cone, solver, integrator):
    """geom / pair adhesion: adhesive contacts active in the gap (one frictionless row), the attraction in qfrc_passive
    (mj_adhesion), the rows' biased reference acceleration (mj_adhesionRef); a magnetic door catch and adhesive geoms"""
    xml = tmp_path / "adh.xml"
    xml.write_text(ADHESION_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.cone = cone
    m.opt.solver = solver
    m.opt.integrator = integrator
    dma = K.DeviceModel(hip_lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[0] = -1.0
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 200
    ctrl = np.random.default_rng(5).uniform(0, 1, (1, T, m.nu))      # adhesion actuators (body transmission)
    ref, ints = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dma, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    print("adhesion scene cone", cone, "solver", solver, "integrator", integrator, "rel err", relerr(out, ref), "max ncon", ints[0, :, 0].max())
    assert relerr(out, ref) <= TOL
    assert b.get("warning").sum() == 0
    if solver == 0:
        rb.mj_resetData(m, d)
        d.qvel[0] = -1.0
        for stop in (1, 25, 60):
            while round(d.time / m.opt.timestep) < stop:
                d.ctrl[:] = ctrl[0, int(round(d.time / m.opt.timestep))]
                rb.mj_step(m, d)
            st = [dict(qpos=np.array(d.qpos), qvel=np.array(d.qvel), qacc_warmstart=np.array(d.qacc_warmstart), ctrl=np.array(d.ctrl))]
            check_forward(rb, m, b, st, tol=TOL)


@pytest.mark.parametrize("integrator", [1, 2, 3])
def test_fluid_forces_vs_live_oracle(rb, hip_lib, tmp_path, integrator):
    """inertia-box fluid forces with wind: RK4, and the implicit integrators (the forces' velocity derivative in qDeriv)"""
    xml = tmp_path / "fluid.xml"
    xml.write_text(FLUID_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.integrator = integrator
    dmf = K.DeviceModel(hip_lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[:] = np.random.default_rng(1).normal(0, 1.0, m.nv)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 200
    ctrl = np.random.default_rng(0).uniform(-1, 1, (1, T, m.nu))
    ref, _ = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dmf, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    print("fluid scene rel err", relerr(out, ref))
    assert relerr(out, ref) <= TOL


def test_camera_sensors_vs_live_oracle(rb, hip_lib, tmp_path):
    """cameras in every mj_camlight mode, frame sensors attached / referenced to them, camprojection: sensordata every step"""
    xml = tmp_path / "cam.xml"
    xml.write_text(CAMERA_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    dmc = K.DeviceModel(hip_lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[:] = np.random.default_rng(1).normal(0, .5, m.nv)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 200
    ctrl = np.random.default_rng(0).uniform(-3, 3, (1, T, m.nu))
    ref = np.zeros((1, T, s0.shape[1])); sref = np.zeros((1, T, m.nsensordata))
    rb.mj_setState(m, d, s0[0], rb.mjSTATE_FULLPHYSICS)
    for t in range(T):
        d.ctrl[:] = ctrl[0, t]
        rb.mj_step(m, d)
        ref[0, t] = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
        sref[0, t] = d.sensordata
    b = K.Batch(dmc, 1)
    out, sd = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl, want_sensordata=True)
    print("camera scene rel err state", relerr(out, ref), "sensordata", relerr(sd, sref))
    assert relerr(out, ref) <= TOL and relerr(sd, sref) <= TOL


@pytest.mark.parametrize("integrator", [0, 2, 3])
def test_ellipsoid_fluid_model_vs_live_oracle(rb, hip_lib, tmp_path, integrator):
    """the ellipsoid fluid model per geom (added mass, lift, drag) and its velocity derivative under the implicit integrators"""
    xml = tmp_path / "efluid.xml"
    xml.write_text(ELLIPSOID_FLUID_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.integrator = integrator
    dmf = K.DeviceModel(hip_lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    # (three environments with their own initial velocities and controls: every per-environment address is exercised)
    NE = 3
    s0 = np.zeros((NE, rb.mj_stateSize(m, rb.mjSTATE_FULLPHYSICS)))
    for k in range(NE):
        d.qvel[:] = np.random.default_rng(1 + k).normal(0, 1.0, m.nv)
        s0[k] = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
    T = 150
    ctrl = np.random.default_rng(0).uniform(-1, 1, (NE, T, m.nu))
    ref, _ = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dmf, NE)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    print("ellipsoid fluid scene integrator", integrator, "rel err", relerr(out, ref))
    assert relerr(out, ref) <= TOL


@pytest.mark.parametrize("solver", [0, 2])
def test_spatial_tendons_vs_live_oracle(rb, hip_lib, tmp_path, solver):
    """spatial tendons through sites with pulleys, springs, a limit and a tendon equality"""
    xml = tmp_path / "tendon.xml"
    xml.write_text(TENDON_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.solver = solver
    dmt = K.DeviceModel(hip_lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[:] = np.random.default_rng(1).normal(0, .5, m.nv)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 200
    ctrl = np.random.default_rng(0).uniform(-1, 1, (1, T, m.nu))
    ref, _ = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dmt, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    print("spatial tendon scene solver", solver, "rel err", relerr(out, ref))
    assert relerr(out, ref) <= TOL


@pytest.mark.parametrize("jacobian,integrator", [(0, 0), (1, 1), (1, 3)])
def test_tendon_armature_vs_live_oracle(rb, hip_lib, tmp_path, jacobian, integrator):
    """tendon armature on spatial / pulley / fixed tendons and as an actuator's armature: inertia term, bias force with the
    time derivative of the tendon Jacobian (tests/test_hostsim_parity.py::test_tendon_armature_bit_exact is the bit-exact
    statement; the device evaluates its own sin / cos in the kinematics)"""
    import test_hostsim_parity as th
    out, ref, eM, eb = th._tendon_armature(rb, hip_lib, tmp_path, jacobian, integrator)
    print("tendon armature: rel err", relerr(out, ref), "M", eM, "bias", eb)
    assert relerr(out, ref) <= TOL and eM <= 1e-12 and eb <= 1e-9


@pytest.mark.parametrize("integrator", [0, 3])
def test_tendon_wrapping_vs_live_oracle(rb, hip_lib, tmp_path, integrator):
    """tendons wrapping around spheres / cylinders (mjh_math.h: mjh_wrap) on the device, where acos / asin go through
    mjh_atan2: trajectory within the 1e-6 bar of the reference as built, lengths and moments to rounding"""
    import test_hostsim_parity as th
    out, ref, eL, eJ, nwrap = th._tendon_wrap(rb, hip_lib, tmp_path, integrator)
    print("tendon wrapping: rel err", relerr(out, ref), "length", eL, "moment", eJ, "wrap points", nwrap)
    assert nwrap > 10 and relerr(out, ref) <= TOL and eL <= 1e-12 and eJ <= 1e-10


@pytest.mark.parametrize("integrator", [0, 3])
def test_disabled_actuator_groups_and_tendon_force_limits_vs_live_oracle(rb, hip_lib, tmp_path, integrator):
    """opt.disableactuator groups (no force, frozen activation) and tendon-level limits on the total actuator force"""
    import test_hostsim_parity as th
    out, ref, eF, clamped = th._actuator_groups(rb, hip_lib, tmp_path, integrator)
    print("actuator groups / tendon force limits: rel err", relerr(out, ref), "force", eF, "steps at a limit", clamped)
    assert clamped >= 2 and relerr(out, ref) <= TOL and eF <= 1e-10


@pytest.mark.parametrize("integrator", [0, 3])
def test_muscle_actuators_vs_live_oracle(rb, hip_lib, tmp_path, integrator):
    """muscle dynamics / gain / bias on wrapping and fixed tendons and on a joint (arm26.xml's actuators)"""
    import test_hostsim_parity as th
    out, ref, eF, fmax = th._muscles(rb, hip_lib, tmp_path, integrator)
    print("muscles: rel err", relerr(out, ref), "force / act_dot", eF, "largest force", fmax)
    assert fmax > 5.0 and relerr(out, ref) <= TOL and eF <= 1e-9


@pytest.mark.parametrize("scene", ["SITE_ACT_XML", "BALL_ACT_XML"])
def test_cartesian_and_ball_actuators_vs_live_oracle(rb, hip_lib, tmp_path, scene):
    """site transmissions and actuators on ball / free joints, implicitfast (their moments enter qDeriv)"""
    import parity_utils
    xml = tmp_path / "scene.xml"
    xml.write_text(getattr(parity_utils, scene))
    m = rb.MjModel.from_xml_path(str(xml))
    m.opt.integrator = 3
    dmx = K.DeviceModel(hip_lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[:] = np.random.default_rng(1).normal(0, .5, m.nv)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 150
    ctrl = np.random.default_rng(0).uniform(-.5, .5, (1, T, m.nu))
    ref, _ = oracle_rollout(rb, m, s0, ctrl)
    b = K.Batch(dmx, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    print(scene, "rel err", relerr(out, ref))
    assert relerr(out, ref) <= TOL


def test_sensors_vs_live_oracle(rb, hip_lib, tmp_path):
    """sensordata of every rollout step: 113 readings of 45 sensors incl. IMU / force / torque"""
    xml = tmp_path / "sens.xml"
    xml.write_text(SENSOR_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    dmx = K.DeviceModel(hip_lib, m)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[:] = np.random.default_rng(1).normal(0, .5, m.nv)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 300
    ctrl = np.random.default_rng(0).uniform(-3, 3, (1, T, m.nu))
    sref = np.zeros((1, T, m.nsensordata))
    ref = np.zeros((1, T, s0.shape[1]))
    for t in range(T):
        d.ctrl[:] = ctrl[0, t]
        rb.mj_step(m, d)
        ref[0, t] = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
        sref[0, t] = d.sensordata
    b = K.Batch(dmx, 1)
    out, sd = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl, want_sensordata=True)
    print("sensor scene: state rel err", relerr(out, ref), "sensordata rel err", relerr(sd, sref))
    assert relerr(out, ref) <= TOL
    assert relerr(sd, sref) <= TOL


@pytest.mark.parametrize("scene", ["sensor", "boxbox", "equality"])
def test_batched_feature_scenes_vs_live_oracle(rb, hip_lib, tmp_path, scene):
    """many environments of the feature scenes at once (different initial velocities and controls
    per environment): per-environment indexing of every new field, sensordata included"""
    xml = tmp_path / "scene.xml"
    xml.write_text({"sensor": SENSOR_XML, "boxbox": BOXBOX_XML, "equality": EQ_XML}[scene])
    m = rb.MjModel.from_xml_path(str(xml))
    dmx = K.DeviceModel(hip_lib, m, 64, 200)
    nenv, T = 48, 40
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    s0 = np.tile(rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS), (nenv, 1))
    rng = np.random.default_rng(21)
    s0[:, 1 + m.nq:1 + m.nq + m.nv] = rng.normal(0, .4, (nenv, m.nv))
    ctrl = rng.uniform(-2, 2, (nenv, T, m.nu))
    ref = np.zeros((nenv, T, s0.shape[1]))
    sref = np.zeros((nenv, T, m.nsensordata))
    for e in range(nenv):
        rb.mj_resetData(m, d)
        rb.mj_setState(m, d, s0[e], rb.mjSTATE_FULLPHYSICS)
        for t in range(T):
            d.ctrl[:] = ctrl[e, t]
            rb.mj_step(m, d)
            ref[e, t] = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
            sref[e, t] = d.sensordata
    b = K.Batch(dmx, nenv)
    out, sd = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl, want_sensordata=True)
    print("batched", scene, "state rel err", relerr(out, ref), "sensordata rel err", relerr(sd, sref) if sref.size else 0.0)
    assert relerr(out, ref) <= TOL
    if sref.size:
        assert relerr(sd, sref) <= TOL
    assert b.get("warning").sum() == 0


def test_newton_solver_vs_live_oracle(rb, hip_lib, golden):
    """humanoid with the reference's default options (Newton solver) on the GPU"""
    m = rb.MjModel.from_binary_path(os.path.join(GOLDEN, "humanoid.mjb"))
    mm = mujoco_amd.MjbModel(hip_lib, os.path.join(GOLDEN, "humanoid.mjb"))
    dmn = K.DeviceModel(hip_lib, mm)
    fx = golden("humanoid")
    n, T = fx["state0"].shape[0], 40
    ref, ints = oracle_rollout(rb, m, fx["state0"], fx["ctrl"][:, :T])
    b = K.Batch(dmn, n)
    out = b.rollout_host(T, K.mjSTATE_CTRL, fx["state0"], None, fx["ctrl"][:, :T])
    print("newton rollout rel err", relerr(out, ref), " bit-exact:", np.array_equal(out, ref))
    assert relerr(out, ref) <= TOL
    c = b.get("counts")
    assert np.array_equal(c[:, 0], ints[:, -1, 0]) and np.array_equal(c[:, 1], ints[:, -1, 1])
    # the solver is an operation-for-operation restatement (mjh_newton.h): from identical (state, warm
    # start, ctrl) the Newton iteration count of every step is the reference's, pyramidal and elliptic
    for cone in (0, 1):
        m.opt.cone = cone
        mm.set_option("cone", cone)
        dmc = K.DeviceModel(hip_lib, mm)
        d = rb.MjData(m)
        S, W, C, R, NI = [], [], [], [], []
        for e in range(n):
            rb.mj_resetData(m, d)
            rb.mj_setState(m, d, fx["state0"][e], rb.mjSTATE_FULLPHYSICS)
            for t in range(T):
                S.append(rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)); W.append(np.array(d.qacc_warmstart)); C.append(fx["ctrl"][e, t])
                d.ctrl[:] = fx["ctrl"][e, t]
                rb.mj_step(m, d)
                R.append(rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)); NI.append((d.ncon, d.nefc, d.solver_niter[0]))
        bb = K.Batch(dmc, len(S))
        one = bb.rollout_host(1, K.mjSTATE_CTRL, np.array(S), np.array(W), np.array(C)[:, None])[:, 0]
        cc = bb.get("counts")
        NI = np.array(NI)
        assert relerr(one, np.array(R)) <= TOL
        assert np.array_equal(cc[:, 0], NI[:, 0]) and np.array_equal(cc[:, 1], NI[:, 1])
        assert np.array_equal(cc[:, 5], NI[:, 2]), (cone, np.nonzero(cc[:, 5] != NI[:, 2])[0][:10])
        print("newton cone", cone, "single steps", len(S), "rel err", relerr(one, np.array(R)), "solver_niter exact, max", NI[:, 2].max())


def test_full_size_batch_properties(hip_lib, dm, golden):
    """BASELINE size (4096 envs): size-independent properties -- replicated envs give identical
    bits, an env's trajectory does not depend on its position in the batch, no warnings."""
    fx = golden("humanoid")
    n, T = 4096, 20
    rep = np.arange(n) % fx["state0"].shape[0]
    s0 = fx["state0"][rep]
    ctrl = fx["ctrl"][rep][:, :T]
    b = K.Batch(dm, n)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    k = fx["state0"].shape[0]
    for e in range(k):
        assert np.array_equal(out[e::k], np.broadcast_to(out[e], out[e::k].shape)), "replicas differ"
    assert relerr(out[:k, :10], fx["state"][:, :10]) <= TOL
    assert b.get("warning").sum() == 0
    perm = np.random.default_rng(0).permutation(n)
    out2 = b.rollout_host(T, K.mjSTATE_CTRL, s0[perm], None, ctrl[perm])
    assert np.array_equal(out2, out[perm])


def test_closed_loop_step_matches_rollout(hip_lib, dm, golden):
    fx = golden("humanoid")
    n = fx["state0"].shape[0]
    b = K.Batch(dm, n)
    ref = b.rollout_host(3, K.mjSTATE_CTRL, fx["state0"], None, fx["ctrl"][:, :3])
    b.reset()
    b.set("time", fx["state0"][:, :1]); b.set("qpos", fx["state0"][:, 1:29]); b.set("qvel", fx["state0"][:, 29:])
    for t in range(3):
        b.set("ctrl", fx["ctrl"][:, t])
        b.step(1)
    assert np.array_equal(b.get("qpos"), ref[:, 2, 1:29])
    assert np.array_equal(b.get("qvel"), ref[:, 2, 29:])


@pytest.mark.parametrize("variant", ["generic", "lean"])
def test_kernel_variants_field_parity(rb, hip_lib, dm, variant):
    """both feature sets of the stage sources (generic or lean: mjh_modes.h) against the live oracle,
    field by field, on contact-rich states"""
    m = humanoid_pgs_oracle(rb)
    states = contact_rich_states(rb, m, 35, seed=23)       # odd count: the last wavefront is partly empty
    b = K.Batch(dm, len(states))
    b.set_variant(variant)
    assert b.kernel_variant() == variant
    worst = check_forward(rb, m, b, states, tol=TOL)
    worst_lds = check_forward(rb, m, b, states, tol=TOL, lds=True)
    print(variant, "forward worst rel err", worst, "on the LDS plan", worst_lds)


def test_kernel_variants_bit_identical_rollouts(hip_lib, dm, golden):
    """the variants are the same arithmetic in a different lane mapping: identical bits, 120 steps"""
    fx = golden("humanoid")
    n, T = fx["state0"].shape[0], fx["ctrl"].shape[1]
    outs = {}
    for variant in ["generic", "lean"]:
        b = K.Batch(dm, n)
        b.set_variant(variant)
        outs[variant] = b.rollout_host(T, K.mjSTATE_CTRL, fx["state0"], None, fx["ctrl"])
        assert b.get("warning").sum() == 0
    assert np.array_equal(outs["lean"], outs["generic"])
    assert relerr(outs["lean"][:, :10], fx["state"][:, :10]) <= TOL


# ---- warning semantics on the GPU path (rollout.cc:135-155; engine_forward.c:54-113) -------------------

def test_bad_state_freezes_and_resets_on_gpu(rb, hip_lib, dm):
    """a bad qvel raises mjWARN_BADQVEL, auto-resets the environment (mj_checkVel) and the rollout
    loop back-fills the rest of its trajectory; the other environments of the batch are unaffected"""
    m = humanoid_pgs_oracle(rb)
    d = rb.MjData(m)
    base = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
    rng = np.random.default_rng(3)
    n, T = 6, 6
    s0 = np.tile(base, (n, 1))
    s0[:, 29:] = rng.normal(0, .1, size=(n, m.nv))
    s0[2, 1 + 28 + 3] = 1e12            # bad qvel
    s0[5, 1 + 4] = np.nan               # bad qpos
    ctrl = rng.uniform(-1, 1, size=(n, T, m.nu))
    ref, _ = oracle_rollout(rb, m, s0, ctrl)        # (a plain mj_step loop: no freeze rule)
    ok = [0, 1, 3, 4]
    for variant in ["lean", "generic"]:
        b = K.Batch(dm, n)
        b.set_variant(variant)
        out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
        assert relerr(out[ok], ref[ok]) <= TOL, variant
        assert relerr(out[[2, 5], 0], ref[[2, 5], 0]) <= TOL, variant      # the step that raised the warning
        for e in (2, 5):
            for t in range(1, T):
                assert np.array_equal(out[e, t], out[e, 0]), (variant, e, t)
        w = b.get("warning")
        assert w[2, 4] == 1 and w[5, 3] == 1 and w[[0, 1, 3, 4]].sum() == 0, variant


def test_capacity_overflow_on_gpu(rb, hip_lib):
    """too small a contact / constraint capacity raises mjWARN_CONTACTFULL / mjWARN_CNSTRFULL (the
    reference's full-arena warnings, engine_collision_driver.c:2028, engine_core_constraint.c:145),
    the environment freezes, and mjhip_batch_trouble reports it"""
    m = humanoid_pgs_oracle(rb)
    d = rb.MjData(m)
    rb.mj_resetDataKeyframe(m, d, 2)                # prone: a dozen contacts
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None]
    for caps, slot in (((2, 0), 1), ((0, 8), 2)):
        dmc = K.DeviceModel(hip_lib, m, *caps)
        b = K.Batch(dmc, 1)
        out = b.rollout_host(4, K.mjSTATE_CTRL, s0, None, np.zeros((1, 4, m.nu)))
        w = b.get("warning")[0]
        assert w[slot] >= 1, (caps, w)
        assert np.array_equal(out[0, 1], out[0, 0]) and np.array_equal(out[0, 3], out[0, 0])
        assert b.trouble()[0] >= 1


def test_rollout_full_horizon_per_step_parity(rb, hip_lib, dm, golden):
    """all 120 steps of the golden trajectories, asserted: every step restarted from the oracle's
    previous state and running warm start (so a chaotic trajectory cannot hide or amplify anything)"""
    fx = golden("humanoid")
    m = humanoid_pgs_oracle(rb)
    n, T = fx["state0"].shape[0], fx["ctrl"].shape[1]
    d = rb.MjData(m)
    s_in = np.zeros((n, T, 56)); ws_in = np.zeros((n, T, m.nv))
    for e in range(n):
        rb.mj_resetData(m, d)
        rb.mj_setState(m, d, fx["state0"][e], rb.mjSTATE_FULLPHYSICS)
        for t in range(T):
            s_in[e, t] = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
            ws_in[e, t] = np.array(d.qacc_warmstart)
            d.ctrl[:] = fx["ctrl"][e, t]
            rb.mj_step(m, d)
            assert np.array_equal(rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS), fx["state"][e, t])
    b = K.Batch(dm, n*T)
    one = b.rollout_host(1, K.mjSTATE_CTRL, s_in.reshape(n*T, 56), ws_in.reshape(n*T, m.nv),
                         fx["ctrl"].reshape(n*T, 1, -1))[:, 0].reshape(n, T, 56)
    err = relerr(one, fx["state"])
    print("per-step parity over the full 120-step horizon:", err)
    assert err <= TOL


def test_rollout_api_user_inputs_and_multi_model_on_gpu(rb, hip_lib, tmp_path):
    """the drop-in on the GPU: mjSTATE_USER control spec (xfrc_applied, eq_active, mocap pose,
    userdata) and one model per rollout (rollout_test.py:363 test_multi_model)"""
    from mujoco_amd import rollout
    from test_rollout_api_cpu import USER_XML, _py_rollout
    models = []
    for i in range(3):
        xml = tmp_path / f"m{i}.xml"
        xml.write_text(USER_XML.replace('name="hand" pos="0 0 .35"', f'name="hand" pos="{.1*i} 0 {.35 + .05*i}"'))
        models.append(rb.MjModel.from_xml_path(str(xml)))
    m = models[0]
    d = rb.MjData(m)
    rng = np.random.default_rng(5)
    order = [0, 1, 2, 1, 0, 1]
    mlist = [models[k] for k in order]
    nbatch, nstep = len(order), 20
    s0 = np.zeros((nbatch, 1 + m.nq + m.nv))
    for e, mm in enumerate(mlist):
        rb.mj_resetData(mm, d)
        s0[e] = rb.mj_getState(mm, d, rb.mjSTATE_FULLPHYSICS)
    s0[:, 1 + m.nq:] = rng.normal(0, .2, size=(nbatch, m.nv))
    spec = rb.mjSTATE_USER
    n = rb.mj_stateSize(m, spec)
    control = np.zeros((nbatch, nstep, n))
    o = 0
    control[:, :, o:o + m.nu] = rng.uniform(-.02, .02, size=(nbatch, nstep, m.nu)); o += m.nu
    control[:, :, o:o + m.nv] = rng.normal(0, .1, size=(nbatch, nstep, m.nv)); o += m.nv
    xf = rng.normal(0, 2, size=(nbatch, nstep, m.nbody, 6)); xf[:, :, :2] = 0
    control[:, :, o:o + 6*m.nbody] = xf.reshape(nbatch, nstep, -1); o += 6*m.nbody
    eqa = np.ones((nbatch, nstep, m.neq)); eqa[0, 8:16, 0] = 0; eqa[1, 5:, 1] = 0
    control[:, :, o:o + m.neq] = eqa; o += m.neq
    control[:, :, o:o + 3] = [.05, .02, .42]; o += 3
    control[:, :, o:o + 4] = [1, .1, 0, .05]; o += 4
    control[:, :, o:] = rng.normal(size=(nbatch, nstep, m.nuserdata))
    state, _ = rollout.rollout(mlist, d, s0, control, control_spec=spec)
    ref = _py_rollout(rb, mlist, s0, control, spec)
    assert relerr(state, ref) <= TOL
    print("multi-model + all user inputs: rel err", relerr(state, ref))


def test_step1_step2_split_on_gpu(rb, hip_lib, dm):
    """mj_step1 / mj_step2 (engine_forward.c:1884-1939) on the GPU: a controller reads positions and
    velocities between the halves; against the oracle's own split"""
    m = humanoid_pgs_oracle(rb)
    states = contact_rich_states(rb, m, 5, seed=4)
    b = K.Batch(dm, len(states))
    b.set("qpos", np.stack([s["qpos"] for s in states]))
    b.set("qvel", np.stack([s["qvel"] for s in states]))
    b.set("qacc_warmstart", np.stack([s["qacc_warmstart"] for s in states]))
    ds = []
    for s in states:
        d = rb.MjData(m)
        d.qpos[:] = s["qpos"]; d.qvel[:] = s["qvel"]; d.qacc_warmstart[:] = s["qacc_warmstart"]
        ds.append(d)
    worst = 0.0
    for t in range(5):
        b.step1()
        xpos = b.get("xpos").reshape(len(states), -1, 3)
        qvel = b.get("qvel")
        ctrl = np.clip(-0.5*qvel[:, 6:6 + m.nu] + 0.3*(xpos[:, 1, 2:3] - 1.2), -1, 1)
        b.set("ctrl", ctrl)
        b.step2()
        got_q, got_v = b.get("qpos"), b.get("qvel")
        for e, d in enumerate(ds):
            rb.mj_step1(m, d)
            d.ctrl[:] = ctrl[e]
            rb.mj_step2(m, d)
            worst = max(worst, relerr(got_q[e], np.array(d.qpos)), relerr(got_v[e], np.array(d.qvel)))
            # re-synchronise (the controller feeds differences back)
            d.qpos[:] = got_q[e]; d.qvel[:] = got_v[e]
    print("step1/step2 closed loop: worst rel err", worst)
    assert worst <= TOL


def test_mfma_ar_tolerance_parity(rb, hip_lib, dm, golden):
    """opt-in matrix-core build of AR = Y Y' (v_mfma_f64_16x16x4_f64): AR agrees with the reference
    to rounding, the step within the 1e-6 bar; contact / constraint counts stay exact, solver
    iteration counts may move by one (tolerance parity, include/mjhip.h: mjhip_batch_set_mfma)"""
    m = humanoid_pgs_oracle(rb)
    states = contact_rich_states(rb, m, 48, seed=31)
    b = K.Batch(dm, len(states))
    b.set_mfma(True)
    from parity_utils import load_states
    load_states(b, states)
    b.forward()
    counts = b.get("counts")
    AR = b.get("efc_AR")
    qacc = b.get("qacc")
    d = rb.MjData(m)
    worst_ar, worst_q, dn = 0.0, 0.0, 0
    for e, s in enumerate(states):
        d.qpos[:] = s["qpos"]; d.qvel[:] = s["qvel"]; d.qacc_warmstart[:] = s["qacc_warmstart"]; d.ctrl[:] = s["ctrl"]
        rb.mj_forward(m, d)
        n = d.nefc
        assert counts[e][0] == d.ncon and counts[e][1] == n
        if n:
            ref = np.array(d.efc_AR).reshape(n, n)
            got = AR[e][:n*n].reshape(n, n)
            worst_ar = max(worst_ar, float(np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref)))))
        worst_q = max(worst_q, relerr(qacc[e], np.array(d.qacc)))
        dn = max(dn, abs(int(counts[e][5]) - int(d.solver_niter[0])))
    print("mfma AR: worst rel err", worst_ar, " qacc", worst_q, " max |delta niter|", dn, " max nefc", counts[:, 1].max())
    assert counts[:, 1].max() > 16            # more than one 16 x 16 tile is exercised
    assert worst_ar <= 1e-12 and worst_q <= 1e-6
    fx = golden("humanoid")
    n, T = fx["state0"].shape[0], 40
    b2 = K.Batch(dm, n)
    b2.set_mfma(True)
    out = b2.rollout_host(T, K.mjSTATE_CTRL, fx["state0"], None, fx["ctrl"][:, :T])
    assert relerr(out[:, :10], fx["state"][:, :10]) <= TOL


def test_pgs_residual_mode_tolerance_parity_on_gpu(rb, hip_lib, dm, golden):
    """opt-in residual-update PGS sweep (mjhip_batch_set_pgs_mode(1)): next states within the 1e-6 bar, counts exact,
    iteration counts reported; the default mode stays bit-exact (every other test of this file)"""
    from parity_utils import pgs_residual_parity
    m = humanoid_pgs_oracle(rb)
    states = contact_rich_states(rb, m, 48, seed=37)
    worst_f, worst_q, worst_s, dn, nmax = pgs_residual_parity(rb, K, m, dm, states, T=6)
    print("pgs residual: force", worst_f, "qacc", worst_q, "state", worst_s, "max |delta niter|", dn, "max nefc", nmax)
    assert nmax > 16
    assert worst_s <= TOL and worst_q <= 1e-6
    fx = golden("humanoid")
    n, T = fx["state0"].shape[0], 40
    b2 = K.Batch(dm, n)
    b2.set_pgs_mode(1)
    out = b2.rollout_host(T, K.mjSTATE_CTRL, fx["state0"], None, fx["ctrl"][:, :T])
    assert relerr(out[:, :10], fx["state"][:, :10]) <= TOL


def test_pgs_residual_mode_two_constraints_per_lane_on_gpu(rb, hip_lib, dm):
    """the residual-update sweep beyond 64 rows (solve_pgs_resid_wide) on the device: tolerance parity"""
    from parity_utils import pgs_residual_parity
    m = humanoid_pgs_oracle(rb)
    states, nefcs = many_constraint_states(rb, m, 16)
    assert min(nefcs) > 64
    worst_f, worst_q, worst_s, dn, nmax = pgs_residual_parity(rb, K, m, dm, states, T=4)
    print("pgs residual wide: force", worst_f, "qacc", worst_q, "state", worst_s, "max |delta niter|", dn, "max nefc", nmax)
    assert nmax > 64 and worst_s <= TOL and worst_q <= 1e-6


def test_broadphase_and_midphase_counts_exact_on_gpu(rb, hip_lib, tmp_path):
    """the reproduced sweep-and-prune / BVH culls on the GPU: exactly touching spheres (contact
    counts equal to the reference's in every scene) and a pile of multi-geom bodies that takes the
    midphase route (contact lists equal along a 100-step rollout)"""
    from parity_utils import MULTIGEOM_XML
    rng = np.random.default_rng(0)
    for trial in range(60):
        r1, r2 = [float(np.round(rng.uniform(.03, .2), 3)) for _ in range(2)]
        dirv = rng.normal(size=3)
        if trial % 3 == 0:
            dirv = np.eye(3)[trial % 9 // 3]
        dirv /= np.linalg.norm(dirv)
        p1 = np.round(rng.uniform(-1, 1, 3), 2)
        bodies = [(p1, r1), (p1 + dirv*(r1 + r2), r2)]
        if trial % 2:
            bodies = bodies[::-1]
        bodies += [(rng.uniform(-2, 2, 3), .05) for _ in range(rng.integers(0, 3))]
        f = tmp_path / "t.xml"
        f.write_text('<mujoco><option gravity="0 0 0"/><worldbody>' + "".join(
            '<body pos="%.17g %.17g %.17g"><freejoint/><geom type="sphere" size="%g"/></body>' % (*p, r) for p, r in bodies
        ) + "</worldbody></mujoco>")
        m = rb.MjModel.from_xml_path(str(f))
        d = rb.MjData(m)
        rb.mj_forward(m, d)
        b = K.Batch(K.DeviceModel(hip_lib, m), 1)
        b.forward()
        assert int(b.get("counts")[0, 0]) == d.ncon, trial
    xml = tmp_path / "multigeom.xml"
    xml.write_text(MULTIGEOM_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    T = 100
    ref, ints = oracle_rollout(rb, m, s0, np.zeros((1, T, 0)))
    dmm = K.DeviceModel(hip_lib, m)
    b = K.Batch(dmm, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, np.zeros((1, T, 0)))
    assert relerr(out[:, :30], ref[:, :30]) <= TOL
    bb = K.Batch(dmm, 1)
    for t in (0, 20, 40, 60, 80, 99):
        st = s0[0] if t == 0 else ref[0, t - 1]
        rb.mj_setState(m, d, st, rb.mjSTATE_FULLPHYSICS)
        rb.mj_forward(m, d)
        bb.set("qpos", st[None, 1:1 + m.nq]); bb.set("qvel", st[None, 1 + m.nq:])
        bb.forward()
        assert int(bb.get("counts")[0, 0]) == d.ncon, t
        if d.ncon:
            assert np.array_equal(bb.get("con_geom")[0].reshape(-1, 2)[:d.ncon], d.contact["geom"]), t


# ---- GJK / EPA / multicontact narrowphase (mjh_convex.h) on the GPU ---------------------------------------

def test_convex_pairs_on_gpu(rb, hip_lib, tmp_path):
    """mjc_Convex / mjc_PlaneConvex (engine_collision_convex.c:881,:1004; engine_collision_gjk.c) on the
    HIP path: pose sweeps over primitive pairs the reference routes to GJK/EPA (cylinder-box,
    capsule-cylinder, ellipsoids) and over convex meshes (exhaustive and hill-climbing support,
    multicontact clipping).  The narrowphase uses only + - * / sqrt, all correctly rounded on the
    device and evaluated in the reference's order, so contact count, geom ids, distance, position and
    frame are asserted identical, not merely close."""
    from convex_scenes import scene, sweep
    pairs = [("box", "cyl"), ("cyl", "cap"), ("cyl", "cyl"), ("ell", "box"), ("ell", "ell"), ("pla", "ell"),
             ("cub", "cub"), ("cub", "box"), ("ico", "cub"), ("tet", "tet"), ("cub", "cyl"), ("sph", "cub"), ("pla", "cub"), ("pla", "tet")]
    xml = tmp_path / "s.xml"
    contacts = multi = 0
    for pair in pairs:
        for margin, aligned in ((0.0, False), (0.02, False), (0.0, True)):
            xml.write_text(scene(pair[0], pair[1], margin, aligned))
            hist, bad = sweep(rb, K, hip_lib, xml, 128, seed=5, aligned=aligned)
            assert bad == 0, (pair, margin, aligned, hist)
            contacts += sum(c for n, c in hist.items() if n > 0)
            multi += sum(c for n, c in hist.items() if n > 1)
    print("convex sweep: poses with contacts", contacts, "with several contacts", multi)
    assert contacts > 1500 and multi > 300


def test_cube_3x3x3_single_steps_on_gpu(rb, hip_lib):
    """BASELINE config 4: model/cube/cube_3x3x3.xml as shipped (26 convex mesh cubelets, Newton,
    implicitfast, nv = 66, ~150 mesh-mesh contacts).  Every step of the committed reference trajectory
    is re-stepped from (state, warm start, ctrl): contact / row / Newton-iteration counts exact, the next
    state within the north star's 1e-6; then the same against the live oracle with the contact list
    compared record by record"""
    fx = np.load(os.path.join(GOLDEN, "cube_3x3x3_steps.npz"))
    mm = mujoco_amd.MjbModel(hip_lib, os.path.join(GOLDEN, "cube_3x3x3.mjb"))
    dm = K.DeviceModel(hip_lib, mm)
    n = fx["state"].shape[0]
    b = K.Batch(dm, n)
    out = b.rollout_host(1, K.mjSTATE_CTRL, fx["state"], fx["warmstart"], fx["ctrl"][:, None])
    assert b.get("warning").sum() == 0
    c = b.get("counts")
    assert np.array_equal(c[:, 0], fx["ints"][:, 0]) and np.array_equal(c[:, 1], fx["ints"][:, 1])
    assert np.array_equal(c[:, 5], fx["ints"][:, 2])
    err = relerr(out[:, 0], fx["next"])
    exact = np.array_equal(out[:, 0], fx["next"])
    print("cube: ncon up to", fx["ints"][:, 0].max(), "nefc up to", fx["ints"][:, 1].max(), "single-step rel err", err, "bit-exact:", exact)
    assert err <= TOL
    # nv = 66 puts the reference on its sparse path; mjh_sparse.h follows it operation for operation (bit for bit on
    # the host emulation, tests/test_sparse_hostsim.py); on the device most steps are bit-exact as well
    assert dm.size("sparse") == 1
    nexact = int(np.all(out[:, 0] == fx["next"], axis=1).sum())
    print("cube: steps reproduced bit for bit on the device against the reference as built (glibc sin / cos):", nexact, "of", n)
    # (measured: 158-160 of 160; a last-bit libm difference in a hinge quaternion moves about 1 % of the reference's own
    # cube steps beyond 1e-6, tests/test_oracle_golden.py::test_cube_contact_discontinuity -- none of them in this sample)
    assert nexact >= int(0.95*n)
    # against the reference linked with the kernels' own sin / cos (oracle/devmath_shim.cc) EVERY step is bit-exact:
    # the six face-centre hinges are the only libm calls of this model's step
    md = rb.MjModel.from_binary_path(os.path.join(GOLDEN, "cube_3x3x3.mjb"), kind="devmath")
    dd = rb.MjData(md)
    nxt = np.zeros_like(fx["next"])
    cnt = np.zeros((n, 3), np.int64)
    for e in range(n):
        rb.mj_setState(md, dd, fx["state"][e], rb.mjSTATE_FULLPHYSICS)
        dd.qacc_warmstart[:] = fx["warmstart"][e]
        dd.ctrl[:] = fx["ctrl"][e]
        rb.mj_step(md, dd)
        nxt[e] = rb.mj_getState(md, dd, rb.mjSTATE_FULLPHYSICS)
        cnt[e] = (dd.ncon, dd.nefc, dd.solver_niter[0])
    assert np.array_equal(c[:, 0], cnt[:, 0]) and np.array_equal(c[:, 1], cnt[:, 1]) and np.array_equal(c[:, 5], cnt[:, 2])
    assert np.array_equal(out[:, 0], nxt), int(np.sum(np.any(out[:, 0] != nxt, axis=1)))
    # contact records against the live oracle
    m = rb.MjModel.from_binary_path(os.path.join(GOLDEN, "cube_3x3x3.mjb"))
    d = rb.MjData(m)
    b.reset()
    nq = m.nq
    b.set("qpos", fx["state"][:, 1:1 + nq])
    b.set("qvel", fx["state"][:, 1 + nq:])
    b.forward()
    cd = b.get("con_dist"); cg = b.get("con_geom").reshape(n, -1, 2); cp = b.get("con_pos").reshape(n, -1, 3)
    for e in range(0, n, 8):
        rb.mj_resetData(m, d)
        rb.mj_setState(m, d, fx["state"][e], rb.mjSTATE_FULLPHYSICS)
        rb.mj_forward(m, d)
        k = d.ncon
        assert b.get("counts")[e, 0] == k
        rc = d.contact[:k]
        assert np.array_equal(cg[e, :k], rc["geom"])
        if k:
            assert np.abs(cd[e, :k] - rc["dist"]).max() <= 1e-9 and np.abs(cp[e, :k] - rc["pos"]).max() <= 1e-9


# scenes whose steps call no libm function other than sin / cos / atan2 / exp / sqrt (default solimp power: no pow): bit
# for bit against the reference linked with the kernels' own sin / cos / atan2 / exp (oracle/devmath_shim.cc)
DEVICE_EXACT_SCENES = ("chain", "equality", "islands", "tendon", "condim", "boxbox")


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["chain", "equality", "islands", "tendon", "condim", "boxbox"])
@pytest.mark.parametrize("solver,cone", [(2, 0), (2, 1), (1, 0), (0, 0), (0, 1)], ids=["newton-pyr", "newton-ell", "cg-pyr", "pgs-pyr", "pgs-ell"])
def test_sparse_primal_solvers_on_gpu(rb, hip_lib, tmp_path, scene, solver, cone):
    """the sparse constraint path (opt.jacobian = sparse; mjh_sparse.h, SPA instantiation of mjh_newton.h) on the
    device (tests/test_sparse_hostsim.py is the CPU counterpart, bit-exact against the reference as built).
    DEVICE_EXACT_SCENES: state trajectories, Newton / CG iteration counts, contact / row / island counts identical
    to the reference linked with the kernels' sin / cos.  The others (ball-joint limits, welds: the device's atan2 is
    not glibc's to the last bit): integer observables exact, states to 1e-9 per step from identical inputs and 1e-6
    over the rollout, iteration counts equal on >= 90 % of the steps"""
    from test_sparse_hostsim import SCENES, _run
    make, T = SCENES[scene]
    exact = scene in DEVICE_EXACT_SCENES
    ints = _run(rb, hip_lib, make(), tmp_path, solver, cone, T, exact=exact, kind="devmath" if exact else None)
    assert ints[:, 1].max() > 0 and ints[:, 2].max() > 0


@pytest.mark.parametrize("solver,cone", [(2, 0), (2, 1), (1, 0)], ids=["newton-pyr", "newton-ell", "cg-pyr"])
def test_primal_solvers_partial_islands_dense_on_gpu(rb, hip_lib, tmp_path, solver, cone):
    """islands that leave trees out, dense path: sums over the reference's island-local vectors (round 6; the CPU
    counterpart is tests/test_hostsim_parity.py::test_primal_solvers_partial_islands_dense_bit_exact).  Bit for bit
    against the reference linked with the kernels' sin / cos."""
    import test_hostsim_parity as th
    out, ref = th._partial_islands(rb, hip_lib, tmp_path, solver, cone, kind="devmath")
    assert np.array_equal(out, ref), relerr(out, ref)


@pytest.mark.parametrize("solver,cone", [(0, 0), (2, 0), (2, 1), (1, 1)], ids=["pgs-pyr", "newton-pyr", "newton-ell", "cg-ell"])
def test_noslip_solver_on_gpu(rb, hip_lib, tmp_path, solver, cone):
    """mj_solNoSlip after PGS / Newton / CG on the device (round 6; CPU counterpart:
    tests/test_hostsim_parity.py::test_noslip_solver_bit_exact): bit for bit against the reference linked with the
    kernels' sin / cos, iteration counts of island 0 included"""
    import test_hostsim_parity as th
    out, ref = th._noslip(rb, hip_lib, tmp_path, solver, cone, kind="devmath")
    assert np.array_equal(out, ref), relerr(out, ref)


def test_drop_in_rollout_with_host_arrays_equals_device_resident_rollout(rb, hip_lib, golden, monkeypatch):
    """`mujoco_amd.rollout.rollout` with numpy arrays (mjhip_rollout: chunked launches, strided copies overlapped with the
    kernels on two copy streams) returns the bytes of the device-resident rollout bench.py times.  (Device arrays through
    the HIP runtime the library itself is linked against: initialising torch's own copy of the runtime after libmjhip
    has initialised HIP fails with "No HIP GPUs are available".)"""
    import ctypes as C
    from mujoco_amd import rollout as ro
    monkeypatch.setenv("MJHIP_ROLLOUT_CHUNK", "50")          # (the second call below runs with the default: one launch)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]

    def dev_array(nbytes, src=None):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), nbytes) == 0
        if src is not None:
            assert hip.hipMemcpy(p, src.ctypes.data, nbytes, 1) == 0        # hipMemcpyHostToDevice
        return p

    fx = golden("humanoid")
    m = humanoid_pgs_oracle(rb)
    d = rb.MjData(m)
    n, T = 96, 230                                          # (4.6 chunks of 50 steps)
    rng = np.random.default_rng(5)
    s0 = np.ascontiguousarray(np.tile(fx["state0"], (n // fx["state0"].shape[0] + 1, 1))[:n])
    s0[:, 1 + 7:1 + m.nq] += rng.normal(0, 0.02, size=(n, m.nq - 7))
    ctrl = rng.uniform(-1, 1, size=(n, T, m.nu))
    state, sens = ro.rollout(m, d, s0, ctrl)
    assert state.shape == (n, T, s0.shape[1])
    dmv = K.DeviceModel(hip_lib, m)
    b = K.Batch(dmv, n)
    sd, cd = dev_array(s0.nbytes, s0), dev_array(ctrl.nbytes, ctrl)
    od = dev_array(state.nbytes)
    b.rollout_device(T, K.mjSTATE_CTRL, sd.value, 0, cd.value, od.value)
    b.sync()
    one = np.empty_like(state)
    assert hip.hipMemcpy(one.ctypes.data, od, state.nbytes, 2) == 0        # hipMemcpyDeviceToHost
    for p in (sd, cd, od): hip.hipFree(p)
    assert np.array_equal(one, state)
    # and a second call reuses the cached batch / staging buffers -- as ONE launch (the default)
    monkeypatch.delenv("MJHIP_ROLLOUT_CHUNK")
    state2, _ = ro.rollout(m, d, s0, ctrl)
    assert np.array_equal(state2, state)
