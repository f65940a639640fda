"""The drop-in `mujoco_amd.rollout` against the reference's own test cases
(python/mujoco/rollout_test.py), on CPU: the library underneath is the host emulation of the kernels
(monkeypatched in place of libmjhip.so), the expected values come from the oracle's serial mj_step
loop (the reference's `py_rollout`, rollout_test.py:976-1001).  Exact equality, as in the reference."""
import numpy as np
import pytest

import mujoco_amd
from mujoco_amd import _capi as K
from mujoco_amd import rollout
from conftest import humanoid_pgs_oracle
from parity_utils import oracle_rollout


@pytest.fixture()
def api(rb, hostsim_lib, monkeypatch):
    monkeypatch.setattr(mujoco_amd, "lib", lambda: hostsim_lib)
    m = humanoid_pgs_oracle(rb)
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
    rng = np.random.default_rng(7)

    def states(n):
        out = np.tile(s0, (n, 1))
        out[:, 8:29] += rng.normal(0, 0.05, size=(n, 21))      # hinge angles
        out[:, 29:] = rng.normal(0, 0.1, size=(n, m.nv))
        return out
    return m, d, states, rng


def test_multi_step_and_tiling(rb, api):                       # rollout_test.py:199-253
    m, d, states, rng = api
    nbatch, nstep = 3, 4
    s0 = states(nbatch)
    ctrl = rng.uniform(-1, 1, size=(nbatch, nstep, m.nu))
    state, sensordata = rollout.rollout(m, d, s0, ctrl)
    ref, _ = oracle_rollout(rb, m, s0, ctrl)
    np.testing.assert_array_equal(state, ref)
    assert sensordata.shape == (nbatch, nstep, 0)
    # infer nbatch from initial_state: control [nstep, nu] is tiled
    state2, _ = rollout.rollout(m, d, s0, ctrl[0])
    ref2, _ = oracle_rollout(rb, m, s0, np.tile(ctrl[0], (nbatch, 1, 1)))
    np.testing.assert_array_equal(state2, ref2)
    # infer nbatch from control: a single initial state is tiled
    state3, _ = rollout.rollout(m, d, s0[0], ctrl)
    ref3, _ = oracle_rollout(rb, m, np.tile(s0[0], (nbatch, 1)), ctrl)
    np.testing.assert_array_equal(state3, ref3)


def test_fixed_ctrl_nstep_from_output(rb, api):                 # rollout_test.py:324-345, :391-410
    m, d, states, rng = api
    nstep = 3
    s0 = states(1)
    ctrl = rng.uniform(-1, 1, size=m.nu)
    state = np.empty((1, nstep, 56))
    rollout.rollout(m, d, s0[0], ctrl, state=state)
    ref, _ = oracle_rollout(rb, m, s0, np.tile(ctrl, (1, nstep, 1)))
    np.testing.assert_array_equal(state, ref)
    # explicit nstep with a fixed control
    state2, _ = rollout.rollout(m, d, s0[0], ctrl, nstep=nstep)
    np.testing.assert_array_equal(state2, ref)


def test_warmstart_argument(rb, api):                           # rollout_test.py:653-683
    m, d, states, rng = api
    s0 = states(2)
    for e in range(2):                      # keyframes with contacts, so the solver has work to do
        rb.mj_resetDataKeyframe(m, d, e)
        s0[e] = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
    ctrl = rng.uniform(-1, 1, size=(2, 2, m.nu))
    ws = rng.normal(0, 1, size=(2, m.nv))
    state, _ = rollout.rollout(m, d, s0, ctrl, initial_warmstart=ws)
    ref, _ = oracle_rollout(rb, m, s0, ctrl, warmstart0=ws)
    np.testing.assert_array_equal(state, ref)
    ref0, _ = oracle_rollout(rb, m, s0, ctrl)
    assert not np.array_equal(ref, ref0), "the warm start is supposed to matter here"


def test_generalized_control(rb, api):                          # rollout_test.py:412-438
    m, d, states, rng = api
    nbatch, nstep = 2, 3
    s0 = states(nbatch)
    spec = K.mjSTATE_CTRL | K.mjSTATE_QFRC_APPLIED
    control = np.concatenate([rng.uniform(-1, 1, size=(nbatch, nstep, m.nu)),
                              rng.normal(0, 5, size=(nbatch, nstep, m.nv))], axis=2)
    state, _ = rollout.rollout(m, d, s0, control, control_spec=spec)
    # py_rollout with mj_setState(control_spec)
    ref = np.zeros_like(state)
    for e in range(nbatch):
        rb.mj_resetData(m, d)
        rb.mj_setState(m, d, s0[e], rb.mjSTATE_FULLPHYSICS)
        for t in range(nstep):
            rb.mj_setState(m, d, control[e, t], spec)
            rb.mj_step(m, d)
            ref[e, t] = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
    np.testing.assert_array_equal(state, ref)
    # every mjSTATE_USER bit at once (humanoid: no equalities / mocap bodies / userdata, so those are
    # empty); Cartesian forces on bodies go through mj_xfrcAccumulate
    spec = rb.mjSTATE_USER
    xfrc = rng.normal(0, 20, size=(nbatch, nstep, 6*m.nbody))
    xfrc[:, :, :6] = 0                       # the world body's wrench is ignored by the reference
    xfrc[:, :, 6*3:6*9] = 0                  # and most bodies carry none
    control = np.concatenate([rng.uniform(-1, 1, size=(nbatch, nstep, m.nu)),
                              rng.normal(0, 5, size=(nbatch, nstep, m.nv)), xfrc], axis=2)
    assert control.shape[2] == rb.mj_stateSize(m, spec)
    state, _ = rollout.rollout(m, d, s0, control, control_spec=spec)
    ref = np.zeros_like(state)
    for e in range(nbatch):
        rb.mj_resetData(m, d)
        rb.mj_setState(m, d, s0[e], rb.mjSTATE_FULLPHYSICS)
        for t in range(nstep):
            rb.mj_setState(m, d, control[e, t], spec)
            rb.mj_step(m, d)
            ref[e, t] = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
    np.testing.assert_array_equal(state, ref)
    ref_noxfrc, _ = oracle_rollout(rb, m, s0, control[:, :, :m.nu])
    assert not np.array_equal(ref, ref_noxfrc)


def test_invalid_and_bad_sizes(api):                            # rollout_test.py:730-787
    m, d, states, rng = api
    s0 = states(1)
    with pytest.raises(ValueError, match="control must be a numpy array or float"):
        rollout.rollout(m, d, s0, "string")
    with pytest.raises(ValueError, match="control can have at most 3 dimensions"):
        rollout.rollout(m, d, s0, np.zeros((2, 3, 4, 5)))
    with pytest.raises(ValueError, match="trailing dimension of initial_state must be 56, got 57"):
        rollout.rollout(m, d, np.zeros((1, 57)))
    with pytest.raises(ValueError, match="trailing dimension of control must be 21, got 22"):
        rollout.rollout(m, d, s0, np.zeros((1, 3, m.nu + 1)))
    with pytest.raises(ValueError, match="dimension 1 inferred as 3 but state has 4"):
        rollout.rollout(m, d, s0, np.zeros((1, 3, m.nu)), state=np.zeros((1, 4, 56)))
    with pytest.raises(ValueError, match="control_spec can only contain bits in mjSTATE_USER"):
        rollout.rollout(m, d, s0, np.zeros((1, 3, m.nu)), control_spec=K.mjSTATE_ACT)


def test_stateless_and_final_data(rb, api):                     # rollout_test.py:789-814, rollout.cc:73
    m, d, states, rng = api
    s0 = states(1)[0]
    ctrl = rng.uniform(-1, 1, size=(3, 3, m.nu))
    state, _ = rollout.rollout(m, d, s0, ctrl)
    d.ctrl[:] = rng.normal(size=m.nu)
    d.qfrc_applied[:] = rng.normal(size=m.nv)
    d.xfrc_applied[:] = rng.normal(size=d.xfrc_applied.shape)
    state2, _ = rollout.rollout(m, d, s0, ctrl)
    np.testing.assert_array_equal(state, state2)
    # d holds the last step of the last rollout
    np.testing.assert_array_equal(np.array(d.qpos), state[-1, -1, 1:29])
    np.testing.assert_array_equal(np.array(d.qvel), state[-1, -1, 29:])
    assert d.time == state[-1, -1, 0]


def test_model_list_and_closed_object(rb, api):                 # rollout_test.py:816-829, rollout.py:102
    m, d, states, rng = api
    s0 = states(2)
    ctrl = rng.uniform(-1, 1, size=(2, 2, m.nu))
    state, _ = rollout.rollout([m], d, s0, ctrl)                 # length-one model list
    ref, _ = oracle_rollout(rb, m, s0, ctrl)
    np.testing.assert_array_equal(state, ref)
    with pytest.raises(ValueError, match="nbatch inferred as 2 but model is length 3"):
        rollout.rollout([m, m, m], d, s0, ctrl)
    r = rollout.Rollout(nthread=4)
    with r:
        st, _ = r.rollout(m, d, s0, ctrl)
        np.testing.assert_array_equal(st, ref)
    with pytest.raises(RuntimeError, match="after thread pool shutdown"):
        r.rollout(m, d, s0, ctrl)


def test_sensordata_output(rb, hostsim_lib, monkeypatch, tmp_path):   # rollout_test.py: sensordata of every step
    """a model with sensors: `rollout` returns the per-step sensordata next to the state, and the
    caller's mjData ends with the last rollout's final readings (rollout.cc:73,:130-133)"""
    from parity_utils import SENSOR_XML
    monkeypatch.setattr(mujoco_amd, "lib", lambda: hostsim_lib)
    xml = tmp_path / "sens.xml"
    xml.write_text(SENSOR_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
    rng = np.random.default_rng(11)
    nbatch, nstep = 2, 12
    init = np.tile(s0, (nbatch, 1))
    init[:, 1 + m.nq:] = rng.normal(0, .3, size=(nbatch, m.nv))
    ctrl = rng.uniform(-2, 2, size=(nbatch, nstep, m.nu))
    state, sensordata = rollout.rollout(m, d, init, ctrl)
    assert sensordata.shape == (nbatch, nstep, m.nsensordata)
    dd = rb.MjData(m)
    for r in range(nbatch):
        rb.mj_resetData(m, dd)
        rb.mj_setState(m, dd, init[r], rb.mjSTATE_FULLPHYSICS)
        for t in range(nstep):
            dd.ctrl[:] = ctrl[r, t]
            rb.mj_step(m, dd)
            np.testing.assert_array_equal(state[r, t], rb.mj_getState(m, dd, rb.mjSTATE_FULLPHYSICS))
            np.testing.assert_array_equal(sensordata[r, t], np.array(dd.sensordata))
    np.testing.assert_array_equal(np.array(d.sensordata), sensordata[-1, -1])


USER_XML = """
<mujoco>
  <option timestep="0.004" solver="PGS" iterations="50"/>
  <size nuserdata="3"/>
  <worldbody>
    <geom type="plane" size="3 3 .01"/>
    <body name="target" mocap="true" pos="0 0 .4" quat="1 0 0 0"><geom type="sphere" size=".02" contype="0" conaffinity="0"/></body>
    <body name="hand" pos="0 0 .35"><freejoint/><geom type="sphere" size=".05"/>
      <body pos="0 0 -.08"><joint name="finger" type="slide" axis="0 1 0" range="-.03 .03" limited="true" damping="1"/><geom type="capsule" size=".01 .02"/></body></body>
    <body name="ball" pos=".5 .05 .05"><freejoint/><geom type="sphere" size=".05" condim="3"/></body>
    <body name="p1" pos="-.5 0 .5"><joint name="q1" axis="0 1 0" damping=".02"/><geom type="capsule" fromto="0 0 0 .2 0 0" size=".02"/></body>
    <body name="p2" pos="-.5 .3 .5"><joint name="q2" axis="0 1 0" damping=".02"/><geom type="capsule" fromto="0 0 0 .2 0 0" size=".02"/></body>
  </worldbody>
  <equality>
    <weld body1="hand" body2="target" solref=".02 1"/>
    <joint joint1="q1" joint2="q2" polycoef="0 1 0 0 0"/>
  </equality>
  <actuator><position joint="finger" kp="20"/></actuator>
</mujoco>
"""


def _py_rollout(rb, models, s0, control, spec, warm=None):
    """the reference's py_rollout with a control spec and one model per rollout (rollout_test.py:976)"""
    nbatch, nstep = control.shape[:2]
    out = np.zeros((nbatch, nstep, s0.shape[1]))
    for e in range(nbatch):
        m = models[e]
        d = rb.MjData(m)
        rb.mj_resetData(m, d)
        rb.mj_setState(m, d, s0[e], rb.mjSTATE_FULLPHYSICS)
        if warm is not None:
            d.qacc_warmstart[:] = warm[e]
        for t in range(nstep):
            rb.mj_setState(m, d, control[e, t], spec)
            rb.mj_step(m, d)
            out[e, t] = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
    return out


def test_every_user_input_drives_the_rollout(rb, hostsim_lib, monkeypatch, tmp_path):
    """control_spec = mjSTATE_USER on a model that HAS equalities, a mocap body and userdata: ctrl,
    qfrc_applied, xfrc_applied, eq_active (switching constraints on and off mid-rollout), mocap
    pose and userdata, in mj_setState's bit order (engine_support.c:282; rollout.cc:160)"""
    monkeypatch.setattr(mujoco_amd, "lib", lambda: hostsim_lib)
    xml = tmp_path / "user.xml"
    xml.write_text(USER_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    s0 = np.tile(rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS), (3, 1))
    rng = np.random.default_rng(5)
    s0[:, 1 + m.nq:] = rng.normal(0, .2, size=(3, m.nv))
    nbatch, nstep = 3, 25
    spec = rb.mjSTATE_USER
    n = rb.mj_stateSize(m, spec)
    control = np.zeros((nbatch, nstep, n))
    o = 0
    control[:, :, o:o + m.nu] = rng.uniform(-.02, .02, size=(nbatch, nstep, m.nu)); o += m.nu
    control[:, :, o:o + m.nv] = rng.normal(0, .1, size=(nbatch, nstep, m.nv)); o += m.nv
    xf = rng.normal(0, 2, size=(nbatch, nstep, m.nbody, 6)); xf[:, :, 0] = 0; xf[:, :, 1] = 0; xf[:, ::2, 3] = 0
    control[:, :, o:o + 6*m.nbody] = xf.reshape(nbatch, nstep, -1); o += 6*m.nbody
    eqa = np.ones((nbatch, nstep, m.neq)); eqa[0, 8:16, 0] = 0; eqa[1, 5:, 1] = 0; eqa[2, :, :] = 0; eqa[2, 12:, 0] = 1
    control[:, :, o:o + m.neq] = eqa; o += m.neq
    for t in range(nstep):
        control[:, t, o:o + 3] = [.1*np.sin(t/7), .02*t/nstep, .4 + .05*np.cos(t/5)]
    o += 3*m.nmocap
    control[:, :, o:o + 4] = [1, .1, 0, .05]; o += 4*m.nmocap          # not normalised on purpose
    control[:, :, o:o + m.nuserdata] = rng.normal(size=(nbatch, nstep, m.nuserdata)); o += m.nuserdata
    assert o == n
    state, _ = rollout.rollout(m, d, s0, control, control_spec=spec)
    ref = _py_rollout(rb, [m]*nbatch, s0, control, spec)
    np.testing.assert_array_equal(state, ref)
    # the caller's mjData holds the last rollout's final inputs as well (rollout.cc:73)
    np.testing.assert_array_equal(np.array(d.userdata), control[-1, -1, n - m.nuserdata:])
    np.testing.assert_array_equal(np.array(d.eq_active), eqa[-1, -1])
    # same physics, inputs outside the spec reset: switching equalities off must have mattered
    on = control.copy(); on[:, :, 1 + m.nv + 6*m.nbody:1 + m.nv + 6*m.nbody + m.neq] = 1
    assert not np.array_equal(_py_rollout(rb, [m]*nbatch, s0, on, spec), ref)


def test_multi_model(rb, hostsim_lib, monkeypatch, tmp_path):     # rollout_test.py:363-389
    """one model PER ROLLOUT (the second body of every model is shifted, as in the reference's
    test): rollouts are grouped by model content, so interleaved duplicates share a batch"""
    monkeypatch.setattr(mujoco_amd, "lib", lambda: hostsim_lib)
    models = []
    for i in range(3):
        xml = tmp_path / f"m{i}.xml"
        xml.write_text(USER_XML.replace('name="hand" pos="0 0 .35"', f'name="hand" pos="{.1*i} 0 {.35 + .05*i}"'))
        models.append(rb.MjModel.from_xml_path(str(xml)))
    m0 = models[0]
    d = rb.MjData(m0)
    rng = np.random.default_rng(3)
    order = [0, 1, 2, 1, 0, 1]                   # interleaved: groups are not contiguous rows
    mlist = [models[k] for k in order]
    nbatch, nstep = len(order), 6
    s0 = np.zeros((nbatch, 1 + m0.nq + m0.nv))
    for e, m in enumerate(mlist):
        rb.mj_resetData(m, d)
        s0[e] = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
    s0[:, 1 + m0.nq:] = rng.normal(0, .3, size=(nbatch, m0.nv))
    control = rng.uniform(-.02, .02, size=(nbatch, nstep, m0.nu))
    state, _ = rollout.rollout(mlist, d, s0, control)
    ref = _py_rollout(rb, mlist, s0, control, rb.mjSTATE_CTRL)
    np.testing.assert_array_equal(state, ref)
    assert not np.array_equal(state[0], state[1])
    with pytest.raises(ValueError, match="identical sizes"):
        rollout.rollout([m0, humanoid_pgs_oracle(rb)], d, s0[:2], control[:2], skip_checks=True, nstep=nstep,
                        state=np.zeros((2, nstep, s0.shape[1])))


def test_control_none_steps_with_the_callers_inputs(rb, api):      # rollout.cc:85-115
    """control=None with mjSTATE_CTRL in the spec: the reference steps with whatever ctrl its
    mjData holds (only inputs OUTSIDE the spec are cleared)"""
    m, d, states, rng = api
    s0 = states(2)
    d.ctrl[:] = rng.uniform(-1, 1, size=m.nu)
    d.qfrc_applied[:] = rng.normal(size=m.nv)               # not in the spec: cleared
    held = np.array(d.ctrl)
    state, _ = rollout.rollout(m, d, s0, nstep=4)
    ref, _ = oracle_rollout(rb, m, s0, np.tile(held, (2, 4, 1)))
    np.testing.assert_array_equal(state, ref)
    zero, _ = oracle_rollout(rb, m, s0, np.zeros((2, 4, m.nu)))
    assert not np.array_equal(ref, zero)


def test_rollout_shards_over_devices(rb, api, monkeypatch):
    """mjhip_rollout cuts a batch into one contiguous piece per visible GPU, one host thread each,
    every piece writing its own rows of the caller's arrays; here two emulated devices"""
    m, d, states, rng = api
    nbatch, nstep = 9, 3
    s0 = states(nbatch)
    ctrl = rng.uniform(-1, 1, size=(nbatch, nstep, m.nu))
    ref, _ = oracle_rollout(rb, m, s0, ctrl)
    monkeypatch.setenv("MJH_HOSTSIM_DEVICES", "2")
    state, _ = rollout.rollout(m, d, s0, ctrl)
    np.testing.assert_array_equal(state, ref)
    np.testing.assert_array_equal(np.array(d.qpos), ref[-1, -1, 1:29])
    monkeypatch.setenv("MJHIP_DEVICES", "1")
    state1, _ = rollout.rollout(m, d, s0, ctrl)
    np.testing.assert_array_equal(state1, ref)


def test_capacity_overflow_is_reported(rb, api, monkeypatch):
    """an environment that overflows the contact capacity is frozen like after any warning -- the
    reference's arena would have grown, so the drop-in says so (return code 1 -> RuntimeWarning)"""
    m, d, states, rng = api
    rb.mj_resetDataKeyframe(m, d, 2)                 # prone: many contacts
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None]
    monkeypatch.setenv("MJHIP_NCONMAX", "2")
    mujoco_amd.lib().c.mjhip_rollout_clear_cache()
    with pytest.warns(RuntimeWarning, match="capacity"):
        rollout.rollout(m, d, s0, np.zeros((1, 3, m.nu)))
    monkeypatch.delenv("MJHIP_NCONMAX")
    mujoco_amd.lib().c.mjhip_rollout_clear_cache()


def test_partial_mocap_spec_and_userdata_follow_the_reference(rb, hostsim_lib, monkeypatch, tmp_path):
    """rollout.cc:98-109 resets mocap_pos and mocap_quat INDEPENDENTLY (each only when its bit is absent
    from the control spec) and never touches userdata: with control=None and only MOCAP_POS in the spec,
    the rollout steps with the caller's mocap_pos, the MODEL's mocap_quat and the caller's userdata --
    also when an earlier call left other values in the cached device batch"""
    monkeypatch.setattr(mujoco_amd, "lib", lambda: hostsim_lib)
    xml = tmp_path / "user.xml"
    xml.write_text(USER_XML)
    m = rb.MjModel.from_xml_path(str(xml))
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    nstep = 12
    # an earlier call that leaves foreign userdata / mocap values in the cached batch
    spec_all = rb.mjSTATE_USER
    junk = np.zeros((1, 2, rb.mj_stateSize(m, spec_all)))
    junk[..., -m.nuserdata:] = 7.0
    junk[..., -m.nuserdata - 4:-m.nuserdata] = [0, 1, 0, 0]
    junk[..., 1 + m.nv + 6*m.nbody:1 + m.nv + 6*m.nbody + m.neq] = 1
    rollout.rollout(m, d, s0, junk, control_spec=spec_all)
    # now: only MOCAP_POS in the spec, no control array
    rb.mj_resetData(m, d)
    d.mocap_pos[:] = [[.05, -.02, .45]]
    d.mocap_quat[:] = [[0, 0, 1, 0]]             # must be reset to the model's (1,0,0,0)
    d.userdata[:] = [1.5, -2.5, 3.5]             # must survive
    spec = rb.mjSTATE_MOCAP_POS
    state, _ = rollout.rollout(m, d, s0, nstep=nstep, control_spec=spec)
    dd = rb.MjData(m)
    rb.mj_resetData(m, dd)
    rb.mj_setState(m, dd, s0[0], rb.mjSTATE_FULLPHYSICS)
    dd.mocap_pos[:] = [[.05, -.02, .45]]
    ref = np.zeros((nstep, s0.shape[1]))
    for t in range(nstep):
        rb.mj_step(m, dd)
        ref[t] = rb.mj_getState(m, dd, rb.mjSTATE_FULLPHYSICS)
    np.testing.assert_array_equal(state[0], ref)
    np.testing.assert_array_equal(np.array(d.userdata), [1.5, -2.5, 3.5])
    np.testing.assert_array_equal(np.array(d.mocap_quat), [[1, 0, 0, 0]])
    np.testing.assert_array_equal(np.array(d.mocap_pos), [[.05, -.02, .45]])


def test_rollout_shards_over_eight_devices_with_ragged_batch(rb, api, monkeypatch):
    """BASELINE config 3's shape: 8 GPUs, a batch that does not divide by 8.  Every rollout is owned by
    exactly one device piece (disjoint, covering row ranges), the outputs equal the serial reference
    row by row, and d[0] holds the LAST rollout's final state whichever device stepped it"""
    m, d, states, rng = api
    nbatch, nstep = 19, 2
    s0 = states(nbatch)
    ctrl = rng.uniform(-1, 1, size=(nbatch, nstep, m.nu))
    ref, _ = oracle_rollout(rb, m, s0, ctrl)
    monkeypatch.setenv("MJH_HOSTSIM_DEVICES", "8")
    monkeypatch.setenv("MJHIP_DEVICES", "8")
    state = np.full((nbatch, nstep, s0.shape[1]), np.nan)
    out, _ = rollout.rollout(m, d, s0, ctrl, state=state)
    assert not np.isnan(out).any()                    # every row written by some piece
    np.testing.assert_array_equal(out, ref)
    np.testing.assert_array_equal(np.array(d.qpos), ref[-1, -1, 1:29])
    # the partition mjhip_rollout uses (one contiguous piece per device: [n*k/8, n*(k+1)/8))
    edges = [nbatch*k//8 for k in range(9)]
    assert edges[0] == 0 and edges[-1] == nbatch and all(b > a for a, b in zip(edges, edges[1:]))


def test_long_host_rollout_runs_in_overlapped_chunks(rb, hostsim_lib, golden, monkeypatch):
    """host arrays, $MJHIP_ROLLOUT_CHUNK = 50, nstep >= 100: rollout_impl launches the rollout in 50-step chunks (controls up / states and sensor data
    down as strided 2-D copies around the kernels, addressed through RolloutArgs.pitch / tbase).  The result has to be
    the one-launch result: compared with the same rollout taken through device-resident arrays (on the emulation "device"
    memory is host memory, so the device-pointer entry point can be fed numpy arrays), and with the golden trajectory."""
    monkeypatch.setenv("MJHIP_ROLLOUT_CHUNK", "50")          # (the default is one launch: see rollout_impl)
    fx = golden("humanoid")
    m = humanoid_pgs_oracle(rb)
    dm = K.DeviceModel(hostsim_lib, m)
    n, T = 3, fx["ctrl"].shape[1]
    assert T >= 100
    s0, ctrl = np.ascontiguousarray(fx["state0"][:n]), np.ascontiguousarray(fx["ctrl"][:n])
    b = K.Batch(dm, n)
    chunked = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    one = np.zeros_like(chunked)
    b2 = K.Batch(dm, n)
    b2.rollout_device(T, K.mjSTATE_CTRL, s0.ctypes.data, 0, ctrl.ctypes.data, one.ctypes.data)
    b2.sync()
    assert np.array_equal(chunked, one)
    ref = fx["state"][:n]
    assert np.max(np.abs(chunked - ref)/np.maximum(1.0, np.abs(ref))) <= 1e-6
