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
    # xfrc_applied is a legal user input for the reference; mjhip reports it as unsupported
    with pytest.raises(ValueError, match="not supported"):
        rollout.rollout(m, d, s0, np.zeros((nbatch, nstep, m.nu + 6*m.nbody)),
                        control_spec=K.mjSTATE_CTRL | K.mjSTATE_XFRC_APPLIED)


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
