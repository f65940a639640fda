"""ad-hoc soak: 4096 environments of the sensor scene on the GPU, sampled environments checked
against the oracle (state and sensordata)"""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mujoco_amd as ma
from mujoco_amd import _capi as K
from oracle import refbind as rb
from parity_utils import SENSOR_XML, BOXBOX_XML, relerr

for name, xml, caps in (("sensor", SENSOR_XML, (0, 0)), ("boxbox", BOXBOX_XML, (64, 200))):
    open("/tmp/s.xml", "w").write(xml)
    m = rb.MjModel.from_xml_path("/tmp/s.xml")
    lib = ma.lib()
    dm = K.DeviceModel(lib, m, *caps)
    nenv, T = 4096, 30
    d = rb.MjData(m); rb.mj_resetData(m, d)
    s0 = np.tile(rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS), (nenv, 1))
    rng = np.random.default_rng(5)
    s0[:, 1 + m.nq:1 + m.nq + m.nv] = rng.normal(0, .4, (nenv, m.nv))
    ctrl = rng.uniform(-2, 2, (nenv, T, m.nu))
    b = K.Batch(dm, nenv)
    t0 = time.time()
    out, sd = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl, want_sensordata=True)
    dt = time.time() - t0
    worst = 0.0
    for e in rng.choice(nenv, 12, replace=False):
        rb.mj_resetData(m, d); rb.mj_setState(m, d, s0[e], rb.mjSTATE_FULLPHYSICS)
        for t in range(T):
            d.ctrl[:] = ctrl[e, t]; rb.mj_step(m, d)
            worst = max(worst, relerr(out[e, t], rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)))
            if m.nsensordata: worst = max(worst, relerr(sd[e, t], np.array(d.sensordata)))
    print(name, "4096 envs x", T, "steps: %.2f s incl. PCIe (%.2f M env-steps/s), worst rel err of 12 sampled envs %.2e, warnings %d" % (dt, nenv*T/dt/1e6, worst, int(b.get("warning").sum())))
