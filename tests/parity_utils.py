"""Comparison helpers shared by the hostsim (CPU) and HIP (GPU) parity tests."""
import numpy as np

from mujoco_amd import _capi as K

# every per-env array the forward pass produces, with the oracle attribute it mirrors
FORWARD_FIELDS = [
    "xpos", "xquat", "xmat", "xipos", "ximat", "xanchor", "xaxis", "geom_xpos", "geom_xmat",
    "subtree_com", "cinert", "cdof", "ten_length", "ten_J", "crb", "M", "qLD", "qLDiagInv",
    "actuator_length", "actuator_velocity", "actuator_force", "ten_velocity", "cvel", "cdof_dot",
    "qfrc_spring", "qfrc_damper", "qfrc_passive", "qfrc_bias", "qfrc_actuator", "qfrc_smooth",
    "qacc_smooth", "qfrc_constraint", "qacc",
]
EFC_FIELDS = ["efc_J", "efc_pos", "efc_margin", "efc_diagA", "efc_R", "efc_D", "efc_KBIP", "efc_Y",
              "efc_AR", "efc_vel", "efc_aref", "efc_b", "efc_force"]


def relerr(got, ref):
    got = np.asarray(got, dtype=np.float64).ravel()
    ref = np.asarray(ref, dtype=np.float64).ravel()
    if ref.size == 0:
        return 0.0
    return float(np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))))


def load_states(batch, states):
    batch.set("qpos", np.stack([s["qpos"] for s in states]))
    batch.set("qvel", np.stack([s["qvel"] for s in states]))
    batch.set("qacc_warmstart", np.stack([s["qacc_warmstart"] for s in states]))
    batch.set("ctrl", np.stack([s["ctrl"] for s in states]))


def check_forward(rb, m, batch, states, tol, exact_ints=True, lds=False):
    """run forward on the batch, mj_forward on the oracle for every env, compare everything.
    Integer observables (counts, types, ids, contact geoms, efc addresses, solver iterations) must
    match exactly; floats to `tol` relative (tol=0 -> bit-exact)."""
    load_states(batch, states)
    batch.forward(lds=lds)
    got = {f: batch.get(f) for f in FORWARD_FIELDS + EFC_FIELDS}
    counts = batch.get("counts")
    ints = {f: batch.get(f) for f in ["con_geom", "con_dim", "con_exclude", "con_efcadr", "efc_type", "efc_id", "efc_state"]}
    cons = {f: batch.get(f) for f in ["con_dist", "con_pos", "con_frame", "con_mu"]}
    d = rb.MjData(m)
    worst = 0.0
    nv = m.nv
    for e, s in enumerate(states):
        d.qpos[:] = s["qpos"]; d.qvel[:] = s["qvel"]; d.qacc_warmstart[:] = s["qacc_warmstart"]; d.ctrl[:] = s["ctrl"]
        rb.mj_forward(m, d)
        ncon, nefc = d.ncon, d.nefc
        assert counts[e][0] == ncon, f"env {e}: ncon {counts[e][0]} != {ncon}"
        assert counts[e][1] == nefc, f"env {e}: nefc {counts[e][1]} != {nefc}"
        assert counts[e][3] == d.nf and counts[e][4] == d.nl
        assert counts[e][5] == d.solver_niter[0], f"env {e}: PGS iterations {counts[e][5]} != {d.solver_niter[0]}"
        c = d.contact
        if ncon:
            assert np.array_equal(ints["con_geom"][e].reshape(-1, 2)[:ncon], c["geom"])
            assert np.array_equal(ints["con_dim"][e][:ncon], c["dim"])
            assert np.array_equal(ints["con_exclude"][e][:ncon], c["exclude"])
            assert np.array_equal(ints["con_efcadr"][e][:ncon], c["efc_address"])
            for f, ref in [("con_dist", c["dist"]), ("con_pos", c["pos"]), ("con_frame", c["frame"]), ("con_mu", c["mu"])]:
                err = relerr(cons[f][e][:np.asarray(ref).size], ref)
                worst = max(worst, err)
                assert err <= tol, f"env {e} {f}: {err}"
        if nefc:
            assert np.array_equal(ints["efc_type"][e][:nefc], d.efc_type)
            assert np.array_equal(ints["efc_id"][e][:nefc], d.efc_id)
            assert np.array_equal(ints["efc_state"][e][:nefc], d.efc_state)
            for f in EFC_FIELDS:
                ref = np.asarray(getattr(d, f)).ravel()
                err = relerr(got[f][e][:ref.size], ref)
                worst = max(worst, err)
                assert err <= tol, f"env {e} {f}: {err}"
        for f in FORWARD_FIELDS:
            ref = np.asarray(getattr(d, f)).ravel()
            err = relerr(got[f][e][:ref.size], ref)
            worst = max(worst, err)
            assert err <= tol, f"env {e} {f}: {err}"
    return worst


def pgs_residual_parity(rb, K, m, dm, states, T=6):
    """The opt-in residual-update PGS sweep (mjhip_batch_set_pgs_mode(1), mjh_solver.h: solve_pgs_resid) against the
    reference: one mj_forward from contact-rich states (forces / accelerations to rounding, counts exact) and T mj_step
    calls from each of them re-synchronised to the oracle's state (every next state within 1e-6).  Returns the worst
    relative errors and the largest iteration-count difference."""
    b = K.Batch(dm, len(states))
    b.set_pgs_mode(1)
    load_states(b, states)
    b.forward()
    counts, qacc, force = b.get("counts"), b.get("qacc"), b.get("efc_force")
    d = rb.MjData(m)
    worst_q = worst_f = 0.0
    dn = 0
    for e, s in enumerate(states):
        d.qpos[:] = s["qpos"]; d.qvel[:] = s["qvel"]; d.qacc_warmstart[:] = s["qacc_warmstart"]; d.ctrl[:] = s["ctrl"]
        rb.mj_forward(m, d)
        n = d.nefc
        assert counts[e][0] == d.ncon and counts[e][1] == n
        if n:
            ref = np.array(d.efc_force)[:n]
            worst_f = max(worst_f, float(np.max(np.abs(force[e][:n] - ref) / np.maximum(1.0, np.abs(ref)))))
        worst_q = max(worst_q, relerr(qacc[e], np.array(d.qacc)))
        dn = max(dn, abs(int(counts[e][5]) - int(d.solver_niter[0])))
    # steps from identical inputs (state, warm start, control), the oracle's trajectory as the common thread
    s0 = np.stack([np.concatenate([[s["time"]], s["qpos"], s["qvel"]]) for s in states])
    ws = np.stack([s["qacc_warmstart"] for s in states])
    u = np.stack([s["ctrl"] for s in states])
    worst_s = 0.0
    spec = rb.mjSTATE_FULLPHYSICS
    cur_s, cur_w = s0.copy(), ws.copy()
    for t in range(T):
        out = b.rollout_host(1, K.mjSTATE_CTRL, cur_s, cur_w, u[:, None])[:, 0]
        c = b.get("counts")
        for e in range(len(states)):
            rb.mj_setState(m, d, cur_s[e], spec)
            d.qacc_warmstart[:] = cur_w[e]
            d.ctrl[:] = u[e]
            rb.mj_step(m, d)
            ref = rb.mj_getState(m, d, spec)
            worst_s = max(worst_s, relerr(out[e], ref))
            assert c[e][0] == d.ncon and c[e][1] == d.nefc, (t, e)
            dn = max(dn, abs(int(c[e][5]) - int(d.solver_niter[0])))
            cur_s[e] = ref
            cur_w[e] = np.array(d.qacc_warmstart)
    return worst_f, worst_q, worst_s, dn, int(counts[:, 1].max())


def oracle_rollout(rb, m, state0, ctrl, warmstart0=None):
    """serial mj_step loop: the py_rollout of the reference's rollout_test.py:976-1001"""
    nenv, nstep = ctrl.shape[:2]
    d = rb.MjData(m)
    out = np.zeros((nenv, nstep, state0.shape[1]))
    ints = np.zeros((nenv, nstep, 3), np.int32)
    for e in range(nenv):
        rb.mj_resetData(m, d)
        rb.mj_setState(m, d, state0[e], rb.mjSTATE_FULLPHYSICS)
        d.qacc_warmstart[:] = 0 if warmstart0 is None else warmstart0[e]
        for t in range(nstep):
            d.ctrl[:] = ctrl[e, t]
            rb.mj_step(m, d)
            out[e, t] = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
            ints[e, t] = (d.ncon, d.nefc, d.solver_niter[0])
    return out, ints


# free cylinders (tilted, lying, upright) and a capsule dropping on a plane: multi-point
# plane-cylinder contacts, pyramidal + frictionless, one constraint island per body
CYL_XML = """
<mujoco>
  <option timestep="0.004" solver="PGS" iterations="60"/>
  <worldbody>
    <geom type="plane" size="2 2 .01"/>
    <body pos="0 0 .12" euler="25 40 0"><freejoint/><geom type="cylinder" size=".05 .08" condim="3"/></body>
    <body pos="1 0 .09" euler="90 0 10"><freejoint/><geom type="cylinder" size=".07 .03" condim="1"/></body>
    <body pos="0 1 .2" euler="0 0 0"><freejoint/><geom type="cylinder" size=".04 .1" condim="3"/></body>
    <body pos="1 1 .1"><freejoint/><geom type="capsule" size=".03 .06" condim="3"/></body>
  </worldbody>
</mujoco>
"""


# every equality type the GPU path implements, mixed with limits, friction loss and contacts:
# a 4-link chain closed on the world by a body connect, a site-based connect between two free
# spheres, a soft weld between free capsules, a site weld with torquescale to a static post, a
# static-static weld (dropped by the empty-Jacobian guard), a quadratic joint coupling, a tendon
# coupling, and one equality that starts inactive
EQ_XML = """
<mujoco>
  <option timestep="0.002" solver="PGS" iterations="40" tolerance="0"/>
  <default><geom type="capsule" size=".02" condim="3"/><joint damping=".02"/></default>
  <worldbody>
    <geom type="plane" size="3 3 .01"/>
    <body name="post" pos="1 0 0"><geom type="sphere" size=".03"/><site name="post_s" pos="0 0 .1"/></body>
    <body name="post2" pos="1 .5 0"><geom type="sphere" size=".03"/></body>
    <body name="l1" pos="0 0 .4">
      <joint name="h1" axis="0 1 0"/><geom fromto="0 0 0 .15 0 0"/>
      <body name="l2" pos=".15 0 0">
        <joint name="h2" axis="0 1 0" range="-100 100" limited="true"/><geom fromto="0 0 0 .15 0 0"/>
        <body name="l3" pos=".15 0 0">
          <joint name="h3" axis="0 1 0"/><geom fromto="0 0 0 .15 0 0"/>
          <body name="l4" pos=".15 0 0">
            <joint name="h4" type="ball"/><geom fromto="0 0 0 .15 0 0"/>
          </body>
        </body>
      </body>
    </body>
    <body name="s1" pos="0 .6 .0505"><freejoint/><geom type="sphere" size=".05"/><site name="s1_s" pos=".05 0 0"/></body>
    <body name="s2" pos=".14 .6 .0405"><freejoint/><geom type="sphere" size=".04"/><site name="s2_s" pos="-.04 0 0"/></body>
    <body name="c1" pos="-.6 0 .061" euler="0 30 0"><freejoint/><geom fromto="-.08 0 0 .08 0 0"/></body>
    <body name="c2" pos="-.6 .1 .07" euler="0 35 40"><freejoint/><geom fromto="-.08 0 0 .08 0 0"/></body>
    <body name="w1" pos="1 0 .25"><freejoint/><geom type="sphere" size=".04"/><site name="w1_s" pos="0 0 -.1" euler="0 0 20"/></body>
    <body name="p1" pos="-.3 -.6 .3"><joint name="q1" axis="1 0 0" frictionloss=".02"/><geom fromto="0 0 0 0 0 -.2"/></body>
    <body name="p2" pos="0 -.6 .3"><joint name="q2" axis="1 0 0"/><geom fromto="0 0 0 0 0 -.2"/></body>
    <body name="p3" pos=".3 -.6 .3"><joint name="q3" axis="1 0 0"/><geom fromto="0 0 0 0 0 -.25"/></body>
    <body name="p4" pos=".6 -.6 .3"><joint name="q4" type="slide" axis="0 0 1" range="-.3 .3" limited="true"/><geom type="sphere" size=".04"/></body>
  </worldbody>
  <tendon>
    <fixed name="t1"><joint joint="q3" coef="1"/></fixed>
    <fixed name="t2" frictionloss=".08" solreffriction=".03 1"><joint joint="q4" coef="2"/><joint joint="q2" coef=".3"/></fixed>
  </tendon>
  <equality>
    <connect body1="l4" body2="world" anchor=".15 0 0"/>
    <connect site1="s1_s" site2="s2_s" solref=".01 1"/>
    <weld body1="c1" body2="c2" solref=".03 .7" solimp=".8 .95 .01"/>
    <weld site1="w1_s" site2="post_s" torquescale=".5"/>
    <weld body1="post" body2="post2"/>
    <joint joint1="q1" joint2="q2" polycoef=".1 .8 .3 0 0"/>
    <tendon tendon1="t1" tendon2="t2" polycoef="0 1.2 0 0 0"/>
    <joint joint1="q3" active="false"/>
  </equality>
  <actuator><motor joint="h1" gear="1"/><motor joint="q1" gear=".5"/></actuator>
</mujoco>
"""


# implicitfast coverage: velocity-dependent actuators (velocity servo, position servo with kv, affine
# gain with a velocity term, force-limited), tendon damping, joint damping, and standalone free
# bodies whose COM is off the joint origin (gyroscopic 6x6 solve) next to a free body with a child
IMPL_XML = """
<mujoco>
  <option timestep="0.004" solver="PGS" iterations="40" integrator="implicitfast"/>
  <default><geom type="capsule" size=".03" condim="3"/><joint damping=".3"/></default>
  <worldbody>
    <geom type="plane" size="3 3 .01"/>
    <body name="a1" pos="0 0 .8">
      <joint name="j1" axis="0 1 0"/><geom fromto="0 0 0 .25 0 0"/>
      <body name="a2" pos=".25 0 0">
        <joint name="j2" axis="0 1 0" range="-120 120" limited="true"/><geom fromto="0 0 0 .2 0 0"/>
        <body name="a3" pos=".2 0 0">
          <joint name="j3" axis="0 0 1"/><geom fromto="0 0 0 .15 0 0"/>
          <body name="a4" pos=".15 0 0"><joint name="j4" type="slide" axis="1 0 0" damping="2"/><geom type="sphere" size=".04"/></body>
        </body>
      </body>
    </body>
    <body name="f1" pos="-.5 0 .3" euler="20 30 0"><freejoint/><geom type="capsule" fromto="0 0 0 .2 .05 0" size=".04"/><geom type="sphere" size=".06" pos=".1 .1 .05"/></body>
    <body name="f2" pos="-.5 .6 .09" euler="0 0 0"><joint type="free" damping=".05"/><geom type="cylinder" size=".08 .03" pos=".02 0 0"/></body>
    <body name="f3" pos=".6 .6 .3"><freejoint/><geom type="sphere" size=".05"/>
      <body pos=".12 0 0"><joint name="k1" axis="0 1 0"/><geom fromto="0 0 0 .1 0 0"/></body></body>
  </worldbody>
  <tendon>
    <fixed name="t1" damping=".8"><joint joint="j2" coef="1"/><joint joint="j3" coef="-.7"/></fixed>
  </tendon>
  <actuator>
    <velocity joint="j1" kv="2" ctrlrange="-3 3"/>
    <position joint="j2" kp="8" kv=".6"/>
    <general joint="j3" gaintype="affine" gainprm="1.5 0 -.4" biastype="affine" biasprm="0 -1 -.2"/>
    <velocity joint="j4" kv="5" forcerange="-.4 .4"/>
    <motor joint="k1" gear=".3"/>
  </actuator>
</mujoco>
"""


# torsional (condim 4) and rolling (condim 6) friction: spinning / rolling spheres, a capsule and a
# cylinder on a plane, plus a sphere-sphere stack; used with both pyramidal and elliptic cones
CONDIM_XML = """
<mujoco>
  <option timestep="0.004" solver="PGS" iterations="50" impratio="3"/>
  <worldbody>
    <geom type="plane" size="3 3 .01" condim="3" friction="1 .02 .002"/>
    <body pos="0 0 .0502"><freejoint/><geom type="sphere" size=".05" condim="4" friction=".8 .03 .001"/></body>
    <body pos=".4 0 .0604"><freejoint/><geom type="sphere" size=".06" condim="6" friction=".6 .02 .004"/></body>
    <body pos=".4 0 .161"><freejoint/><geom type="sphere" size=".04" condim="6" friction=".9 .01 .003"/></body>
    <body pos="-.4 0 .041" euler="0 88 0"><freejoint/><geom type="capsule" size=".04 .1" condim="6" friction=".7 .05 .002"/></body>
    <body pos="0 .5 .0505"><freejoint/><geom type="cylinder" size=".07 .05" condim="4" friction="1.2 .04 .001"/></body>
    <body pos=".5 .5 .2"><joint type="slide" axis="0 0 1" damping="1"/><joint type="hinge" axis="0 0 1" damping=".01"/>
      <geom type="sphere" size=".05" condim="4" friction=".5 .05 .001"/></body>
  </worldbody>
</mujoco>
"""


def condim_scene_state(rb, m):
    """initial state of CONDIM_XML: bodies resting on the plane, sliding slowly and spinning fast"""
    import numpy as np
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    rng = np.random.default_rng(3)
    v = rng.normal(0, 1.0, m.nv)
    for k in range(5):            # the five free bodies
        v[6*k:6*k+3] *= 0.3
        v[6*k+2] = -0.05
        v[6*k+3:6*k+6] *= 4.0
    v[30] = -0.5
    v[31] = 6.0
    d.qvel[:] = v
    return rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()


def chain_xml(nlinks=20):
    """serial chain deeper than the register-resident L'DL routines handle (depth > 16), with a
    ball joint (limited), a slide joint, hinges, motors and a position actuator"""
    xml = ['<mujoco>', '  <option timestep="0.003" solver="PGS" iterations="50" jacobian="dense"/>',
           '  <default><geom type="capsule" size=".02" condim="3"/><joint damping=".05" armature=".001"/></default>',
           '  <worldbody>', '    <geom type="plane" size="5 5 .01" pos="0 0 -.6"/>',
           '    <body pos="0 0 0">', '      <joint name="j0" type="ball" limited="true" range="0 1.2"/>',
           '      <geom fromto="0 0 0 .1 0 0"/>']
    for i in range(1, nlinks + 1):
        if i == 7:
            jt = 'type="slide" axis="1 0 0" range="-.02 .02" limited="true"'
        elif i == 12:
            jt = 'type="ball"'
        else:
            jt = 'type="hinge" axis="0 %d %d" range="-60 60" limited="true"' % (i % 2, (i + 1) % 2)
        xml += ['<body pos=".1 0 0">', '<joint name="j%d" %s/>' % (i, jt), '<geom fromto="0 0 0 .1 0 0"/>']
    xml += ['</body>'] * nlinks
    xml += ['    </body>', '  </worldbody>',
            '  <actuator><motor joint="j3" gear="2"/><motor joint="j9" gear="1"/><position joint="j15" kp="5"/></actuator>',
            '</mujoco>']
    return '\n'.join(xml)


# stateful actuators: first-order filters (Euler and exact), an integrated-velocity servo with an
# activation range, one with actearly; joint damping so that every integrator has work to do
ACT_XML = """
<mujoco>
  <option timestep="0.004" solver="PGS" iterations="40"/>
  <default><geom type="capsule" size=".03" condim="3"/><joint damping=".2" armature=".01"/></default>
  <worldbody>
    <geom type="plane" size="3 3 .01"/>
    <body name="a1" pos="0 0 .7">
      <joint name="j1" axis="0 1 0"/><geom fromto="0 0 0 .25 0 0"/>
      <body name="a2" pos=".25 0 0">
        <joint name="j2" axis="0 1 0" range="-120 120" limited="true"/><geom fromto="0 0 0 .2 0 0"/>
        <body name="a3" pos=".2 0 0">
          <joint name="j3" axis="0 0 1"/><geom fromto="0 0 0 .15 0 0"/>
          <body name="a4" pos=".15 0 0"><joint name="j4" type="slide" axis="1 0 0" range="-.1 .1" limited="true"/><geom type="sphere" size=".04"/></body>
        </body>
      </body>
    </body>
    <body name="b1" pos="0 .5 .4"><joint name="k1" axis="1 0 0"/><geom fromto="0 0 0 0 0 -.3"/>
      <body pos="0 0 -.3"><joint name="k2" axis="1 0 0"/><geom fromto="0 0 0 0 0 -.25"/></body></body>
  </worldbody>
  <actuator>
    <position joint="j1" kp="20" kv="1" timeconst=".05" ctrlrange="-1 1"/>
    <general joint="j2" dyntype="filter" dynprm=".03" gainprm="4" biastype="affine" biasprm="0 0 -.3"/>
    <intvelocity joint="j3" kp="15" kv=".5" actrange="-1.2 1.2"/>
    <general joint="j4" dyntype="integrator" gainprm="3" actlimited="true" actrange="-.5 .5" actearly="true"/>
    <general joint="k1" dyntype="filterexact" dynprm=".02" gainprm="2" actearly="true"/>
    <motor joint="k2" gear=".5"/>
  </actuator>
</mujoco>
"""


# sensors: every kind the GPU path evaluates, on a model with contacts, a weld, limits, a tendon,
# a ball joint and a free body (IMU sites on moving bodies, frame sensors with reference frames)
SENSOR_XML = """
<mujoco>
  <option timestep="0.004" solver="PGS" iterations="50" magnetic="0 -.3 .4"/>
  <default><geom type="capsule" size=".03" condim="3"/><joint damping=".1"/></default>
  <worldbody>
    <geom name="floor" type="plane" size="3 3 .01"/>
    <geom name="tgt_box" type="box" size=".2 .3 .1" pos="1.3 0 .3" euler="10 20 0" contype="0" conaffinity="0"/><geom name="tgt_cyl" type="cylinder" size=".15 .3" pos="-.5 .4 -.05" euler="20 0 0" contype="0" conaffinity="0"/>
    <geom name="tgt_ell" type="ellipsoid" size=".2 .3 .1" pos=".6 0 .1" contype="0" conaffinity="0"/><geom name="ghost" type="sphere" size=".3" pos="-.5 .4 .1" rgba="1 0 0 0" contype="0" conaffinity="0"/>
    <site name="world_s" pos=".2 .1 .3" euler="10 20 30"/>
    <site name="zone_box" type="box" size=".3 .3 .06" pos="-.5 0 .05"/><site name="zone_cyl" type="cylinder" size=".3 .2" pos=".5 0 .5" euler="0 20 0"/>
    <site name="zone_sph" type="sphere" size=".25" pos=".45 0 .45"/><site name="zone_cap" type="capsule" size=".15 .3" pos=".4 0 .6" euler="0 90 0"/>
    <body name="a1" pos="0 0 .6">
      <joint name="j1" axis="0 1 0" range="-40 40" limited="true" stiffness="3" springref="10"/><geom name="g1" fromto="0 0 0 .25 0 0"/>
      <site name="imu1" pos=".1 0 .02" euler="0 15 40"/><site name="rf_side" pos="0 0 .05" euler="0 100 0"/>
      <body name="a2" pos=".25 0 0">
        <joint name="j2" type="ball" stiffness=".8"/><geom fromto="0 0 0 .2 0 0"/>
        <site name="imu2" pos=".15 .01 0" euler="30 0 0"/>
        <body name="a3" pos=".2 0 0"><joint name="j3" type="slide" axis="1 0 0" range="-.05 .05" limited="true"/>
          <geom name="g3" type="sphere" size=".05"/><site name="tip" pos=".05 0 0"/><site name="rf_tip" pos=".06 0 0" euler="0 130 20"/></body>
      </body>
    </body>
    <body name="f1" pos="-.5 0 .08" euler="0 70 20"><freejoint/><geom name="gf" fromto="-.1 0 0 .1 0 0" size=".04"/>
      <site name="imuf" pos=".05 0 .01" euler="5 10 15"/>
      <site name="touch_box" type="box" size=".16 .06 .06"/><site name="touch_sph" type="sphere" size=".045" pos=".1 0 0"/>
      <site name="touch_ell" type="ellipsoid" size=".13 .05 .03" pos="-.04 0 0"/>
      <site name="touch_cap" type="capsule" size=".042 .07" pos=".02 0 0" euler="0 90 0"/><site name="touch_cyl" type="cylinder" size=".045 .12" euler="0 90 5"/></body>
    <body name="f2" pos="-.5 .4 .3"><freejoint/><geom type="sphere" size=".05"/><site name="f2s"/><site name="rf_down" pos="0 0 -.06" euler="180 0 0"/><site name="rf_up" pos="0 0 .06"/></body>
    <body name="p1" pos=".3 -.5 .5"><joint name="q1" axis="1 0 0"/><geom fromto="0 0 0 0 0 -.2"/>
      <body pos="0 0 -.2"><joint name="q2" axis="1 0 0" range="-30 30" limited="true"/><geom fromto="0 0 0 0 0 -.2"/><site name="p_end" pos="0 0 -.2"/></body></body>
  </worldbody>
  <tendon><fixed name="t1" range="-.2 .2" limited="true" stiffness="4" springlength="-.05 .05"><joint joint="q1" coef="1"/><joint joint="q2" coef=".5"/></fixed></tendon>
  <equality><weld body1="f2" body2="a1" solref=".02 1"/></equality>
  <actuator>
    <motor name="m1" joint="j1" gear="2"/><position name="m2" joint="j3" kp="30"/><motor name="m3" joint="q1" gear=".8"/>
  </actuator>
  <sensor>
    <jointpos joint="j1"/><jointvel joint="j1"/><jointpos joint="j3"/><jointvel joint="q2"/>
    <tendonpos tendon="t1"/><tendonvel tendon="t1"/>
    <actuatorpos actuator="m2"/><actuatorvel actuator="m2"/><actuatorfrc actuator="m2"/><actuatorfrc actuator="m1" cutoff=".5"/>
    <jointactuatorfrc joint="j1"/>
    <ballquat joint="j2"/><ballangvel joint="j2"/>
    <jointlimitpos joint="j1"/><jointlimitvel joint="j1"/><jointlimitfrc joint="j1"/>
    <jointlimitpos joint="j3"/><jointlimitfrc joint="j3"/>
    <tendonlimitpos tendon="t1"/><tendonlimitvel tendon="t1"/><tendonlimitfrc tendon="t1"/>
    <framepos objtype="site" objname="tip"/><framepos objtype="body" objname="a3" reftype="site" refname="world_s"/>
    <framequat objtype="xbody" objname="a2"/><framequat objtype="geom" objname="g1" reftype="body" refname="f1"/>
    <framexaxis objtype="site" objname="imu1"/><frameyaxis objtype="body" objname="a2" reftype="xbody" refname="a1"/><framezaxis objtype="geom" objname="gf"/>
    <framelinvel objtype="site" objname="tip"/><frameangvel objtype="body" objname="a3"/>
    <framelinvel objtype="site" objname="tip" reftype="site" refname="imuf"/><frameangvel objtype="xbody" objname="f1" reftype="body" refname="a2"/>
    <framelinacc objtype="site" objname="tip"/><frameangacc objtype="body" objname="f1"/>
    <subtreecom body="a1"/><subtreelinvel body="a1"/><subtreeangmom body="a1"/><subtreeangmom body="p1"/>
    <clock/><e_potential/><e_kinetic/>
    <distance geom1="g3" geom2="floor" cutoff="1"/><normal geom1="g1" geom2="gf" cutoff="2"/><fromto geom1="gf" geom2="g3" cutoff="3"/>
    <distance body1="a2" body2="f1" cutoff="2"/><fromto body1="f1" body2="a1" cutoff="2"/><normal geom1="floor" geom2="gf" cutoff=".5"/>
    <fromto geom1="g3" geom2="tgt_box" cutoff="2"/><distance geom1="tgt_cyl" geom2="g3" cutoff="2"/><fromto geom1="ghost" geom2="g3" cutoff=".2"/>
    <distance geom1="floor" geom2="g3" cutoff=".01"/>
    <velocimeter site="imu1"/><gyro site="imu2"/><accelerometer site="imu1"/><accelerometer site="imuf"/>
    <force site="imu2"/><torque site="imu2"/><force site="imuf"/><torque site="p_end"/>
    <magnetometer site="imu1"/>
    <touch site="touch_box"/><touch site="touch_sph"/><touch site="touch_ell"/><touch site="touch_cap"/><touch site="touch_cyl"/>
    <rangefinder site="rf_down" data="dist normal"/><rangefinder site="rf_side" data="dist dir origin point normal depth"/><rangefinder site="rf_tip" data="dist point normal"/><rangefinder site="rf_up"/>
    <insidesite objtype="body" objname="f1" site="zone_box"/><insidesite objtype="site" objname="tip" site="zone_cyl"/><insidesite objtype="xbody" objname="a3" site="zone_sph"/><insidesite objtype="geom" objname="g3" site="zone_cap"/>
    <framepos objtype="site" objname="world_s"/><framelinvel objtype="site" objname="world_s"/>
    <contact geom1="gf" num="2" data="found force torque dist pos normal tangent"/><contact body2="f1" num="2" data="force normal" cutoff="0.05"/>
    <contact body1="f1" num="3" data="found dist pos" reduce="mindist"/><contact site="zone_box" num="2" data="found force" reduce="maxforce"/>
    <contact subtree1="a1" data="found force torque pos" reduce="netforce"/><contact subtree1="f1" geom2="floor" data="force torque pos" reduce="netforce"/>
    <contact/>
  </sensor>
</mujoco>
"""


# plane-box (1..4 corner contacts), sphere-box (outside, and centre inside the box), sphere-cylinder
# (side, cap and rim cases); the boxes never meet each other or a capsule (those colliders are not
# on the GPU path yet)
BOX_XML = """
<mujoco>
  <option timestep="0.004" solver="PGS" iterations="60"/>
  <worldbody>
    <geom type="plane" size="4 4 .01"/>
    <geom name="table" type="box" size=".3 .3 .05" pos="1.5 0 .05"/>
    <geom name="pillar" type="cylinder" size=".15 .2" pos="-1.5 0 .2"/>
    <body pos="0 0 .12" euler="20 35 0"><freejoint/><geom type="box" size=".06 .09 .05" condim="3"/></body>
    <body pos=".6 0 .051"><freejoint/><geom type="box" size=".1 .1 .05" condim="4"/></body>
    <body pos="0 .7 .2" euler="44 44 10"><freejoint/><geom type="box" size=".07 .07 .07" condim="1"/></body>
    <body pos="1.5 .05 .16"><freejoint/><geom type="sphere" size=".06" condim="3"/></body>
    <body pos="1.78 .1 .1"><freejoint/><geom type="sphere" size=".05" condim="3"/></body>
    <body pos="1.45 -.1 .08"><freejoint/><geom type="sphere" size=".02" condim="1"/></body>
    <body pos="-1.5 .02 .465"><freejoint/><geom type="sphere" size=".07" condim="3"/></body>
    <body pos="-1.29 0 .2"><freejoint/><geom type="sphere" size=".06" condim="3"/></body>
    <body pos="-1.38 .13 .44"><freejoint/><geom type="sphere" size=".05" condim="1"/></body>
  </worldbody>
</mujoco>
"""


# box-box: a box resting on a static table (face manifold, 4 points), a rotated box on it (clipped
# polygon), a stack of two free boxes, an edge-on-edge crossing and a corner-into-face landing
BOXBOX_XML = """
<mujoco>
  <option timestep="0.004" solver="PGS" iterations="60"/>
  <worldbody>
    <geom type="plane" size="4 4 .01"/>
    <geom name="table" type="box" size=".4 .4 .05" pos="0 0 .3"/>
    <geom name="bar" type="box" size=".5 .03 .03" pos="1.5 0 .3" euler="0 0 20"/>
    <body pos="-.15 -.1 .401"><freejoint/><geom type="box" size=".08 .06 .05" condim="3"/></body>
    <body pos=".2 .15 .405" euler="0 0 33"><freejoint/><geom type="box" size=".07 .1 .05" condim="3"/>
      </body>
    <body pos=".2 .15 .505" euler="0 0 10"><freejoint/><geom type="box" size=".05 .05 .05" condim="4"/></body>
    <body pos="1.5 0 .37" euler="90 0 80"><freejoint/><geom type="box" size=".03 .03 .3" condim="3"/></body>
    <body pos="-.2 .2 .49" euler="40 35 10"><freejoint/><geom type="box" size=".06 .06 .06" condim="1"/></body>
    <body pos="-1 0 .051"><freejoint/><geom type="box" size=".2 .2 .05" condim="3"/></body>
    <body pos="-1.05 .02 .152" euler="0 0 45"><freejoint/><geom type="box" size=".1 .1 .05" condim="3"/></body>
  </worldbody>
</mujoco>
"""


# capsules against boxes (mjc_CapsuleBox): lying on a table (two contacts along the capsule), standing
# on it, hanging over an edge, leaning on a corner, crossing a thin bar, plus one inside a box's margin
CAPBOX_XML = """
<mujoco>
  <option timestep="0.004" solver="PGS" iterations="60"/>
  <worldbody>
    <geom type="plane" size="4 4 .01"/>
    <geom name="table" type="box" size=".4 .3 .05" pos="0 0 .3" euler="0 0 15"/>
    <geom name="bar" type="box" size=".5 .03 .03" pos="1.5 0 .3" euler="0 0 20"/>
    <geom name="step" type="box" size=".2 .2 .1" pos="-1 0 .1" margin=".01"/>
    <body pos="-.1 -.05 .395" euler="90 0 30"><freejoint/><geom type="capsule" size=".04 .15" condim="3"/></body>
    <body pos=".15 .1 .55" euler="3 2 0"><freejoint/><geom type="capsule" size=".04 .15" condim="3"/></body>
    <body pos=".38 0 .39" euler="0 90 15"><freejoint/><geom type="capsule" size=".03 .2" condim="4"/></body>
    <body pos="-.3 -.28 .42" euler="50 20 0"><freejoint/><geom type="capsule" size=".05 .1" condim="3"/></body>
    <body pos="1.5 0 .365" euler="90 0 80"><freejoint/><geom type="capsule" size=".03 .3" condim="1"/></body>
    <body pos="-1 .1 .26" euler="0 80 45"><freejoint/><geom type="capsule" size=".05 .12" condim="3"/></body>
    <body pos="-.8 -.1 .3" euler="20 0 0"><freejoint/><geom type="capsule" size=".04 .2" condim="6"/></body>
  </worldbody>
</mujoco>
"""


# 90 free spheres dropping onto a plane: more than 64 pairs survive the bounding-sphere filter (every
# plane-sphere pair does), so the collision stage takes its chunk-by-chunk route, and the 30 spheres that
# land first give 120 pyramid rows: the two-constraints-per-lane PGS
def many_spheres_xml():
    bodies = "".join(f'<body pos="{0.25*(i % 10)} {0.25*(i // 10)} {0.06 + 0.002*(i % 3)}"><freejoint/>'
                     f'<geom type="sphere" size=".05" condim="1"/></body>' for i in range(90))
    return (f'<mujoco><option timestep="0.004" solver="PGS" iterations="30"/><worldbody>'
            f'<geom type="plane" size="5 5 .01"/>{bodies}</worldbody></mujoco>')


# mocap bodies: a gripper-like box welded to a mocap target that the control array moves
# (mjSTATE_MOCAP_POS | mjSTATE_MOCAP_QUAT in the control spec), a second mocap body carrying a
# collision geom that pushes a free sphere around
MOCAP_XML = """
<mujoco>
  <option timestep="0.004" solver="PGS" iterations="50"/>
  <worldbody>
    <geom type="plane" size="3 3 .01"/>
    <body name="target" mocap="true" pos="0 0 .4" quat="1 0 0 0"><geom type="sphere" size=".02" contype="0" conaffinity="0"/></body>
    <body name="paddle" mocap="true" pos=".6 0 .06" euler="0 0 20"><geom type="box" size=".02 .2 .05"/></body>
    <body name="hand" pos="0 0 .35"><freejoint/><geom type="box" size=".05 .04 .03"/>
      <body pos="0 0 -.06"><joint name="finger" type="slide" axis="0 1 0" range="-.03 .03" limited="true" damping="1"/><geom type="capsule" size=".01 .02"/></body></body>
    <body pos=".75 .05 .05"><freejoint/><geom type="sphere" size=".05" condim="3"/></body>
  </worldbody>
  <equality><weld body1="hand" body2="target" solref=".02 1"/></equality>
  <actuator><position joint="finger" kp="20"/></actuator>
</mujoco>
"""


# predefined contact <pair>s: one that replaces the automatic sphere-plane pair with its own
# condim / friction / solref / margin, one between parent and child (filtered automatically, active
# as a pair), one with solreffriction (used by elliptic friction rows) and an <exclude>
PAIR_XML = """
<mujoco>
  <option timestep="0.004" solver="PGS" iterations="50"/>
  <worldbody>
    <geom name="floor" type="plane" size="3 3 .01"/>
    <body name="s1" pos="0 0 .06"><freejoint/><geom name="gs1" type="sphere" size=".05"/></body>
    <body name="s2" pos=".3 0 .06"><freejoint/><geom name="gs2" type="sphere" size=".05" condim="3"/></body>
    <body name="c1" pos="-.4 0 .3">
      <joint name="h1" axis="0 1 0" damping=".05"/><geom name="gc1" type="capsule" fromto="0 0 0 .2 0 0" size=".03"/>
      <body name="c2" pos=".2 0 0"><joint name="h2" axis="0 1 0" range="-150 150" limited="true" damping=".05"/>
        <geom name="gc2" type="capsule" fromto="0 0 0 -.15 0 .05" size=".03"/></body>
    </body>
    <body name="b1" pos=".8 0 .051"><freejoint/><geom name="gb1" type="box" size=".06 .06 .05"/></body>
    <body name="s3" pos=".8 0 .16"><freejoint/><geom name="gs3" type="sphere" size=".05"/></body>
    <body name="s4" pos="1.2 0 .05"><freejoint/><geom name="gs4" type="sphere" size=".05"/></body>
  </worldbody>
  <contact>
    <pair geom1="floor" geom2="gs1" condim="4" friction=".7 .6 .02 .001 .001" solref=".015 1.2" margin=".01" gap=".002"/>
    <pair geom1="gc2" geom2="gc1" condim="1" solref=".01 1"/>
    <pair geom1="gs3" geom2="gb1" condim="3" friction="1.1 .9 .01 .001 .001" solreffriction=".03 .8"/>
    <exclude body1="s4" body2="b1"/>
  </contact>
</mujoco>
"""


# fluid forces (inertia-box model): a 3-link swimmer-like chain and a tumbling free box in a dense,
# viscous medium with wind
FLUID_XML = """
<mujoco>
  <option timestep="0.004" solver="PGS" iterations="40" density="3000" viscosity=".2" wind=".3 -.1 .05" integrator="RK4"/>
  <default><geom type="capsule" size=".04" condim="1"/><joint damping=".02"/></default>
  <worldbody>
    <geom type="plane" size="4 4 .01" pos="0 0 -1"/>
    <body name="head" pos="0 0 0">
      <joint type="slide" axis="1 0 0"/><joint type="slide" axis="0 1 0"/><joint type="hinge" axis="0 0 1"/>
      <geom fromto="0 0 0 -.3 0 0"/>
      <body pos="-.3 0 0"><joint name="r1" axis="0 0 1" range="-100 100" limited="true"/><geom fromto="0 0 0 -.3 0 0"/>
        <body pos="-.3 0 0"><joint name="r2" axis="0 0 1" range="-100 100" limited="true"/><geom fromto="0 0 0 -.3 0 0" size=".03"/></body></body>
    </body>
    <body pos="1 1 0" euler="20 30 40"><freejoint/><geom type="box" size=".1 .05 .2" density="700"/></body>
  </worldbody>
  <actuator><motor joint="r1" gear="3"/><motor joint="r2" gear="3"/></actuator>
</mujoco>
"""


# the same medium with the ELLIPSOID fluid model on some geoms (fluidshape = ellipsoid: added mass, Kutta and Magnus lift,
# blunt / slender / angular drag with non-default coefficients) next to inertia-box bodies: capsules, a cylinder, a box,
# an ellipsoid and a sphere; one body mixes a fluid geom with a plain one (the whole body then leaves the inertia-box model)
ELLIPSOID_FLUID_XML = """
<mujoco>
  <option timestep="0.004" solver="PGS" iterations="40" density="1200" viscosity=".15" wind=".3 -.1 .05"/>
  <default><geom type="capsule" size=".04" condim="1"/><joint damping=".02"/></default>
  <worldbody>
    <geom type="plane" size="4 4 .01" pos="0 0 -1"/>
    <body name="head" pos="0 0 0">
      <joint type="slide" axis="1 0 0"/><joint type="slide" axis="0 1 0"/><joint type="hinge" axis="0 0 1"/>
      <geom fromto="0 0 0 -.3 0 0" fluidshape="ellipsoid"/>
      <body pos="-.3 0 0"><joint name="r1" axis="0 0 1" range="-100 100" limited="true"/>
        <geom fromto="0 0 0 -.3 0 0" fluidshape="ellipsoid" fluidcoef=".6 .3 1.2 .8 .9"/><geom type="sphere" size=".05" pos="-.15 0 .06"/>
        <body pos="-.3 0 0"><joint name="r2" axis="0 0 1" range="-100 100" limited="true"/><geom fromto="0 0 0 -.3 0 0" size=".03"/></body></body>
    </body>
    <body pos="1 1 0" euler="20 30 40"><freejoint/><geom type="box" size=".1 .05 .2" density="700" fluidshape="ellipsoid"/></body>
    <body pos="-1 1 0" euler="50 10 0"><freejoint/><geom type="ellipsoid" size=".1 .06 .03" density="500" fluidshape="ellipsoid" fluidcoef=".5 .25 1.5 1 1"/>
      <body pos=".15 0 0"><joint axis="0 1 0"/><geom type="cylinder" size=".03 .08" fluidshape="ellipsoid"/></body></body>
    <body pos="0 -1 0"><freejoint/><geom type="sphere" size=".07" density="300" fluidshape="ellipsoid"/></body>
  </worldbody>
  <actuator><motor joint="r1" gear="3"/><motor joint="r2" gear="3"/></actuator>
</mujoco>
"""


# cameras in every mode of mj_camlight (fixed, track, trackcom, targetbody, targetbodycom) with frame sensors attached to
# them / referenced to them, and sites projected into their images (pinhole by fovy, and by intrinsics + sensor size)
CAMERA_XML = """
<mujoco>
  <option timestep="0.004" solver="PGS" iterations="50"/>
  <worldbody>
    <geom type="plane" size="3 3 .01"/>
    <camera name="world_cam" pos="0 -2 1.5" xyaxes="1 0 0 0 .6 .8" fovy="50" resolution="640 480"/>
    <body name="base" pos="0 0 .8">
      <joint name="j0" axis="0 0 1"/><geom type="capsule" fromto="0 0 0 .3 0 0" size=".03"/>
      <camera name="fixed_cam" pos=".1 0 .1" euler="10 20 30" fovy="40" resolution="320 240"/>
      <camera name="track_cam" mode="track" pos="0 -1 .5" xyaxes="1 0 0 0 0 1"/>
      <camera name="trackcom_cam" mode="trackcom" pos="0 -1.5 .7" xyaxes="1 0 0 0 .5 1"/>
      <body name="arm" pos=".3 0 0">
        <joint name="j1" axis="0 1 0"/><geom type="capsule" fromto="0 0 0 .25 0 0" size=".025"/>
        <site name="tip" pos=".25 0 0"/>
        <camera name="target_cam" mode="targetbody" target="ball" pos="0 0 .2"/>
        <camera name="intr_cam" pos=".1 .05 .05" euler="0 90 0" resolution="800 600" sensorsize=".0036 .0027" focal=".004 .004"/>
      </body>
    </body>
    <body name="ball" pos=".6 .4 .5"><freejoint/><geom type="sphere" size=".05"/><site name="ballsite"/>
      <camera name="targetcom_cam" mode="targetbodycom" target="base" pos="0 0 .1"/></body>
  </worldbody>
  <actuator><motor joint="j0" gear="2"/><motor joint="j1" gear="2"/></actuator>
  <sensor>
    <framepos objtype="camera" objname="fixed_cam"/><framequat objtype="camera" objname="fixed_cam"/>
    <framexaxis objtype="camera" objname="track_cam"/><framepos objtype="camera" objname="track_cam"/>
    <framepos objtype="camera" objname="trackcom_cam"/><framezaxis objtype="camera" objname="trackcom_cam"/>
    <framepos objtype="camera" objname="target_cam"/><framexaxis objtype="camera" objname="target_cam"/>
    <frameyaxis objtype="camera" objname="target_cam"/><framezaxis objtype="camera" objname="target_cam"/>
    <framezaxis objtype="camera" objname="targetcom_cam"/><framepos objtype="camera" objname="targetcom_cam"/>
    <framelinvel objtype="camera" objname="fixed_cam"/><frameangvel objtype="camera" objname="target_cam"/>
    <framelinacc objtype="camera" objname="fixed_cam"/><frameangacc objtype="camera" objname="intr_cam"/>
    <framepos objtype="site" objname="ballsite" reftype="camera" refname="fixed_cam"/>
    <framequat objtype="body" objname="ball" reftype="camera" refname="target_cam"/>
    <framelinvel objtype="site" objname="tip" reftype="camera" refname="world_cam"/>
    <camprojection site="ballsite" camera="world_cam"/><camprojection site="tip" camera="fixed_cam"/>
    <camprojection site="ballsite" camera="intr_cam"/><camprojection site="ballsite" camera="target_cam"/>
    <camprojection site="tip" camera="trackcom_cam"/>
  </sensor>
</mujoco>
"""


# three separate trees, each tied to the world by its own equality: three constraint islands made
# of equality rows only (the primal solvers visit them one after the other)
ISLANDS_XML = """
<mujoco>
  <option timestep="0.002"/>
  <worldbody>
    <site name="w1" pos="0 0 1"/><site name="w2" pos="1 0 1"/>
    <body name="b1" pos="0 0 .8"><freejoint/><geom type="capsule" size=".03 .1"/><site name="s1" pos="0 0 .15"/></body>
    <body name="b2" pos="1 0 .8"><freejoint/><geom type="sphere" size=".06"/><site name="s2" pos="0 .05 .1" euler="10 20 0"/></body>
    <body name="b3" pos="2 0 .8"><freejoint/><geom type="capsule" size=".03 .1"/></body>
  </worldbody>
  <equality>
    <connect site1="s1" site2="w1"/>
    <weld site1="s2" site2="w2" solref=".01 1"/>
    <connect body1="b3" anchor="0 0 .2"/>
  </equality>
</mujoco>
"""


# spatial tendons through sites: a two-body spring-damper tendon with a length limit, a pulley
# tendon (two branches, divisor 2) between three bodies, a tendon that ends on a world site, and a
# tendon equality coupling a spatial and a fixed tendon
TENDON_XML = """
<mujoco>
  <option timestep="0.003" solver="PGS" iterations="50"/>
  <default><geom type="capsule" size=".02" condim="1"/><joint damping=".05"/></default>
  <worldbody>
    <geom type="plane" size="3 3 .01" pos="0 0 -1"/>
    <site name="anchor" pos="0 0 1"/><site name="anchor2" pos=".8 0 1"/>
    <body pos="0 0 .6"><joint name="p1" axis="0 1 0"/><joint name="p1b" axis="1 0 0"/><geom fromto="0 0 0 0 0 -.3"/><site name="a1" pos=".02 0 -.1"/>
      <body pos="0 0 -.3"><joint name="p2" axis="0 1 0"/><geom fromto="0 0 0 .25 0 0"/><site name="a2" pos=".2 0 .02"/><site name="a2b" pos=".1 .02 0"/></body></body>
    <body pos=".8 0 .5"><freejoint/><geom type="sphere" size=".06"/><site name="f1" pos="0 0 .06"/><site name="f1b" pos=".05 0 0"/></body>
    <body pos=".4 .5 .5"><joint name="s1" type="slide" axis="0 0 1"/><geom type="sphere" size=".05"/><site name="m1"/></body>
  </worldbody>
  <tendon>
    <spatial name="sp1" stiffness="20" damping=".5" springlength=".25" range="0 .55" limited="true"><site site="a1"/><site site="a2"/></spatial>
    <spatial name="sp2" stiffness="40" damping="1"><site site="anchor2"/><site site="f1"/></spatial>
    <spatial name="pul" stiffness="15"><site site="anchor"/><site site="a2b"/><pulley divisor="2"/><site site="anchor"/><site site="m1"/><pulley divisor="2"/><site site="anchor2"/><site site="f1b"/></spatial>
    <fixed name="fx"><joint joint="s1" coef="1"/></fixed>
  </tendon>
  <equality><tendon tendon1="sp1" tendon2="fx" polycoef=".3 .5 0 0 0" solref=".02 1"/></equality>
  <actuator><motor joint="p1" gear="1"/><motor joint="p2" gear=".5"/>
    <position tendon="sp1" kp="10" kv=".2"/><motor tendon="pul" gear="2"/><general tendon="fx" dyntype="filter" dynprm=".05" gainprm="3"/></actuator>
  <sensor><tendonpos tendon="sp1"/><tendonvel tendon="pul"/><tendonactuatorfrc tendon="sp1"/><tendonactuatorfrc tendon="pul"/><tendonlimitfrc tendon="sp1"/></sensor>
</mujoco>
"""


# tendons wrapping around a sphere and a cylinder (mju_wrap), with and without side sites, one side site INSIDE its sphere
# (the inside wrap: Newton iteration on asin terms), a pulley branch whose second half wraps, springs / limits / an actuator
WRAP_XML = """
<mujoco>
  <option timestep="0.002" solver="PGS" iterations="50"/>
  <default><geom type="capsule" size=".02" contype="0" conaffinity="0"/><joint damping=".05"/></default>
  <worldbody>
    <site name="w0" pos="-.35 .02 .75"/><site name="w1" pos=".5 0 .9"/><site name="w2" pos="-.1 .3 .8"/>
    <body pos="0 0 .6"><joint name="a" axis="0 1 0"/><joint name="a2" axis="1 0 0"/><geom fromto="0 0 0 .3 0 0"/>
      <geom name="ws" type="sphere" size=".07" pos=".12 0 0"/><site name="ss" pos=".12 0 .09"/><site name="sin" pos=".13 .01 -.02"/>
      <geom name="wc" type="cylinder" size=".05 .1" pos=".25 0 0" euler="90 10 0"/><site name="sc" pos=".25 0 -.08"/>
      <body pos=".3 0 0"><joint name="b" axis="0 1 0"/><joint name="b2" axis="0 0 1"/><geom fromto="0 0 0 .25 0 0"/>
        <site name="e1" pos=".2 .01 .03"/><site name="e2" pos=".1 -.02 -.03"/><site name="e3" pos=".22 0 -.02"/>
        <geom name="ws2" type="sphere" size=".04" pos=".12 0 0"/></body></body>
    <body pos=".6 .3 .7"><freejoint/><geom type="sphere" size=".05"/><site name="f" pos="0 0 .05"/></body>
  </worldbody>
  <tendon>
    <spatial name="t_sphere" stiffness="30" damping=".3" springlength=".5"><site site="w0"/><geom geom="ws" sidesite="ss"/><site site="e1"/></spatial>
    <spatial name="t_cyl" stiffness="20" range="0 .9" limited="true"><site site="w0"/><geom geom="wc" sidesite="sc"/><site site="e2"/></spatial>
    <spatial name="t_noside" stiffness="25"><site site="w2"/><geom geom="ws2"/><site site="f"/></spatial>
    <spatial name="t_inside" stiffness="15"><site site="w0"/><geom geom="ws" sidesite="sin"/><site site="e3"/></spatial>
    <spatial name="t_pulley" stiffness="10"><site site="w1"/><site site="f"/><pulley divisor="2"/><site site="w1"/><geom geom="wc"/><site site="e1"/></spatial>
  </tendon>
  <actuator><motor joint="a" gear="2"/><motor joint="b" gear="1"/><motor tendon="t_sphere" gear="3"/><position tendon="t_cyl" kp="5"/></actuator>
  <sensor><tendonpos tendon="t_sphere"/><tendonvel tendon="t_cyl"/><tendonpos tendon="t_inside"/></sensor>
</mujoco>
"""


# actuator groups disabled through opt.disableactuator (mj_actuatorDisabled: no force, the activation frozen at integration
# time) and tendons that limit the TOTAL force of the actuators acting on them (engine_forward.c:880-915), next to the
# per-actuator force range
ACT_GROUP_XML = """
<mujoco>
  <option timestep="0.004" solver="PGS" iterations="30" actuatorgroupdisable="1 3"/>
  <default><geom type="capsule" size=".03" contype="0" conaffinity="0"/><joint damping=".2"/></default>
  <worldbody>
    <body pos="0 0 1"><joint name="j1" axis="0 1 0"/><geom fromto="0 0 0 .3 0 0"/><site name="s1" pos=".15 0 .05"/>
      <body pos=".3 0 0"><joint name="j2" axis="0 1 0"/><geom fromto="0 0 0 .25 0 0"/><site name="s2" pos=".12 0 .05"/>
        <body pos=".25 0 0"><joint name="j3" axis="0 0 1"/><geom fromto="0 0 0 .2 0 0"/><site name="s3" pos=".1 0 .05"/></body></body></body>
  </worldbody>
  <tendon>
    <spatial name="sp" actuatorfrclimited="true" actuatorfrcrange="-.6 .4"><site site="s1"/><site site="s2"/><site site="s3"/></spatial>
    <fixed name="fx" actuatorfrclimited="true" actuatorfrcrange="0 .5"><joint joint="j1" coef=".4"/><joint joint="j3" coef="-.7"/></fixed>
    <fixed name="free"><joint joint="j2" coef="1"/></fixed>
  </tendon>
  <actuator>
    <motor joint="j1" gear="2" group="0"/><position joint="j2" kp="8" kv=".3" group="1"/>
    <intvelocity joint="j3" kp="6" actrange="-1 1" group="3"/><general joint="j1" dyntype="filter" dynprm=".05" gainprm="2" group="3" forcelimited="true" forcerange=".1 .3"/>
    <general joint="j2" dyntype="integrator" gainprm="1.5" group="2" actlimited="true" actrange="-.5 .5"/>
    <motor tendon="sp" gear="1"/><motor tendon="sp" gear="-.5" forcelimited="true" forcerange="-.2 .2"/><motor tendon="sp" gear="2" group="1"/>
    <motor tendon="fx" gear="1.5"/><motor tendon="fx" gear="1"/><motor tendon="free" gear="1"/>
  </actuator>
  <sensor><tendonactuatorfrc tendon="sp"/><tendonactuatorfrc tendon="fx"/></sensor>
</mujoco>
"""


# muscle actuators (activation dynamics with two time constants, hard and smoothed switching; FLV gain; passive force) on a
# wrapping tendon, a fixed tendon and a joint; peak force given directly and through scale / acc0
MUSCLE_XML = """
<mujoco>
  <option timestep="0.002" solver="PGS" iterations="30"/>
  <default><geom type="capsule" size=".02" contype="0" conaffinity="0"/><joint damping=".3" armature=".01"/></default>
  <worldbody>
    <site name="o1" pos="-.05 0 1.05"/><site name="o2" pos=".05 0 .95"/>
    <body pos="0 0 1"><joint name="sh" axis="0 1 0" range="-60 80"/><geom fromto="0 0 0 .3 0 0"/>
      <geom name="wrap" type="cylinder" size=".04 .05" pos="0 0 0" euler="90 0 0"/><site name="side" pos="0 0 .08"/>
      <site name="i1" pos=".15 0 .03"/><site name="i2" pos=".2 0 -.03"/>
      <body pos=".3 0 0"><joint name="el" axis="0 1 0" range="0 120"/><geom fromto="0 0 0 .25 0 0"/><site name="i3" pos=".08 0 .03"/></body></body>
  </worldbody>
  <tendon>
    <spatial name="flex"><site site="o1"/><geom geom="wrap" sidesite="side"/><site site="i1"/></spatial>
    <spatial name="ext"><site site="o2"/><site site="i2"/><site site="i3"/></spatial>
    <fixed name="fx"><joint joint="el" coef=".05"/><joint joint="sh" coef="-.02"/></fixed>
  </tendon>
  <actuator>
    <muscle name="m_flex" tendon="flex" ctrllimited="true" ctrlrange="0 1" force="120"/>
    <muscle name="m_ext" tendon="ext" ctrllimited="true" ctrlrange="0 1" scale="300" timeconst=".02 .06" tausmooth=".3" range=".7 1.1" lmin=".4" lmax="1.7" vmax="2" fpmax="1.5" fvmax="1.3"/>
    <muscle name="m_fx" tendon="fx" ctrllimited="true" ctrlrange="0 1" force="40" tausmooth=".05"/>
    <muscle name="m_jnt" joint="el" ctrllimited="true" ctrlrange="0 1" force="15" lengthrange="0 2"/>
  </actuator>
</mujoco>
"""


# site transmissions without a reference site: Cartesian force / torque actuators (a thruster on a
# free body, forces and torques at an arm's tip), one of them with filter dynamics
SITE_ACT_XML = """
<mujoco>
  <option timestep="0.004" solver="PGS" iterations="40"/>
  <default><geom type="capsule" size=".03" condim="3"/><joint damping=".1"/></default>
  <worldbody>
    <geom type="plane" size="3 3 .01"/>
    <body pos="0 0 .6"><joint name="j1" axis="0 1 0"/><geom fromto="0 0 0 .25 0 0"/>
      <body pos=".25 0 0"><joint name="j2" axis="0 0 1"/><geom fromto="0 0 0 .2 0 0"/><site name="tip" pos=".2 0 0" euler="0 30 10"/></body></body>
    <body pos="-.5 0 .3"><freejoint/><geom type="box" size=".1 .05 .03"/><site name="thr" pos=".05 0 0" euler="10 0 40"/></body>
  </worldbody>
  <actuator>
    <motor site="tip" gear="0 0 1 0 0 0"/><motor site="tip" gear="0 0 0 0 .5 0"/>
    <general site="thr" gear=".6 0 1.5 0 0 .1" dyntype="filter" dynprm=".03"/><position site="thr" gear="0 1 0 0 0 0" kp="2" kv=".5"/>
  </actuator>
</mujoco>
"""


# actuators on ball and free joints (3D / 6D gears, joint and jointinparent transmissions)
BALL_ACT_XML = """<mujoco>
  <option timestep="0.004" solver="PGS" iterations="40"/>
  <default><geom type="capsule" size=".03" condim="3"/><joint damping=".1"/></default>
  <worldbody>
    <geom type="plane" size="3 3 .01"/>
    <body pos="0 0 .8"><joint name="b1" type="ball"/><geom fromto="0 0 0 .25 0 0"/>
      <body pos=".25 0 0"><joint name="b2" type="ball" range="0 60" limited="true"/><geom fromto="0 0 0 .2 0 .05"/></body></body>
    <body pos="-.5 0 .3" euler="10 20 30"><joint name="fr" type="free"/><geom type="box" size=".1 .05 .03"/></body>
  </worldbody>
  <actuator>
    <motor joint="b1" gear=".5 .2 -.3"/><motor joint="b2" gear="0 .4 .1"/>
    <general joint="b2" gear=".3 0 .2" gaintype="affine" gainprm="1 .2 -.1" biastype="affine" biasprm=".1 -1 -.2"/>
    <motor joint="fr" gear="1 0 2 0 .1 .05"/>
    <motor jointinparent="b1" gear="0 .2 .4"/><general jointinparent="fr" gear="0 0 .5 .1 0 .2" dyntype="filter" dynprm=".05"/>
  </actuator>
</mujoco>
"""


# geom surface velocities: a spinning floor (turntable), a conveyor belt box and a box that drives
# itself; sliding, torsional and rolling contacts on them
SURFACEVEL_XML = """<mujoco>
  <option timestep="0.004" solver="PGS" iterations="50"/>
  <worldbody>
    <geom name="floor" type="plane" size="4 4 .01" surfacevel="0 0 0 0 0 1.5"/>
    <geom name="belt" type="box" size="1 .3 .05" pos="0 1.5 .05" surfacevel=".6 0 0 0 0 0" friction="1 .01 .001"/>
    <body pos="-.5 1.5 .151"><freejoint/><geom type="box" size=".08 .06 .05" condim="3"/></body>
    <body pos="0 1.45 .16"><freejoint/><geom type="sphere" size=".06" condim="4" friction=".8 .02 .001"/></body>
    <body pos=".5 .2 .06"><freejoint/><geom type="sphere" size=".06" condim="6"/></body>
    <body pos="-.6 -.3 .051"><freejoint/><geom type="box" size=".1 .1 .05" condim="4" surfacevel="0 .2 0 0 0 0"/></body>
    <body pos="0 -.8 .04" euler="90 0 0"><freejoint/><geom type="capsule" size=".04 .1" condim="3"/></body>
  </worldbody>
</mujoco>
"""


# contact adhesion (geom `adhesion`, pair `adhesion`): a magnetic door catch working across its gap band, adhesive
# spheres on / above the floor (pyramid and ellipse rows biased by mj_adhesionRef, a tether row in the gap), the
# higher priority taking its own adhesion alone, a predefined pair with its own adhesion
ADHESION_XML = """<mujoco>
  <option timestep="0.004" solver="PGS" iterations="50"/>
  <worldbody>
    <geom name="floor" type="plane" size="4 4 .01"/>
    <geom name="strike" type="box" size=".005 .01 .3" pos="-.015 .19 .4" adhesion="1.5" gap=".003"/>
    <body name="door" pos="0 .45 .4">
      <joint name="hinge" axis="0 0 1" pos=".02 .25 0" damping=".02"/>
      <geom type="box" size=".01 .25 .3" mass="1"/>
      <geom name="catch" type="box" size=".005 .01 .3" pos="-.005 -.251 0" adhesion="1.5" gap=".003" mass=".05"/>
    </body>
    <body pos="1 0 .0505"><freejoint/><geom name="sticky" type="sphere" size=".05" adhesion="4" gap=".02" condim="3"/></body>
    <body pos="1 .5 .062"><freejoint/><geom name="hover" type="sphere" size=".05" adhesion=".3" gap=".03" mass=".02"/></body>
    <body pos="1 1 .049"><freejoint/><geom name="prio" type="capsule" size=".05 .1" euler="90 0 0" adhesion="2" priority="1" condim="4" gap=".01"/></body>
    <body pos="-1 0 .3"><freejoint/><geom name="pa" type="sphere" size=".06"/></body>
    <body pos="-1 0 .1"><freejoint/><geom name="pb" type="box" size=".1 .1 .1" adhesion=".5"/></body>
    <body name="grip" pos="1 1.5 .07"><freejoint/><geom type="sphere" size=".05" gap=".03" condim="3"/></body>
    <body name="grip4" pos="1 2 .05"><freejoint/><geom type="box" size=".05 .06 .05" condim="4" gap=".01"/></body>
  </worldbody>
  <contact>
    <pair geom1="pa" geom2="pb" adhesion="3" gap=".02" condim="3"/>
  </contact>
  <!-- adhesion actuators (body transmission): the moment averages the normal Jacobians of the body's contacts, active
       (through the rows of efc_J) and in the gap -->
  <actuator>
    <adhesion body="grip" ctrlrange="0 1" gain="3"/>
    <adhesion body="grip4" ctrlrange="0 1" gain="4"/>
    <adhesion body="door" ctrlrange="0 1" gain="1"/>
  </actuator>
</mujoco>
"""


# geoms that touch EXACTLY (to the last bit): the reference's sweep-and-prune rounds the sweep-axis
# end points to float and breaks ties by array position (engine_collision_driver.c:1456-1470), so
# whether such a body pair reaches the narrowphase depends on body order; plus stacked spheres on a
# plane and a pair that overlaps by one ulp
TOUCH_XML = """
<mujoco>
  <option timestep="0.002" solver="PGS" iterations="30" gravity="0 0 0"/>
  <worldbody>
    <body pos="0 0 1"><freejoint/><geom type="sphere" size=".1"/></body>
    <body pos=".2 0 1"><freejoint/><geom type="sphere" size=".1"/></body>
    <body pos=".4 0 1"><freejoint/><geom type="sphere" size=".1"/></body>
    <body pos="1 0 1"><freejoint/><geom type="sphere" size=".125"/></body>
    <body pos="1 .25 1"><freejoint/><geom type="sphere" size=".125"/></body>
    <body pos="1 .25 1.25"><freejoint/><geom type="sphere" size=".125"/></body>
    <body pos="-1 0 1"><freejoint/><geom type="capsule" size=".05 .1"/></body>
    <body pos="-1 .1 1"><freejoint/><geom type="capsule" size=".05 .1"/></body>
    <body pos="-1 .2000000000000001 1"><freejoint/><geom type="sphere" size=".05"/></body>
  </worldbody>
</mujoco>
"""

# bodies with several geoms of mixed types whose BVH order differs from their geom order: the
# midphase (mj_collideTree) visits leaf pairs in its own order, culls with oriented boxes, and the
# contacts of a body pair are re-sorted afterwards (engine_collision_driver.c:686-716, :996-1240)
MULTIGEOM_XML = """
<mujoco>
  <option timestep="0.003" solver="PGS" iterations="50"/>
  <default><geom condim="3"/></default>
  <worldbody>
    <geom type="plane" size="3 3 .01"/>
    <body name="a" pos="0 0 .3" euler="10 20 0"><freejoint/>
      <geom type="sphere" size=".06" pos=".25 0 0"/>
      <geom type="capsule" size=".03" fromto="-.25 0 0 .2 0 0"/>
      <geom type="sphere" size=".05" pos="-.25 .1 0"/>
      <geom type="capsule" size=".025" fromto="0 -.2 0 0 .2 0"/>
      <geom type="sphere" size=".04" pos="0 0 .15"/>
    </body>
    <body name="b" pos=".05 .05 .55" euler="0 -15 40"><freejoint/>
      <geom type="capsule" size=".03" fromto="0 0 -.2 0 0 .2"/>
      <geom type="sphere" size=".07" pos="0 .2 0"/>
      <geom type="sphere" size=".05" pos="0 -.2 .1"/>
      <geom type="capsule" size=".02" fromto="-.2 0 0 .2 0 .05"/>
    </body>
    <body name="c" pos="-.1 0 .85" euler="30 0 10"><freejoint/>
      <geom type="sphere" size=".08"/>
      <geom type="capsule" size=".03" fromto="-.2 0 0 .2 0 0"/>
      <geom type="capsule" size=".03" fromto="0 -.2 0 0 .2 0"/>
    </body>
  </worldbody>
</mujoco>
"""
