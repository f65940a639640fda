"""CPU: the GJK / EPA / multicontact narrowphase (mujoco_amd/csrc/mjh_convex.h) on the host wavefront
emulation against the compiled reference, bit for bit.

Reference: mjc_Convex (engine_collision_convex.c:881), mjc_PlaneConvex (:1004), mjc_ccd
(engine_collision_gjk.c:2318).  The GPU counterparts are in tests/test_gpu_parity.py.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from convex_scenes import MESH_PAIRS, PRIMITIVE_PAIRS, scene, sweep
from mujoco_amd import _capi as K


@pytest.mark.parametrize("pair", PRIMITIVE_PAIRS, ids=lambda p: "-".join(p))
def test_primitive_convex_pairs_bit_exact(rb, hostsim_lib, tmp_path, pair):
    """the primitive cells of mjCOLLISIONFUNC that the reference routes to GJK/EPA (cylinder-box,
    capsule-cylinder, ellipsoid-anything, plane-ellipsoid): contact lists identical over a pose sweep,
    without margin (one contact, or the perturbation multi-contact of mjc_Convex: up to 5) and with"""
    total = {}
    for margin in (0.0, 0.02):
        xml = tmp_path / "s.xml"
        xml.write_text(scene(pair[0], pair[1], margin))
        hist, bad = sweep(rb, K, hostsim_lib, xml, 60, seed=int(margin*100))
        assert bad == 0, (pair, margin, hist)
        for n, c in hist.items(): total[n] = total.get(n, 0) + c
    assert sum(c for n, c in total.items() if n > 0) >= 30, total


@pytest.mark.parametrize("pair", MESH_PAIRS, ids=lambda p: "-".join(p))
def test_mesh_convex_pairs_bit_exact(rb, hostsim_lib, tmp_path, pair):
    """convex meshes: exhaustive and hill-climbing support (mjc_meshSupport / mjc_hillclimbSupport),
    EPA on discrete geoms (repeated-support termination), multicontact face / edge clipping against
    meshes and boxes (polygonClip, polygonQuad), plane-mesh vertex contacts; random, axis-aligned and
    nearly aligned poses (the degenerate simplices polytope2/3/4 have to repair)"""
    xml = tmp_path / "s.xml"
    any_multi = 0
    for margin, aligned in ((0.0, False), (0.02, False), (0.0, True)):
        xml.write_text(scene(pair[0], pair[1], margin, aligned))
        hist, bad = sweep(rb, K, hostsim_lib, xml, 50, seed=3, aligned=aligned)
        assert bad == 0, (pair, margin, aligned, hist)
        any_multi += sum(c for n, c in hist.items() if n > 1)
    if pair[0] not in ("sph", "ell") and pair[1] not in ("sph", "ell"):
        assert any_multi > 0


def test_cube_3x3x3_single_steps_vs_golden(hostsim_lib):
    """BASELINE config 4 (model/cube/cube_3x3x3.xml: 26 mesh cubelets, Newton, implicitfast, nv = 66,
    ~150 mesh-mesh contacts): single steps from (state, warm start, ctrl) of the committed reference
    trajectory reproduce the reference's next state, contact / row counts and Newton iteration count"""
    fx = np.load(os.path.join(GOLDEN, "cube_3x3x3_steps.npz"))
    mm = K.MjbModel(hostsim_lib, os.path.join(GOLDEN, "cube_3x3x3.mjb"))
    dm = K.DeviceModel(hostsim_lib, mm)
    assert dm.size("ccd_any") == 1 and dm.size("nconmax") >= 300
    idx = [0, 1, 2, 5, 17, 40, 77, 120, 159]
    b = K.Batch(dm, len(idx))
    out = b.rollout_host(1, K.mjSTATE_CTRL, fx["state"][idx], fx["warmstart"][idx], fx["ctrl"][idx][:, None])
    assert b.get("warning").sum() == 0
    c = b.get("counts")
    assert np.array_equal(c[:, 0], fx["ints"][idx, 0])          # ncon
    assert np.array_equal(c[:, 1], fx["ints"][idx, 1])          # nefc
    assert np.array_equal(c[:, 5], fx["ints"][idx, 2])          # solver_niter
    ref = fx["next"][idx]
    assert np.max(np.abs(out[:, 0] - ref)/np.maximum(1.0, np.abs(ref))) <= 1e-9


def test_device_sincos_within_one_ulp():
    """mjh_sincos (mjh_math.h) -- what the kernels evaluate on the GPU instead of the device libm's
    sin / cos -- against 200-bit references: < 1 ulp from |x| ~ 1e-8 to 1e10 (mjMAXVAL) and right
    next to multiples of pi/2, where the three-piece Cody-Waite reduction has to carry ~60 extra bits.
    Its explicit fma calls make it bit-reproducible between host and device, so this CPU test pins the
    GPU's values too."""
    import ctypes
    mpmath = pytest.importorskip("mpmath")
    from conftest import HOSTSIM_LIB
    mpmath.mp.prec = 200
    f = ctypes.CDLL(HOSTSIM_LIB).mjh_test_sincos
    f.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-np.pi, np.pi, 1500), rng.uniform(-100, 100, 1000), rng.uniform(-1e6, 1e6, 500),
                         rng.uniform(-1e10, 1e10, 500), rng.uniform(-1e-8, 1e-8, 200),
                         np.arange(-200, 200)*(np.pi/2) + rng.normal(0, 1e-9, 400)])
    s = np.zeros_like(xs); c = np.zeros_like(xs)
    f(len(xs), xs.ctypes.data, s.ctypes.data, c.ctypes.data)
    worst = 0.0
    for x, si, ci in zip(xs, s, c):
        rs, rc = mpmath.sin(mpmath.mpf(float(x))), mpmath.cos(mpmath.mpf(float(x)))
        worst = max(worst, float(abs(mpmath.mpf(float(si)) - rs)/mpmath.mpf(float(np.spacing(abs(float(rs)))))),
                    float(abs(mpmath.mpf(float(ci)) - rc)/mpmath.mpf(float(np.spacing(abs(float(rc)))))))
    assert worst < 1.0, worst
    assert np.abs(s - np.sin(xs)).max() <= 2.3e-16 and np.abs(c - np.cos(xs)).max() <= 2.3e-16


def test_device_atan2_exp_accuracy():
    """mjh_atan2 / mjh_exp (mjh_math.h) -- what the kernels evaluate on the GPU instead of the device library's
    atan2 (ball-joint limits, welds: mju_quat2Vel) and exp (filterexact actuators) -- against 200-bit references.
    Plain IEEE operations only, so the host build pins the GPU's values and oracle/devmath_shim.cc can hand the very
    same routines to the compiled reference."""
    import ctypes
    mpmath = pytest.importorskip("mpmath")
    from conftest import HOSTSIM_LIB
    mpmath.mp.prec = 200
    f = ctypes.CDLL(HOSTSIM_LIB).mjh_test_atan2_exp
    f.argtypes = [ctypes.c_int] + [ctypes.c_void_p]*4
    rng = np.random.default_rng(0)
    y = np.concatenate([rng.normal(0, 1, 3000), rng.uniform(-1e-3, 1e-3, 800), rng.normal(0, 1e6, 400), [0.0, 1.0, -1.0, 1e-300, 0.0, -0.0]])
    x = np.concatenate([rng.normal(0, 1, 3000), rng.uniform(-1, 1, 800), rng.normal(0, 1, 400), [1.0, 0.0, -0.0, -1.0, -1.0, -1.0]])
    xe = np.concatenate([rng.uniform(-20, 20, 3000), rng.uniform(-1, 1, 800), rng.uniform(-700, 700, 400), [0.0, 1e-30, -1e-30, 0.3, 709.0, -745.0]])
    at = np.zeros_like(y); ex = np.zeros_like(y)
    f(len(y), y.ctypes.data, x.ctypes.data, at.ctypes.data, ex.ctypes.data)
    worst = 0.0
    for yi, xi, ai in zip(y, x, at):
        if yi == 0:
            continue            # (mpmath has no signed zero; the +-0 cases are checked against the host libm below)
        r = mpmath.atan2(mpmath.mpf(float(yi)), mpmath.mpf(float(xi)))
        if r != 0:
            worst = max(worst, float(abs(mpmath.mpf(float(ai)) - r)/mpmath.mpf(float(np.spacing(abs(float(r)))))))
    assert worst < 1.5, worst
    assert np.abs(at - np.arctan2(y, x)).max() <= 9e-16
    f(len(y), y.ctypes.data, xe.ctypes.data, at.ctypes.data, ex.ctypes.data)
    worst = 0.0
    for xi, ei in zip(xe, ex):
        r = mpmath.exp(mpmath.mpf(float(xi)))
        if float(r) > 1e-300:
            worst = max(worst, float(abs(mpmath.mpf(float(ei)) - r)/mpmath.mpf(float(np.spacing(float(r))))))
    assert worst < 1.0, worst


def test_nearly_touching_boxes_keep_their_contacts(rb, hostsim_lib, tmp_path):
    """the oriented-box cull in front of GJK / EPA (mjh_collision.h: filter_obb) must not drop what the reference
    reports: GJK's distance is accurate to ccd_tolerance, not an upper bound, so two mesh boxes an edge apart by a
    nanometre come back as touching (found by the model sweep on stacked_boxes.xml).  Edge-to-edge and face-to-face
    placements with gaps from -1e-5 to +1e-3, plus tests/golden/obb_cull_states.npy: twelve slightly rotated edge-to-edge
    states (gaps 1e-10 .. 1e-6, found by random search) on which a cull without the tolerance band loses contacts the
    reference keeps.  Contact counts and records identical to the oracle's."""
    verts = "-1 -1 -1 1 -1 -1 1 1 -1 1 1 1 1 -1 1 -1 1 -1 -1 1 1 -1 -1 1"
    xml = tmp_path / "gap.xml"
    xml.write_text(f"""
<mujoco>
  <option gravity="0 0 0"/>
  <asset><mesh name="box" vertex="{verts}" scale=".05 .05 .05"/></asset>
  <worldbody>
    <body pos="0 0 0"><freejoint/><geom type="mesh" mesh="box"/></body>
    <body pos="0 -.1 .1"><freejoint/><geom type="mesh" mesh="box"/></body>
  </worldbody>
</mujoco>""")
    m = rb.MjModel.from_xml_path(str(xml))
    dm = K.DeviceModel(hostsim_lib, m)
    d = rb.MjData(m)
    gaps = [-1e-5, -1e-7, -1e-9, -1e-10, 0.0, 1e-10, 1e-9, 3e-9, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3]
    placements = [((0.0, -0.1, 0.1), (0.0, -1.0, 1.0)), ((0.0, 0.0, 0.1), (0.0, 0.0, 1.0))]     # edge-edge, face-face
    states = []
    for base, dirn in placements:
        for g in gaps:
            rb.mj_resetData(m, d)
            s = np.sqrt(sum(c*c for c in dirn))
            d.qpos[7:10] = [b + g*c/s for b, c in zip(base, dirn)]
            states.append(rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS))
    states += list(np.load(os.path.join(os.path.dirname(__file__), "golden", "obb_cull_states.npy")))
    b = K.Batch(dm, len(states))
    S = np.array(states)
    b.set("qpos", S[:, 1:1 + m.nq]); b.set("qvel", S[:, 1 + m.nq:])
    b.forward()
    cnt = b.get("counts")
    cd = b.get("con_dist"); cp = b.get("con_pos")
    total = 0
    for e, s in enumerate(states):
        rb.mj_setState(m, d, s, rb.mjSTATE_FULLPHYSICS)
        rb.mj_forward(m, d)
        assert cnt[e, 0] == d.ncon, (e, cnt[e, 0], d.ncon)
        if d.ncon:
            c = d.contact[:d.ncon]
            assert np.array_equal(cd[e][:d.ncon], c["dist"]) and np.array_equal(cp[e][:3*d.ncon], np.asarray(c["pos"]).ravel())
        total += d.ncon
    assert total > 0


def _mixed_convex_rounds(rb, lib, tmp_path, nstate=12):
    """single-contact convex pairs (sphere / capsule / ellipsoid against a mesh) in the SAME narrowphase round as polyhedral
    pairs (mesh : mesh, box : mesh), the curved pairs first in the list so that they are owned by the low lanes: the
    lane-parallel distance phase of the polyhedral pairs uses the record of the lane it runs on as scratch and must not
    run after those lanes' contacts are finished (found by the model sweep on test/user/testdata/discardvisual.xml, where a
    sphere : mesh pair read dist 1.2 instead of -0.199).  Contact lists identical to the oracle's over random poses."""
    xml = tmp_path / "mixed.xml"
    xml.write_text("""
<mujoco>
  <option gravity="0 0 0"/>
  <asset><mesh name="tet" vertex="0 0 0  1 0 0  0 1 0  0 0 1"/></asset>
  <worldbody>
    <geom name="sp" type="sphere" size=".2" pos=".3 0 .4"/>
    <geom name="cp" type="capsule" size=".1 .2" pos=".9 .1 -.3" euler="20 40 0"/>
    <geom name="el" type="ellipsoid" size=".15 .1 .25" pos=".5 .5 -.2"/>
    <body pos=".2 0 -.5"><geom name="m0" type="mesh" mesh="tet"/></body>
    <body pos=".1 .2 -.6"><geom name="b0" type="box" size=".3 .3 .3"/></body>
    <body pos=".3 0 -.5"><freejoint/><geom name="m1" type="mesh" mesh="tet"/></body>
  </worldbody>
</mujoco>""")
    m = rb.MjModel.from_xml_path(str(xml))
    dm = K.DeviceModel(lib, m)
    d = rb.MjData(m)
    rng = np.random.default_rng(12)
    states = []
    for k in range(nstate):
        rb.mj_resetData(m, d)
        if k:
            d.qpos[:3] += rng.normal(0, .08, 3)
            q = np.array([1, 0, 0, 0.0]) + rng.normal(0, .15, 4)
            d.qpos[3:7] = q/np.linalg.norm(q)
        states.append(rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS))
    S = np.array(states)
    b = K.Batch(dm, len(states))
    b.set("qpos", S[:, 1:1 + m.nq]); b.set("qvel", S[:, 1 + m.nq:])
    b.forward()
    cnt, cd, cg = b.get("counts"), b.get("con_dist"), b.get("con_geom")
    curved = poly = 0
    worst = 0.0
    for e, s in enumerate(states):
        rb.mj_setState(m, d, s, rb.mjSTATE_FULLPHYSICS)
        rb.mj_forward(m, d)
        assert cnt[e, 0] == d.ncon, (e, cnt[e, 0], d.ncon)
        c = d.contact[:d.ncon]
        assert np.array_equal(cg[e][:2*d.ncon], np.asarray(c["geom"]).ravel())
        worst = max(worst, float(np.max(np.abs(cd[e][:d.ncon] - c["dist"]))) if d.ncon else 0.0)
        curved += int(np.sum(np.asarray(c["geom"])[:, 0] < 3))
        poly += int(np.sum(np.asarray(c["geom"])[:, 0] >= 3))
    return worst, curved, poly


def test_curved_pairs_next_to_polyhedral_pairs_bit_exact(rb, hostsim_lib, tmp_path):
    worst, curved, poly = _mixed_convex_rounds(rb, hostsim_lib, tmp_path)
    assert curved >= 12 and poly >= 12
    assert worst == 0.0


def test_cube_steps_in_soa_layout_and_without_lds_plan(hostsim_lib):
    """the convex narrowphase's row workspaces live in the LDS-planned field `ccd_row` -- or, when the plan has no room
    for it (small budgets) or the batch is laid out SoA-across-environments (its fields are strided, a row workspace is
    not), in the environment's block of the global buffer `ccd_ws`: same results either way"""
    fx = np.load(os.path.join(GOLDEN, "cube_3x3x3_steps.npz"))
    mm = K.MjbModel(hostsim_lib, os.path.join(GOLDEN, "cube_3x3x3.mjb"))
    dm = K.DeviceModel(hostsim_lib, mm)
    idx = [0, 5, 40]
    outs = []
    for layout, budget in (("aos", 20480), ("aos", 8192), ("soa", 20480)):
        b = K.Batch(dm, len(idx), layout=layout)
        b.plan_lds(budget)
        in_lds = any("ccd_row" in ln and "lds@" in ln for ln in b.lds_report().splitlines())
        assert in_lds == (budget == 20480 and layout == "aos") or layout == "soa"
        out = b.rollout_host(1, K.mjSTATE_CTRL, fx["state"][idx], fx["warmstart"][idx], fx["ctrl"][idx][:, None])
        assert b.get("warning").sum() == 0
        assert np.array_equal(b.get("counts")[:, 0], fx["ints"][idx, 0])
        outs.append(out)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    assert np.array_equal(outs[0][:, 0], fx["next"][idx])
