"""The model sweep on the device, from committed fixtures (no live oracle, no reference tree).

tests/golden/sweep holds every model file of the reference tree that mjhip accepts (nv <= 320; tools/model_sweep.py
--fixtures, run in the build container): <stem>.mjb.gz = mj_saveModel of the compiled model, <stem>.npz = the initial
state (first keyframe or reset) and the compiled reference's trajectory -- FULLPHYSICS state, (ncon, nefc) and sensordata
after each of 15 mj_step calls -- as shipped and for the sweep's variations (PGS, Newton + elliptic cones, RK4,
implicitfast), each from two builds of the reference: as built (glibc libm; what the host emulation of the kernels
reproduces) and linked with the kernels' own sin / cos / atan2 / exp (`*_dm`; what the device reproduces -- a last-bit
difference in a hinge's sine moves the iteration an unconverged PGS solve stops at: robot_arm.xml under PGS differs by
4e-6 between the two builds after 15 steps).  Here each model is loaded by the PRODUCT's .mjb reader (mjh_mjb.h), stepped by libmjhip.so on cuda:0 and
compared the way the sweep compares (`engine_forward_test.cc`-style, step by step): state and sensordata within 1e-6
relative, contact and constraint counts exact, no warning the reference does not raise.
The full sweep against the live oracle is `tools/model_sweep.py --from-mjb tests/golden/sweep --device`
(profiles/r06_sweep_gpu/sweep.txt)."""
import gzip
import os

import numpy as np
import pytest

from conftest import GOLDEN

SWEEP = os.path.join(GOLDEN, "sweep")
OPTION_CODES = {"solver", "cone", "integrator"}


def _index():
    p = os.path.join(SWEEP, "index.txt")
    if not os.path.exists(p):
        return []
    return [ln.split() for ln in open(p) if ln.strip() and not ln.startswith("#")]


INDEX = _index()


def _rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    if a.size == 0:
        return 0.0
    if not (np.all(np.isfinite(a)) and np.all(np.isfinite(b))):
        return float("inf") if not np.array_equal(np.isfinite(a), np.isfinite(b)) else 0.0
    return float(np.max(np.abs(a - b) / (1.0 + np.abs(b))))


def replay(lib, stem, tmp_path, lds=0, dm=False):
    """every variation of one fixture; returns [(variation, state err, sensor err, counts exact, warnings)].
    dm: compare with the trajectory of the reference linked with the kernels' own sin / cos / atan2 / exp (what the DEVICE
    evaluates; the emulation calls the host's libm like the reference as built)"""
    import mujoco_amd as ma
    fx = np.load(os.path.join(SWEEP, stem + ".npz"))
    mjb = os.path.join(str(tmp_path), stem + ".mjb")
    with open(mjb, "wb") as out:
        out.write(gzip.open(os.path.join(SWEEP, stem + ".mjb.gz"), "rb").read())
    variations = [k[:-len(":ref_state")] for k in fx.files if k.endswith(":ref_state")]
    s0 = fx["state0"]
    results = []
    for var in variations:
        model = ma.MjbModel(lib, mjb)
        for ch in fx[var + ":changes"]:
            k, v = str(ch).split("=")
            assert k in OPTION_CODES
            model.set_option(k, int(v))
        dm = ma.DeviceModel(lib, model)
        nq, nv, na = dm.nq, dm.nv, dm.na
        b = ma.Batch(dm, 1)
        if lds:
            b.plan_lds(lds)
        b.reset()
        b.set("time", s0[None, :1]); b.set("qpos", s0[None, 1:1 + nq]); b.set("qvel", s0[None, 1 + nq:1 + nq + nv])
        if na > 0:
            b.set("act", s0[None, 1 + nq + nv:1 + nq + nv + na])
        if fx["mocap_pos"].size:
            b.set("mocap_pos", fx["mocap_pos"][None]); b.set("mocap_quat", fx["mocap_quat"][None])
        if fx["ctrl0"].size:
            b.set("ctrl", fx["ctrl0"][None])
        # (an array of the device-libm build is stored only where it differs from the reference as built)
        pick = lambda k: fx[var + ":" + k + "_dm"] if dm and (var + ":" + k + "_dm") in fx.files else fx[var + ":" + k]
        ref, ref_int, ref_sens = pick("ref_state"), pick("ref_counts"), pick("ref_sensordata")
        worst = worst_s = 0.0
        exact = True
        for t in range(ref.shape[0]):
            b.step(1)
            got = np.concatenate([b.get("time")[0, :1], b.get("qpos")[0], b.get("qvel")[0]] + ([b.get("act")[0]] if na > 0 else []))
            worst = max(worst, _rel(got, ref[t, :got.size]))
            c = b.get("counts")[0]
            exact = exact and (int(c[0]), int(c[1])) == (int(ref_int[t, 0]), int(ref_int[t, 1]))
            if ref_sens.shape[1]:
                worst_s = max(worst_s, _rel(b.get("sensordata")[0], ref_sens[t]))
        warn = int(b.get("warning")[0].sum())
        results.append((var, worst, worst_s, exact, warn, int(fx[var + ":ref_warnings"])))
        b.close()
    return results


def _assert_ok(stem, short, results):
    assert results, f"{short}: no variation in the fixture"
    for var, err, err_s, exact, warn, ref_warn in results:
        if ref_warn > 0:
            # the reference itself gave up on this variation (planks.xml forced to PGS: 1360 rows, its arena cannot hold
            # efc_AR -- mjWARN_CNSTRFULL); what follows a warning is not a trajectory to reproduce: mjhip has to warn too
            assert warn > 0, f"{short} [{var}]: the reference raises a warning, mjhip does not"
            continue
        assert err <= 1e-6, f"{short} [{var}]: state deviates by {err:.2e}"
        assert err_s <= 1e-6, f"{short} [{var}]: sensordata deviates by {err_s:.2e}"
        assert exact, f"{short} [{var}]: contact / constraint counts differ"
        assert warn == 0 or ref_warn > 0, f"{short} [{var}]: mjhip raised a warning the reference does not"


@pytest.mark.gpu
@pytest.mark.parametrize("stem,short", INDEX, ids=[s for s, _ in INDEX])
def test_reference_model_replays_on_gpu(hip_lib, tmp_path, stem, short):
    _assert_ok(stem, short, replay(hip_lib, stem, tmp_path, dm=True))


def test_fixture_set_is_complete():
    """the index lists >= 120 models incl. every accepted model/flex/*.xml, and every entry has both files"""
    assert len(INDEX) >= 120
    flex = [s for _, s in INDEX if s.startswith("model/flex/")]
    assert len(flex) >= 20, flex
    for stem, _ in INDEX:
        assert os.path.exists(os.path.join(SWEEP, stem + ".mjb.gz")) and os.path.exists(os.path.join(SWEEP, stem + ".npz")), stem


@pytest.mark.parametrize("stem", ["model__car__car", "model__flex__pinch", "model__tendon_arm__arm26"])
def test_fixture_replays_on_the_emulation(hostsim_lib, tmp_path, stem):
    """the replay harness itself, on the host emulation of the kernels (three small fixtures)"""
    short = dict(INDEX).get(stem)
    if short is None:
        pytest.skip(stem + " is not in the fixture set")
    _assert_ok(stem, short, replay(hostsim_lib, stem, tmp_path, lds=40960))
