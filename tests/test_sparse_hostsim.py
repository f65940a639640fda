"""CPU: the sparse constraint path (mujoco_amd/csrc/mjh_sparse.h + the SPA instantiation of mjh_newton.h) on the
host wavefront emulation against the compiled reference, BIT FOR BIT.

With jacobian = sparse (or auto and nv >= 60) the reference stores efc_J in compressed rows and its primal
solvers run mju_mulMatVecSparse / mju_sqrMatTDSparse / mju_cholFactorNumeric / mju_cholUpdateSparse /
mju_cholSolveSparse (engine_util_sparse.c, engine_util_solve.c:145-500); its dual solver (PGS) sweeps compressed rows of
efc_AR (mju_dotSparse).  Every scene below forces that path
(opt.jacobian = mjJAC_SPARSE) and demands identical state trajectories and identical Newton / CG iteration
counts at every step.  The GPU counterparts are in tests/test_gpu_parity.py.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from mujoco_amd import _capi as K
from parity_utils import (CONDIM_XML, EQ_XML, ISLANDS_XML, TENDON_XML, BOXBOX_XML, chain_xml, oracle_rollout)

mjJAC_SPARSE = 1
SCENES = {
    # 67-dof serial chain (nv >= 60: sparse under jacobian=auto as well): long row patterns, ball / slide / hinge limits
    "chain": (lambda: chain_xml(62).replace('jacobian="dense"', 'jacobian="auto"'), 25),
    # connect / weld / joint / tendon equalities, site anchors, a static-static weld (empty chain: dropped)
    "equality": (lambda: EQ_XML, 12),
    # four kinematic trees: several islands, some trees unconstrained at times
    "islands": (lambda: ISLANDS_XML, 30),
    # fixed and spatial tendons with limits and friction loss: tendon row patterns
    "tendon": (lambda: TENDON_XML, 30),
    # condim 1 / 3 / 4 / 6 contacts
    "condim": (lambda: CONDIM_XML, 25),
    # stacked boxes: many contacts per body pair, rows with identical patterns
    "boxbox": (lambda: BOXBOX_XML, 20),
}


def _run(rb, lib, xml_text, tmp_path, solver, cone, T, seed=0, exact=True, kind=None):
    xml = tmp_path / "s.xml"
    xml.write_text(xml_text)
    m = rb.MjModel.from_xml_path(str(xml), kind=kind)
    m.opt.solver = solver
    m.opt.cone = cone
    m.opt.jacobian = mjJAC_SPARSE
    dm = K.DeviceModel(lib, m)
    assert dm.size("sparse") == 1
    d = rb.MjData(m)
    rb.mj_resetData(m, d)
    d.qvel[:] = np.random.default_rng(seed + 1).normal(0, .3, m.nv)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)[None].copy()
    ctrl = np.random.default_rng(seed).uniform(-1, 1, (1, T, m.nu))
    # reference trajectory with per-step integer observables
    ref = np.zeros((T, s0.shape[1]))
    ints = np.zeros((T, 4), np.int64)
    rb.mj_setState(m, d, s0[0], rb.mjSTATE_FULLPHYSICS)
    d.qacc_warmstart[:] = 0
    pre_s, pre_w = [], []
    for t in range(T):
        pre_s.append(rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)); pre_w.append(np.array(d.qacc_warmstart))
        d.ctrl[:] = ctrl[0, t]
        rb.mj_step(m, d)
        ref[t] = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
        ints[t] = (d.ncon, d.nefc, d.solver_niter[0], d.nisland)
    b = K.Batch(dm, 1)
    out = b.rollout_host(T, K.mjSTATE_CTRL, s0, None, ctrl)
    assert b.get("warning").sum() == 0
    rel = lambda a, r: float(np.max(np.abs(a - r)/np.maximum(1.0, np.abs(r))))
    if exact:
        assert np.array_equal(out[0], ref), (np.abs(out[0] - ref).max(), int(np.argmax(np.any(out[0] != ref, axis=1))))
    else:
        # (device libm: atan2 / pow / exp differ from glibc in the last bit, which a rollout then amplifies)
        assert rel(out[0], ref) <= 1e-6
    # every step again from identical inputs: iteration count of island 0, contact / row / island counts
    bb = K.Batch(dm, T)
    one = bb.rollout_host(1, K.mjSTATE_CTRL, np.array(pre_s), np.array(pre_w), ctrl[0][:, None])
    if exact:
        assert np.array_equal(one[:, 0], ref)
    else:
        assert rel(one[:, 0], ref) <= 1e-9
    c = bb.get("counts")
    assert np.array_equal(c[:, 0], ints[:, 0]) and np.array_equal(c[:, 1], ints[:, 1])
    if exact:
        assert np.array_equal(c[:, 5], ints[:, 2]), (c[:, 5], ints[:, 2])
    else:
        assert np.mean(c[:, 5] == ints[:, 2]) >= 0.9, (c[:, 5], ints[:, 2])      # (a last-bit libm difference can move a non-converged solve's stopping iteration)
    assert np.array_equal(c[:, 6], ints[:, 3])
    return ints


# (solver, cone) per scene -- the host emulation switches 64 fibers at every cross-lane exchange and the sparse routines
# make thousands of them per solve, so the CPU matrix is thinned (the GPU test runs every scene with three combinations)
NEWTON, CG, PGS = 2, 1, 0
CASES = [("condim", PGS, 0), ("condim", PGS, 1), ("boxbox", PGS, 0), ("islands", PGS, 0), ("equality", PGS, 1),
         ("chain", NEWTON, 0), ("chain", NEWTON, 1), ("chain", CG, 0), ("chain", CG, 1),
         ("condim", NEWTON, 0), ("condim", NEWTON, 1), ("condim", CG, 0), ("condim", CG, 1),
         ("equality", NEWTON, 0), ("equality", NEWTON, 1), ("islands", NEWTON, 0), ("islands", CG, 0),
         ("tendon", NEWTON, 0), ("tendon", CG, 1), ("boxbox", NEWTON, 1), ("boxbox", CG, 0)]


@pytest.mark.parametrize("scene,solver,cone", CASES, ids=lambda v: v if isinstance(v, str) else str(v))
def test_sparse_primal_solvers_bit_exact(rb, hostsim_lib, tmp_path, scene, solver, cone):
    make, T = SCENES[scene]
    ints = _run(rb, hostsim_lib, make(), tmp_path, solver, cone, T)
    assert ints[:, 1].max() > 0 and ints[:, 2].max() > 0          # constraints were active and the solver iterated


def test_cube_3x3x3_bit_exact(hostsim_lib):
    """BASELINE config 4 as shipped (nv = 66 >= 60: sparse, Newton, implicitfast, islands): every committed
    reference step reproduced bit for bit from (state, warm start, ctrl), with contact / row / iteration counts"""
    fx = np.load(os.path.join(GOLDEN, "cube_3x3x3_steps.npz"))
    mm = K.MjbModel(hostsim_lib, os.path.join(GOLDEN, "cube_3x3x3.mjb"))
    dm = K.DeviceModel(hostsim_lib, mm)
    assert dm.size("sparse") == 1
    idx = np.arange(0, len(fx["state"]), 8)
    b = K.Batch(dm, len(idx))
    out = b.rollout_host(1, K.mjSTATE_CTRL, fx["state"][idx], fx["warmstart"][idx], fx["ctrl"][idx][:, None])
    assert b.get("warning").sum() == 0
    c = b.get("counts")
    assert np.array_equal(c[:, 0], fx["ints"][idx, 0]) and np.array_equal(c[:, 1], fx["ints"][idx, 1])
    assert np.array_equal(c[:, 5], fx["ints"][idx, 2])
    assert np.array_equal(out[:, 0], fx["next"][idx])
