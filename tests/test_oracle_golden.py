"""CPU: the oracle (compiled reference, oracle/_ref) reproduces the committed golden fixtures, and
satisfies the relational pins the reference's own tests put on this path."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, humanoid_pgs_oracle
from parity_utils import oracle_rollout


def test_oracle_reproduces_golden_humanoid(rb, golden):
    fx = golden("humanoid")
    m = humanoid_pgs_oracle(rb)
    n = 3   # a few envs keep the CPU suite short; every env is covered by the gpu/hostsim tests
    out, ints = oracle_rollout(rb, m, fx["state0"][:n], fx["ctrl"][:n])
    assert np.array_equal(out, fx["state"][:n])
    assert np.array_equal(ints[..., 0], fx["ncon"][:n])
    assert np.array_equal(ints[..., 1], fx["nefc"][:n])
    assert np.array_equal(ints[..., 2], fx["niter"][:n])


def test_oracle_reproduces_golden_slider_crank(rb, golden):
    fx = golden("slider_crank")
    m = rb.MjModel.from_binary_path(os.path.join(GOLDEN, "slider_crank.mjb"))
    m.opt.solver = 0
    out, ints = oracle_rollout(rb, m, fx["state0"][:4], fx["ctrl"][:4])
    assert np.array_equal(out, fx["state"][:4])
    assert ints[..., 0].max() == 0 and ints[..., 1].max() == 0   # BASELINE config 1: contact-free


def test_solvers_equivalent_on_humanoid(rb):
    """reference pin: SolverTest.SolversEquivalent (test/engine/engine_solver_test.cc:226): PGS and
    Newton agree on qfrc_constraint when run to convergence with warmstart off."""
    m = rb.MjModel.from_binary_path(os.path.join(GOLDEN, "humanoid.mjb"))
    d = rb.MjData(m)
    rb.mj_resetDataKeyframe(m, d, 2)   # prone: many contacts
    m.opt.tolerance = 0
    m.opt.iterations = 500
    m.opt.disableflags = int(m.opt.disableflags) | (1 << 9)
    res = {}
    for solver in (2, 0):
        m.opt.solver = solver
        rb.mj_forward(m, d)
        res[solver] = np.array(d.qfrc_constraint)
    assert d.nefc > 0
    scale = np.abs(res[2]).max()
    assert np.abs(res[0] - res[2]).max() <= 1e-6 * scale


def test_rollout_equals_serial_step_property(rb, golden):
    """reference pin: rollout_test.py compares rollout against a python mj_step loop bit-exactly;
    the oracle loop is deterministic and stateless w.r.t. the scratch mjData."""
    fx = golden("humanoid")
    m = humanoid_pgs_oracle(rb)
    a, _ = oracle_rollout(rb, m, fx["state0"][:1], fx["ctrl"][:1, :30])
    b, _ = oracle_rollout(rb, m, fx["state0"][:1], fx["ctrl"][:1, :30])
    assert np.array_equal(a, b)
