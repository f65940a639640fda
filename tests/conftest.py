"""Shared fixtures.  Markers: `gpu` = needs a real MI355X (run with -m gpu on the GPU box)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
HOSTSIM_LIB = os.path.join(ROOT, "tests", "hostsim", "libmjhip_hostsim.so")
REF = os.environ.get("MUJOCO_REF", "/root/reference")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X GPU (libmjhip.so HIP path)")


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """the CPU selection (-m "not gpu": host emulation + compiled reference, ~15 core-minutes) spreads over four xdist
    workers unless -n was given; $MJHIP_TEST_WORKERS=0 keeps it serial.  GPU selections stay in one process."""
    if not config.pluginmanager.hasplugin("xdist") or os.environ.get("PYTEST_XDIST_WORKER"):
        return None
    if getattr(config.option, "numprocesses", None) or "not gpu" not in (getattr(config.option, "markexpr", "") or ""):
        return None
    n = int(os.environ.get("MJHIP_TEST_WORKERS", min(4, os.cpu_count() or 1)))
    if n > 1:
        config.option.numprocesses = n
    return None


def _have_reference():
    return os.path.isdir(os.path.join(REF, "src", "engine"))


@pytest.fixture(scope="session")
def rb():
    """the compiled-reference oracle binding (test infrastructure)"""
    from oracle import refbind
    if not refbind.available():
        if _have_reference():
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "-j8", "REF=" + REF], check=True)
        else:
            pytest.skip("compiled oracle (oracle/_ref) not available on this box")
    return refbind


@pytest.fixture(scope="session")
def hostsim_lib():
    """C ABI backed by the host wavefront emulation (test infrastructure, CPU only)"""
    from mujoco_amd import _capi
    if not os.path.exists(HOSTSIM_LIB):
        inc = os.path.join(REF, "include")
        if not os.path.isdir(inc):
            pytest.skip("hostsim library not built and MuJoCo headers unavailable")
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "hostsim"), "MUJOCO_INCLUDE=" + inc], check=True)
    return _capi.Lib(HOSTSIM_LIB)


@pytest.fixture(scope="session")
def hip_lib():
    """the product library on a real GPU"""
    import mujoco_amd
    lib = mujoco_amd.lib()
    if lib.device_count() <= 0:
        pytest.fail("gpu test selected but no HIP device is visible")
    assert lib.backend() == "hip-gfx950"
    return lib


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + "_traj.npz"))
    return load


def humanoid_pgs_oracle(rb):
    m = rb.MjModel.from_binary_path(os.path.join(GOLDEN, "humanoid.mjb"))
    m.opt.solver = 0
    return m


def contact_rich_states(rb, m, nenv, seed=0, settle=(40, 160)):
    """states reached by the oracle under random actions from keyframes: contacts, limits, motion."""
    d = rb.MjData(m)
    rng = np.random.default_rng(seed)
    out = []
    for e in range(nenv):
        if m.nkey:
            rb.mj_resetDataKeyframe(m, d, e % m.nkey)
        else:
            rb.mj_resetData(m, d)
        d.qvel[:] = rng.normal(0, 0.5, size=m.nv)
        for _ in range(int(rng.integers(*settle))):
            d.ctrl[:] = rng.uniform(-1, 1, size=m.nu)
            rb.mj_step(m, d)
        out.append(dict(qpos=np.array(d.qpos), qvel=np.array(d.qvel),
                        qacc_warmstart=np.array(d.qacc_warmstart),
                        ctrl=rng.uniform(-1, 1, size=m.nu), time=d.time))
    return out


def many_constraint_states(rb, m, want, seed=5, lo=64, hi=128):
    """humanoid states whose next mj_step has lo < nefc <= hi: dropped lying with bent limbs (the
    two-constraints-per-lane PGS path)"""
    d = rb.MjData(m)
    rng = np.random.default_rng(seed)
    states, nefcs = [], []
    for _ in range(400):
        rb.mj_resetData(m, d)
        d.qpos[2] = 0.12 + 0.2*rng.uniform()
        ax = rng.normal(size=3); ax[2] *= 0.2; ax /= np.linalg.norm(ax)
        ang = np.pi/2 + rng.normal(0, .2)
        d.qpos[3:7] = [np.cos(ang/2), *(np.sin(ang/2)*ax)]
        d.qpos[7:] += rng.normal(0, .5, m.nq - 7)
        took = 0
        for _t in range(300):
            d.ctrl[:] = 0.3*rng.uniform(-1, 1, m.nu)
            pre = dict(qpos=np.array(d.qpos), qvel=np.array(d.qvel), qacc_warmstart=np.array(d.qacc_warmstart),
                       ctrl=np.array(d.ctrl), time=d.time)
            rb.mj_step(m, d)
            if lo < d.nefc <= hi and took < 2 and rng.uniform() < 0.3:
                states.append(pre); nefcs.append(int(d.nefc)); took += 1
        if len(states) >= want:
            break
    return states[:want], nefcs[:want]
