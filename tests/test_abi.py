"""CPU: libmjhip.so loads, exports exactly what include/mjhip.h declares, its host-only entry points
work, and every compute entry point fails loudly without a GPU (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

import mujoco_amd
from mujoco_amd import _capi
from conftest import GOLDEN, ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mjhip.h")).read()
    return sorted(set(re.findall(r"MJHIP_API[^;]*?\b(mjhip_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = mujoco_amd.lib()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib.c, n), f"libmjhip.so does not export {n}"
    assert sorted(_capi.Lib.SYMBOLS) == names


def test_backend_is_hip():
    assert mujoco_amd.lib().backend() == "hip-gfx950"


def test_mjb_reader_matches_reference_loader(rb):
    """product-side .mjb reader vs the reference's mj_loadModel on the same file"""
    lib = mujoco_amd.lib()
    path = os.path.join(GOLDEN, "humanoid.mjb")
    mine = mujoco_amd.MjbModel(lib, path)
    ref = rb.MjModel.from_binary_path(path)
    view = rb.MjModel(mine._address, ref._lib, owned=False)   # read my struct through the oracle's field tables
    for f in ["nq", "nv", "nu", "nbody", "ngeom", "njnt", "nC", "ntendon", "nbuffer"]:
        assert getattr(view, f) == getattr(ref, f), f
    for f in ["body_pos", "body_quat", "body_inertia", "jnt_axis", "geom_size", "dof_damping", "M_colind",
              "geom_type", "actuator_gear", "qpos0", "tendon_range", "body_invweight0", "geom_friction"]:
        assert np.array_equal(getattr(view, f), getattr(ref, f)), f
    assert view.opt.timestep == ref.opt.timestep and view.opt.solver == ref.opt.solver
    assert view.stat.meaninertia == ref.stat.meaninertia
    mine.set_option("solver", 0)
    assert view.opt.solver == 0


def test_mjb_reader_rejects_garbage(tmp_path):
    lib = mujoco_amd.lib()
    p = tmp_path / "bad.mjb"
    p.write_bytes(b"not a model")
    with pytest.raises(mujoco_amd.MjhipError):
        mujoco_amd.MjbModel(lib, str(p))
    with pytest.raises(mujoco_amd.MjhipError):
        mujoco_amd.MjbModel(lib, str(tmp_path / "missing.mjb"))


def test_compute_fails_loudly_without_gpu():
    lib = mujoco_amd.lib()
    if lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    m = mujoco_amd.MjbModel(lib, os.path.join(GOLDEN, "humanoid.mjb"))
    m.set_option("solver", 0)
    with pytest.raises(mujoco_amd.MjhipError, match="no HIP device"):
        mujoco_amd.DeviceModel(lib, m)


def test_rollout_module_fails_loudly_without_gpu(rb):
    from mujoco_amd import rollout
    lib = mujoco_amd.lib()
    if lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    m = rb.MjModel.from_binary_path(os.path.join(GOLDEN, "humanoid.mjb"))
    m.opt.solver = 0
    d = rb.MjData(m)
    s0 = rb.mj_getState(m, d, rb.mjSTATE_FULLPHYSICS)
    with pytest.raises(RuntimeError, match="no HIP device"):
        rollout.rollout(m, d, s0, nstep=2)


def test_mjb_reader_rejects_corrupted_sizes(tmp_path):
    """size fields of a .mjb are validated before anything is sized from them (a negative or huge
    count must not reach an allocation or a read)"""
    import struct
    from conftest import GOLDEN, HOSTSIM_LIB
    from mujoco_amd import _capi as K
    if not os.path.exists(HOSTSIM_LIB):
        pytest.skip("hostsim library not built")
    lib = K.Lib(HOSTSIM_LIB)
    raw = bytearray(open(os.path.join(GOLDEN, "humanoid.mjb"), "rb").read())
    for value in (-5, 2**31 - 1):
        bad = bytearray(raw)
        struct.pack_into("<i", bad, 5*4 + 3*4, value)          # an early int size field (after the 5-int header)
        p = tmp_path / f"bad_{value}.mjb"
        p.write_bytes(bytes(bad))
        with pytest.raises(K.MjhipError, match="corrupted|does not match|mismatch"):
            K.MjbModel(lib, str(p))
    K.MjbModel(lib, os.path.join(GOLDEN, "humanoid.mjb"))       # the intact file still loads
