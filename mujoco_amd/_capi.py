"""ctypes binding of the mjhip C ABI (include/mjhip.h).

`Lib(path)` wraps one shared library exporting that ABI.  The product loads
`mujoco_amd/csrc/libmjhip.so` (HIP, gfx950) through `mujoco_amd.lib()`; the test suite may point
the same wrapper at the host wavefront emulation built under tests/hostsim.  Nothing in here
computes physics: it is marshalling only.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

# stage bits of mjhip_batch_forward (mujoco_amd/csrc/mjh_step.h)
STAGE_KINEMATICS = 1 << 0
STAGE_INERTIA = 1 << 1
STAGE_COLLISION = 1 << 2
STAGE_MAKE = 1 << 3
STAGE_PROJECT = 1 << 4
STAGE_TRANSMISSION = 1 << 5
STAGE_VELOCITY = 1 << 6
STAGE_ACTUATION = 1 << 7
STAGE_CONSTRAINT = 1 << 8
STAGE_FINISH = 1 << 10
STAGE_REFERENCE = 1 << 11
STAGE_ALL = ((1 << 9) - 1) | STAGE_FINISH | STAGE_REFERENCE
STAGE_EULER = 1 << 9
STAGE_LDS = 1 << 21      # run forward() on the LDS residency plan with per-stage write-back (debug)

# mjtState bits (include/mujoco/mjtype.h:504-527)
mjSTATE_TIME = 1 << 0
mjSTATE_QPOS = 1 << 1
mjSTATE_QVEL = 1 << 2
mjSTATE_ACT = 1 << 3
mjSTATE_HISTORY = 1 << 4
mjSTATE_WARMSTART = 1 << 5
mjSTATE_CTRL = 1 << 6
mjSTATE_QFRC_APPLIED = 1 << 7
mjSTATE_XFRC_APPLIED = 1 << 8
mjSTATE_EQ_ACTIVE = 1 << 9
mjSTATE_MOCAP_POS = 1 << 10
mjSTATE_MOCAP_QUAT = 1 << 11
mjSTATE_USERDATA = 1 << 12
mjSTATE_PLUGIN = 1 << 13
mjSTATE_PHYSICS = mjSTATE_QPOS | mjSTATE_QVEL | mjSTATE_ACT | mjSTATE_HISTORY
mjSTATE_FULLPHYSICS = mjSTATE_TIME | mjSTATE_PHYSICS | mjSTATE_PLUGIN


class MjhipError(RuntimeError):
    pass


class Lib:
    def __init__(self, path: str):
        if not os.path.exists(path):
            raise MjhipError(
                f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). mujoco_amd has no CPU fallback.")
        self.path = path
        lib = C.CDLL(path)
        vp, ci, cu = C.c_void_p, C.c_int, C.c_uint
        dp = C.c_void_p
        lib.mjhip_backend.restype = C.c_char_p
        lib.mjhip_last_error.restype = C.c_char_p
        lib.mjhip_device_count.restype = ci
        lib.mjhip_model_create.restype = vp
        lib.mjhip_model_create.argtypes = [vp, ci, ci]
        lib.mjhip_model_destroy.argtypes = [vp]
        lib.mjhip_model_size.restype = ci
        lib.mjhip_model_size.argtypes = [vp, C.c_char_p]
        lib.mjhip_load_mjb.restype = vp
        lib.mjhip_load_mjb.argtypes = [C.c_char_p]
        lib.mjhip_free_mjb.argtypes = [vp]
        lib.mjhip_set_option.restype = ci
        lib.mjhip_set_option.argtypes = [vp, C.c_char_p, C.c_double]
        lib.mjhip_batch_create.restype = vp
        lib.mjhip_batch_create.argtypes = [vp, ci, ci]
        lib.mjhip_batch_create_layout.restype = vp
        lib.mjhip_batch_create_layout.argtypes = [vp, ci, ci, ci]
        lib.mjhip_batch_destroy.argtypes = [vp]
        lib.mjhip_batch_nenv.restype = ci
        lib.mjhip_batch_nenv.argtypes = [vp]
        lib.mjhip_batch_reset.restype = ci
        lib.mjhip_batch_reset.argtypes = [vp]
        lib.mjhip_batch_field.restype = ci
        lib.mjhip_batch_field.argtypes = [vp, C.c_char_p, C.POINTER(vp), C.POINTER(ci), C.POINTER(ci)]
        lib.mjhip_batch_get.restype = ci
        lib.mjhip_batch_get.argtypes = [vp, C.c_char_p, vp]
        lib.mjhip_batch_set.restype = ci
        lib.mjhip_batch_set.argtypes = [vp, C.c_char_p, vp]
        lib.mjhip_batch_set_variant.restype = ci
        lib.mjhip_batch_set_variant.argtypes = [vp, C.c_char_p]
        lib.mjhip_batch_set_mfma.restype = ci
        lib.mjhip_batch_set_mfma.argtypes = [vp, ci]
        lib.mjhip_batch_set_pgs_mode.restype = ci
        lib.mjhip_batch_set_pgs_mode.argtypes = [vp, ci]
        lib.mjhip_batch_variant.restype = C.c_char_p
        lib.mjhip_batch_variant.argtypes = [vp]
        lib.mjhip_batch_kernel.restype = C.c_char_p
        lib.mjhip_batch_kernel.argtypes = [vp]
        lib.mjhip_batch_plan_lds.restype = ci
        lib.mjhip_batch_plan_lds.argtypes = [vp, ci]
        lib.mjhip_batch_lds_report.restype = C.c_char_p
        lib.mjhip_batch_lds_report.argtypes = [vp]
        lib.mjhip_batch_forward.restype = ci
        lib.mjhip_batch_forward.argtypes = [vp, ci, vp]
        lib.mjhip_batch_step.restype = ci
        lib.mjhip_batch_step.argtypes = [vp, ci, vp]
        lib.mjhip_batch_step1.restype = ci
        lib.mjhip_batch_step1.argtypes = [vp, vp]
        lib.mjhip_batch_step2.restype = ci
        lib.mjhip_batch_step2.argtypes = [vp, vp]
        lib.mjhip_batch_rollout.restype = ci
        lib.mjhip_batch_rollout.argtypes = [vp, ci, cu, dp, dp, dp, dp, ci, vp]
        lib.mjhip_batch_rollout_sensors.restype = ci
        lib.mjhip_batch_rollout_sensors.argtypes = [vp, ci, cu, dp, dp, dp, dp, dp, ci, vp]
        lib.mjhip_batch_sync.restype = ci
        lib.mjhip_batch_sync.argtypes = [vp, vp]
        lib.mjhip_batch_trouble.restype = ci
        lib.mjhip_batch_trouble.argtypes = [vp, ci, C.POINTER(ci), C.POINTER(ci)]
        lib.mjhip_rollout_clear_cache.restype = None
        lib.mjhip_rollout.restype = ci
        lib.mjhip_rollout.argtypes = [vp, vp, ci, ci, cu, dp, dp, dp, dp, dp]
        self.c = lib

    # every symbol include/mjhip.h declares (checked by the CPU test-suite)
    SYMBOLS = (
        "mjhip_backend", "mjhip_last_error", "mjhip_device_count", "mjhip_model_create",
        "mjhip_model_destroy", "mjhip_model_size", "mjhip_load_mjb", "mjhip_free_mjb", "mjhip_set_option",
        "mjhip_batch_create", "mjhip_batch_create_layout", "mjhip_batch_destroy", "mjhip_batch_nenv", "mjhip_batch_reset",
        "mjhip_batch_field", "mjhip_batch_get", "mjhip_batch_set", "mjhip_batch_forward",
        "mjhip_batch_plan_lds", "mjhip_batch_lds_report", "mjhip_batch_set_variant", "mjhip_batch_variant", "mjhip_batch_kernel",
        "mjhip_batch_step", "mjhip_batch_rollout", "mjhip_batch_rollout_sensors", "mjhip_batch_sync", "mjhip_rollout",
        "mjhip_batch_trouble", "mjhip_rollout_clear_cache", "mjhip_batch_step1", "mjhip_batch_step2", "mjhip_batch_set_mfma", "mjhip_batch_set_pgs_mode",
    )

    def backend(self) -> str:
        return self.c.mjhip_backend().decode()

    def error(self) -> str:
        return self.c.mjhip_last_error().decode(errors="replace")

    def device_count(self) -> int:
        return int(self.c.mjhip_device_count())

    def check(self, rc: int, what: str) -> None:
        if rc != 0:
            raise MjhipError(f"{what} failed ({rc}): {self.error()}")


def _address(obj) -> int:
    """raw mjModel*/mjData* of a MuJoCo binding object (the official bindings and the test
    oracle binding both expose `_address`) or a plain integer address."""
    if isinstance(obj, int):
        return obj
    a = getattr(obj, "_address", None)
    if a is None:
        raise TypeError(f"{type(obj).__name__} has no _address: expected a MuJoCo MjModel/MjData")
    return int(a)


class MjbModel:
    """An mjModel loaded from a .mjb file by the library's own reader (no MuJoCo needed)."""

    def __init__(self, lib: Lib, path: str):
        self._lib = lib
        p = lib.c.mjhip_load_mjb(os.fsencode(path))
        if not p:
            raise MjhipError(lib.error())
        self._address = int(p)

    def set_option(self, name: str, value: float) -> None:
        self._lib.check(self._lib.c.mjhip_set_option(self._address, name.encode(), float(value)), "set_option")

    def __del__(self):
        try:
            if self._address:
                self._lib.c.mjhip_free_mjb(self._address)
                self._address = 0
        except Exception:
            pass


class DeviceModel:
    """Device copy of the model constants (mjhip_model_create)."""

    def __init__(self, lib: Lib, model, nconmax: int = 0, nefcmax: int = 0):
        self._lib = lib
        self._src = model  # keep the mjModel alive
        h = lib.c.mjhip_model_create(_address(model), int(nconmax), int(nefcmax))
        if not h:
            raise MjhipError(lib.error())
        self._h = h

    def size(self, name: str) -> int:
        v = self._lib.c.mjhip_model_size(self._h, name.encode())
        if v < 0:
            raise MjhipError(self._lib.error())
        return int(v)

    def __getattr__(self, name):
        if name.startswith("n") and not name.startswith("_"):
            return self.size(name)
        raise AttributeError(name)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.c.mjhip_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batch:
    """nenv device-resident environments of one model (mjhip_batch_create)."""

    def __init__(self, model: DeviceModel, nenv: int, device: int = 0, layout: Optional[str] = None):
        """layout: "aos" (one wavefront per env, single kernel), "soa" (lane-per-env pipeline) or
        None (library default / $MJHIP_LAYOUT)"""
        self._lib = model._lib
        self.model = model
        if layout is None:
            h = self._lib.c.mjhip_batch_create(model._h, int(nenv), int(device))
        else:
            h = self._lib.c.mjhip_batch_create_layout(model._h, int(nenv), int(device),
                                                     {"aos": 0, "soa": 1}[layout])
        if not h:
            raise MjhipError(self._lib.error())
        self._h = h
        self.nenv = int(nenv)
        self.device = int(device)

    def field_info(self, name: str):
        p, n, isint = C.c_void_p(), C.c_int(), C.c_int()
        rc = self._lib.c.mjhip_batch_field(self._h, name.encode(), C.byref(p), C.byref(n), C.byref(isint))
        self._lib.check(rc, f"field {name}")
        return int(p.value or 0), int(n.value), bool(isint.value)

    def get(self, name: str) -> np.ndarray:
        _, n, isint = self.field_info(name)
        out = np.zeros((self.nenv, n), dtype=np.int32 if isint else np.float64)
        if n:
            self._lib.check(self._lib.c.mjhip_batch_get(self._h, name.encode(), out.ctypes.data), f"get {name}")
        return out

    def set(self, name: str, value) -> None:
        _, n, isint = self.field_info(name)
        a = np.ascontiguousarray(value, dtype=np.int32 if isint else np.float64)
        if a.shape != (self.nenv, n):
            a = np.ascontiguousarray(np.broadcast_to(a.reshape((-1, n)), (self.nenv, n)))
        if n:
            self._lib.check(self._lib.c.mjhip_batch_set(self._h, name.encode(), a.ctypes.data), f"set {name}")

    def trouble(self):
        """(capacity overflows, unsupported-collider hits) summed over the batch's warning counters:
        non-zero means some environment was frozen where the reference would have kept simulating"""
        a, b = C.c_int(), C.c_int()
        self._lib.check(self._lib.c.mjhip_batch_trouble(self._h, 0, C.byref(a), C.byref(b)), "trouble")
        return int(a.value), int(b.value)

    def reset(self) -> None:
        self._lib.check(self._lib.c.mjhip_batch_reset(self._h), "reset")

    def forward(self, stages: int = -1, stream: int = 0, lds: bool = False) -> None:
        if stages < 0:
            stages = STAGE_ALL
        if lds:
            stages |= STAGE_LDS
        self._lib.check(self._lib.c.mjhip_batch_forward(self._h, int(stages), stream or None), "forward")
        self.sync(stream)

    def plan_lds(self, lds_bytes: int) -> int:
        """(re)plan LDS residency with `lds_bytes` per one-wavefront workgroup; 0 = all global.
        Returns the bytes left for the per-step constraint arrays."""
        rc = self._lib.c.mjhip_batch_plan_lds(self._h, int(lds_bytes))
        self._lib.check(min(rc, 0), "plan_lds")
        return rc

    def kernel_variant(self) -> str:
        """name of the kernel mapping that steps this batch: generic | lean"""
        return self._lib.c.mjhip_batch_variant(self._h).decode()

    def kernel_name(self) -> str:
        """name of the HIP kernel a rollout of this batch launches"""
        return self._lib.c.mjhip_batch_kernel(self._h).decode()

    def set_mfma(self, on: bool) -> None:
        """AR = Y Y' on the matrix cores (tolerance parity instead of bit parity)"""
        self._lib.check(self._lib.c.mjhip_batch_set_mfma(self._h, 1 if on else 0), "set_mfma")

    def set_pgs_mode(self, mode: int) -> None:
        """0: the reference's PGS sweep, bit for bit (default); 1: residual-update sweep (tolerance parity, opt-in)"""
        self._lib.check(self._lib.c.mjhip_batch_set_pgs_mode(self._h, int(mode)), "set_pgs_mode")

    def set_variant(self, name: str) -> None:
        self._lib.check(self._lib.c.mjhip_batch_set_variant(self._h, name.encode()), f"set_variant {name}")

    def lds_report(self) -> str:
        return self._lib.c.mjhip_batch_lds_report(self._h).decode()

    def step(self, nstep: int = 1, stream: int = 0, sync: bool = True) -> None:
        self._lib.check(self._lib.c.mjhip_batch_step(self._h, int(nstep), stream or None), "step")
        if sync:
            self.sync(stream)

    def step1(self, stream: int = 0) -> None:
        """mj_step1 for every env: everything up to (and including) the velocity stage"""
        self._lib.check(self._lib.c.mjhip_batch_step1(self._h, stream or None), "step1")
        self.sync(stream)

    def step2(self, stream: int = 0) -> None:
        """mj_step2 for every env: actuation, acceleration, constraint solve, integration"""
        self._lib.check(self._lib.c.mjhip_batch_step2(self._h, stream or None), "step2")
        self.sync(stream)

    def sync(self, stream: int = 0) -> None:
        self._lib.check(self._lib.c.mjhip_batch_sync(self._h, stream or None), "sync")

    def rollout_host(self, nstep: int, control_spec: int, state0, warmstart0=None, control=None,
                     want_state: bool = True, want_sensordata: bool = False):
        """host-array rollout of every env (numpy in / numpy out).  With want_sensordata the
        result is the pair (state, sensordata[nenv, nstep, nsensordata])."""
        nstate = self.model.size("nstate")
        s0 = np.ascontiguousarray(state0, dtype=np.float64)
        assert s0.shape == (self.nenv, nstate), s0.shape
        ws = None if warmstart0 is None else np.ascontiguousarray(warmstart0, dtype=np.float64)
        ct = None if control is None else np.ascontiguousarray(control, dtype=np.float64)
        out = np.zeros((self.nenv, nstep, nstate)) if want_state else None
        sd = np.zeros((self.nenv, nstep, self.model.size("nsensordata"))) if want_sensordata else None
        rc = self._lib.c.mjhip_batch_rollout_sensors(
            self._h, int(nstep), int(control_spec), s0.ctypes.data,
            None if ws is None else ws.ctypes.data, None if ct is None else ct.ctypes.data,
            None if out is None else out.ctypes.data, None if sd is None or sd.size == 0 else sd.ctypes.data,
            0, None)
        self._lib.check(rc, "rollout")
        return (out, sd) if want_sensordata else out

    def rollout_device(self, nstep: int, control_spec: int, state0_ptr: int, warmstart0_ptr: int,
                       control_ptr: int, state_ptr: int, stream: int = 0, cont: bool = False) -> None:
        """device-pointer rollout (asynchronous on `stream`).  cont=True continues from the batch's
        current state / warm start / warnings (MJHIP_ROLLOUT_CONTINUE) instead of loading state0."""
        rc = self._lib.c.mjhip_batch_rollout(
            self._h, int(nstep), int(control_spec), state0_ptr or None, warmstart0_ptr or None,
            control_ptr or None, state_ptr or None, 1 | (2 if cont else 0), stream or None)
        self._lib.check(rc, "rollout")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.c.mjhip_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
