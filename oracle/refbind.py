"""ctypes binding of the compiled reference engine (TEST INFRASTRUCTURE, not product).

`oracle/_ref/liboracle.so` is the reference's own src/engine + src/user + src/xml built
by `oracle/Makefile`; `libintrospect.so` exposes mjModel/mjData fields by name through
the reference's X-macros.  This module gives the tests, `smoke()` and bench.py's
`cpu_baseline` leg a minimal stand-in for the official `mujoco` Python package:

    lib = load()                     # liboracle.so   (parity build)
    m = MjModel.from_xml_path(p)     # mj_loadXML     (include/mujoco/mujoco.h:132)
    m = MjModel.from_binary_path(p)  # mj_loadModel   (:236)
    d = MjData(m)                    # mj_makeData    (:249)
    mj_step(m, d)                    # (:189)
    d.qpos, m.body_pos, d.contact    # numpy views into the C structs

Objects expose `_address` (the raw mjModel*/mjData* value), the same attribute the
official bindings provide, which is what `mujoco_amd.rollout` consumes.

Nothing under `mujoco_amd/` may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF = os.path.join(_HERE, "_ref")

_libs = {}
_intro = None

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


_NAMES = {"parity": "liboracle.so", "fast": "liboracle_fast.so", "devmath": "liboracle_dm.so"}
_default_kind = "parity"


def use(kind: str) -> str:
    """Select the build the module-level functions (mj_step, ...) call from now on; returns the previous choice.
    'parity': the reference as built (glibc libm) -- the oracle; 'devmath': the same objects with sin / cos bound to
    the kernels' mjh_sincos (oracle/devmath_shim.cc).  mjModel / mjData are plain C structs shared by all builds."""
    global _default_kind
    prev, _default_kind = _default_kind, kind
    return prev


def available(kind: str = "parity") -> bool:
    name = _NAMES[kind]
    return os.path.exists(os.path.join(_REF, name)) and os.path.exists(
        os.path.join(_REF, "libintrospect.so"))


def load(kind: Optional[str] = None):
    """Load the compiled reference engine. kind: 'parity' (-O2, no FMA/SIMD), 'devmath' or 'fast'; default: use()'s choice."""
    kind = kind or _default_kind
    if kind in _libs:
        return _libs[kind]
    name = _NAMES[kind]
    path = os.path.join(_REF, name)
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} missing: run `make -C oracle` (needs /root/reference)")
    lib = C.CDLL(path, mode=C.RTLD_LOCAL)
    vp = C.c_void_p
    lib.mj_loadXML.restype = vp
    lib.mj_loadXML.argtypes = [C.c_char_p, vp, C.c_char_p, C.c_int]
    lib.mj_loadModel.restype = vp
    lib.mj_loadModel.argtypes = [C.c_char_p, vp]
    lib.mj_loadModelBuffer.restype = vp
    lib.mj_loadModelBuffer.argtypes = [vp, C.c_int]
    lib.mj_saveModel.restype = None
    lib.mj_saveModel.argtypes = [vp, C.c_char_p, vp, C.c_int]
    lib.mj_deleteModel.restype = None
    lib.mj_deleteModel.argtypes = [vp]
    lib.mj_makeData.restype = vp
    lib.mj_makeData.argtypes = [vp]
    lib.mj_deleteData.restype = None
    lib.mj_deleteData.argtypes = [vp]
    lib.mj_copyData.restype = vp
    lib.mj_copyData.argtypes = [vp, vp, vp]
    for fn in ("mj_step", "mj_forward", "mj_resetData", "mj_kinematics", "mj_step1", "mj_step2",
               "mj_fwdPosition", "mj_fwdVelocity", "mj_collision", "mj_comPos", "mj_makeConstraint"):
        f = getattr(lib, fn)
        f.restype = None
        f.argtypes = [vp, vp]
    lib.mj_resetDataKeyframe.restype = None
    lib.mj_resetDataKeyframe.argtypes = [vp, vp, C.c_int]
    lib.mj_stateSize.restype = C.c_int
    lib.mj_stateSize.argtypes = [vp, C.c_int]
    lib.mj_getState.restype = None
    lib.mj_getState.argtypes = [vp, vp, vp, C.c_int]
    lib.mj_setState.restype = None
    lib.mj_setState.argtypes = [vp, vp, vp, C.c_int]
    lib.mj_name2id.restype = C.c_int
    lib.mj_name2id.argtypes = [vp, C.c_int, C.c_char_p]
    lib.mj_id2name.restype = C.c_char_p
    lib.mj_id2name.argtypes = [vp, C.c_int, C.c_int]
    lib.mj_versionString.restype = C.c_char_p
    _libs[kind] = lib
    return lib


def _introspect():
    global _intro
    if _intro is None:
        lib = C.CDLL(os.path.join(_REF, "libintrospect.so"))
        vp = C.c_void_p
        lib.mjo_model_size.restype = C.c_longlong
        lib.mjo_model_size.argtypes = [vp, C.c_char_p]
        lib.mjo_model_field.restype = C.c_int
        lib.mjo_model_field.argtypes = [vp, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_longlong),
                                        C.POINTER(C.c_longlong), C.c_char_p, C.POINTER(C.c_int)]
        lib.mjo_data_field.restype = C.c_int
        lib.mjo_data_field.argtypes = [vp, vp, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_longlong),
                                       C.POINTER(C.c_longlong), C.c_char_p, C.POINTER(C.c_int)]
        lib.mjo_data_scalar.restype = C.c_int
        lib.mjo_data_scalar.argtypes = [vp, C.c_char_p, C.POINTER(vp), C.c_char_p]
        lib.mjo_option_field.restype = C.c_int
        lib.mjo_option_field.argtypes = [vp, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_int), C.c_char_p]
        lib.mjo_stat_field.restype = C.c_int
        lib.mjo_stat_field.argtypes = [vp, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_int)]
        lib.mjo_sizeof.restype = C.c_int
        lib.mjo_sizeof.argtypes = [C.c_char_p]
        lib.mjo_contact_offset.restype = C.c_int
        lib.mjo_contact_offset.argtypes = [C.c_char_p]
        lib.mjo_warning_number.restype = C.c_int
        lib.mjo_warning_number.argtypes = [vp, C.c_int]
        lib.mjo_model_field_names.restype = C.c_int
        lib.mjo_model_field_names.argtypes = [C.c_char_p, C.c_int]
        lib.mjo_data_field_names.restype = C.c_int
        lib.mjo_data_field_names.argtypes = [C.c_char_p, C.c_int]
        _intro = lib
    return _intro


_NP = {b"d": np.float64, b"f": np.float32, b"i": np.int32, b"b": np.uint8, b"q": np.int64}


def _view(ptr, nr, nc, tcode, itemsize):
    n = int(nr) * int(nc)
    if tcode in _NP:
        dt = np.dtype(_NP[tcode])
    else:
        dt = np.dtype((np.void, itemsize))
    if n == 0 or not ptr:
        shape = (int(nr),) if nc == 1 else (int(nr), int(nc))
        return np.zeros(shape, dtype=dt)
    buf = (C.c_char * (n * dt.itemsize)).from_address(ptr)
    a = np.frombuffer(buf, dtype=dt, count=n)
    if nc != 1:
        a = a.reshape(int(nr), int(nc))
    return a


def contact_dtype():
    """numpy structured dtype of mjContact (include/mujoco/mjdata.h:37-69), offsets taken
    from the compiled reference struct."""
    it = _introspect()
    off = lambda n: it.mjo_contact_offset(n.encode())
    fields = [
        ("dist", np.float64, ()), ("pos", np.float64, (3,)), ("frame", np.float64, (9,)),
        ("includemargin", np.float64, ()), ("friction", np.float64, (5,)),
        ("solref", np.float64, (2,)), ("solreffriction", np.float64, (2,)),
        ("solimp", np.float64, (5,)), ("mu", np.float64, ()), ("H", np.float64, (36,)),
        ("dim", np.int32, ()), ("geom1", np.int32, ()), ("geom2", np.int32, ()),
        ("geom", np.int32, (2,)), ("flex", np.int32, (2,)), ("elem", np.int32, (2,)),
        ("vert", np.int32, (2,)), ("exclude", np.int32, ()), ("efc_address", np.int32, ()),
    ]
    return np.dtype({
        "names": [f[0] for f in fields],
        "formats": [(f[1], f[2]) if f[2] else f[1] for f in fields],
        "offsets": [off(f[0]) for f in fields],
        "itemsize": it.mjo_sizeof(b"mjContact"),
    })


class _Opt:
    def __init__(self, model):
        object.__setattr__(self, "_m", model)

    def _field(self, name):
        it = _introspect()
        p = C.c_void_p()
        n = C.c_int()
        t = C.create_string_buffer(2)
        if not it.mjo_option_field(self._m._address, name.encode(), C.byref(p), C.byref(n), t):
            raise AttributeError(name)
        return _view(p.value, n.value, 1, t.value[:1], 8)

    def __getattr__(self, name):
        a = self._field(name)
        return a if a.size > 1 else a[0].item()

    def __setattr__(self, name, value):
        a = self._field(name)
        a[...] = value


class _Stat:
    def __init__(self, model):
        self._m = model

    def __getattr__(self, name):
        it = _introspect()
        p = C.c_void_p()
        n = C.c_int()
        if not it.mjo_stat_field(self._m._address, name.encode(), C.byref(p), C.byref(n)):
            raise AttributeError(name)
        a = _view(p.value, n.value, 1, b"d", 8)
        return a if a.size > 1 else a[0].item()


class MjModel:
    def __init__(self, address: int, lib, owned: bool = True):
        if not address:
            raise ValueError("null mjModel")
        self._address = int(address)
        self._lib = lib
        self._owned = owned
        self.opt = _Opt(self)
        self.stat = _Stat(self)

    @classmethod
    def from_xml_path(cls, path: str, kind: Optional[str] = None) -> "MjModel":
        lib = load(kind)
        err = C.create_string_buffer(2000)
        p = lib.mj_loadXML(path.encode(), None, err, 2000)
        if not p:
            raise ValueError(f"mj_loadXML({path}): {err.value.decode(errors='replace')}")
        return cls(p, lib)

    @classmethod
    def from_binary_path(cls, path: str, kind: Optional[str] = None) -> "MjModel":
        lib = load(kind)
        with open(path, "rb") as f:
            blob = f.read()
        buf = C.create_string_buffer(blob, len(blob))
        p = lib.mj_loadModelBuffer(buf, len(blob))
        if not p:
            raise ValueError(f"mj_loadModelBuffer({path}) failed")
        return cls(p, lib)

    def save_binary(self, path: str) -> None:
        self._lib.mj_saveModel(self._address, path.encode(), None, 0)

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        it = _introspect()
        v = it.mjo_model_size(self._address, name.encode())
        if v >= 0:
            return int(v)
        p = C.c_void_p()
        nr = C.c_longlong()
        nc = C.c_longlong()
        t = C.create_string_buffer(2)
        isz = C.c_int()
        if it.mjo_model_field(self._address, name.encode(), C.byref(p), C.byref(nr), C.byref(nc), t,
                              C.byref(isz)):
            return _view(p.value, nr.value, nc.value, t.value[:1], isz.value)
        raise AttributeError(name)

    def name2id(self, objtype: int, name: str) -> int:
        return self._lib.mj_name2id(self._address, objtype, name.encode())

    def __del__(self):
        try:
            if self._owned and self._address:
                self._lib.mj_deleteModel(self._address)
                self._address = 0
        except Exception:
            pass


class MjData:
    def __init__(self, model: MjModel):
        self._model = model
        self._lib = model._lib
        p = self._lib.mj_makeData(model._address)
        if not p:
            raise MemoryError("mj_makeData failed")
        self._address = int(p)

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        it = _introspect()
        p = C.c_void_p()
        t = C.create_string_buffer(2)
        if it.mjo_data_scalar(self._address, name.encode(), C.byref(p), t):
            return _view(p.value, 1, 1, t.value[:1], 8)[0].item()
        nr = C.c_longlong()
        nc = C.c_longlong()
        isz = C.c_int()
        if it.mjo_data_field(self._model._address, self._address, name.encode(), C.byref(p),
                             C.byref(nr), C.byref(nc), t, C.byref(isz)):
            if name == "contact":
                n = int(nr.value)
                dt = contact_dtype()
                if n == 0 or not p.value:
                    return np.zeros(0, dtype=dt)
                buf = (C.c_char * (n * dt.itemsize)).from_address(p.value)
                return np.frombuffer(buf, dtype=dt, count=n)
            return _view(p.value, nr.value, nc.value, t.value[:1], isz.value)
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name.startswith("_"):
            object.__setattr__(self, name, value)
            return
        it = _introspect()
        p = C.c_void_p()
        t = C.create_string_buffer(2)
        if it.mjo_data_scalar(self._address, name.encode(), C.byref(p), t):
            _view(p.value, 1, 1, t.value[:1], 8)[0] = value
            return
        getattr(self, name)[...] = value

    def warning_number(self, i: int) -> int:
        return _introspect().mjo_warning_number(self._address, i)

    def __del__(self):
        try:
            if self._address:
                self._lib.mj_deleteData(self._address)
                self._address = 0
        except Exception:
            pass


def mj_step(m: MjModel, d: MjData) -> None:
    m._lib.mj_step(m._address, d._address)


def mj_step1(m: MjModel, d: MjData) -> None:
    m._lib.mj_step1(m._address, d._address)


def mj_step2(m: MjModel, d: MjData) -> None:
    m._lib.mj_step2(m._address, d._address)


def mj_forward(m: MjModel, d: MjData) -> None:
    m._lib.mj_forward(m._address, d._address)


def mj_resetData(m: MjModel, d: MjData) -> None:
    m._lib.mj_resetData(m._address, d._address)


def mj_resetDataKeyframe(m: MjModel, d: MjData, key: int) -> None:
    m._lib.mj_resetDataKeyframe(m._address, d._address, key)


def mj_stateSize(m: MjModel, spec: int) -> int:
    return m._lib.mj_stateSize(m._address, spec)


def mj_getState(m: MjModel, d: MjData, spec: int) -> np.ndarray:
    out = np.empty(mj_stateSize(m, spec), dtype=np.float64)
    m._lib.mj_getState(m._address, d._address, out.ctypes.data, spec)
    return out


def mj_setState(m: MjModel, d: MjData, state: np.ndarray, spec: int) -> None:
    s = np.ascontiguousarray(state, dtype=np.float64)
    assert s.size == mj_stateSize(m, spec)
    m._lib.mj_setState(m._address, d._address, s.ctypes.data, spec)


mjSTATE_PHYSICS = mjSTATE_QPOS | mjSTATE_QVEL | mjSTATE_ACT | mjSTATE_HISTORY
mjSTATE_FULLPHYSICS = mjSTATE_TIME | mjSTATE_PHYSICS | mjSTATE_PLUGIN
mjSTATE_USER = (mjSTATE_CTRL | mjSTATE_QFRC_APPLIED | mjSTATE_XFRC_APPLIED | mjSTATE_EQ_ACTIVE |
                mjSTATE_MOCAP_POS | mjSTATE_MOCAP_QUAT | mjSTATE_USERDATA)
mjSTATE_INTEGRATION = mjSTATE_FULLPHYSICS | mjSTATE_USER | mjSTATE_WARMSTART
