"""Drop-in for `mujoco.rollout` (reference: python/mujoco/rollout.py:27-346, rollout.cc:250-326).

Same call signature, shape inference, singleton tiling, error types and return values as the
reference module; the dispatch underneath is replaced: instead of a CPU thread pool stepping one
mjData per thread, the whole batch is uploaded to the GPU and stepped by libmjhip's HIP kernels
(one wavefront per rollout) through `mjhip_rollout` (include/mjhip.h).

`model` / `data` are the caller's MuJoCo objects (anything exposing `_address`, as the official
bindings do).  `data` is only used as the place where the final state of the last rollout is left
(rollout.cc:73); its length no longer sets the parallelism (`nthread` is accepted and ignored).
"""
from __future__ import annotations

import atexit
import ctypes as C
from collections.abc import Sequence
from typing import Optional

import numpy as np

from . import _capi
from ._capi import (mjSTATE_ACT, mjSTATE_CTRL, mjSTATE_EQ_ACTIVE, mjSTATE_FULLPHYSICS,
                    mjSTATE_HISTORY, mjSTATE_MOCAP_POS, mjSTATE_MOCAP_QUAT, mjSTATE_PLUGIN,
                    mjSTATE_QFRC_APPLIED, mjSTATE_QPOS, mjSTATE_QVEL, mjSTATE_TIME,
                    mjSTATE_USERDATA, mjSTATE_WARMSTART, mjSTATE_XFRC_APPLIED)

mjSTATE_USER = (mjSTATE_CTRL | mjSTATE_QFRC_APPLIED | mjSTATE_XFRC_APPLIED | mjSTATE_EQ_ACTIVE |
                mjSTATE_MOCAP_POS | mjSTATE_MOCAP_QUAT | mjSTATE_USERDATA)


def _lib():
    from . import lib
    return lib()


def _is_model(obj) -> bool:
    return hasattr(obj, "_address") and hasattr(obj, "nq")


def state_size(model, spec: int) -> int:
    """mj_stateSize (src/engine/engine_support.c:190) from the model's size fields."""
    def g(name):
        return int(getattr(model, name, 0) or 0)
    sizes = {
        mjSTATE_TIME: 1, mjSTATE_QPOS: g("nq"), mjSTATE_QVEL: g("nv"), mjSTATE_ACT: g("na"),
        mjSTATE_HISTORY: g("nhistory"), mjSTATE_WARMSTART: g("nv"), mjSTATE_CTRL: g("nu"),
        mjSTATE_QFRC_APPLIED: g("nv"), mjSTATE_XFRC_APPLIED: 6 * g("nbody"),
        mjSTATE_EQ_ACTIVE: g("neq"), mjSTATE_MOCAP_POS: 3 * g("nmocap"),
        mjSTATE_MOCAP_QUAT: 4 * g("nmocap"), mjSTATE_USERDATA: g("nuserdata"),
        mjSTATE_PLUGIN: g("npluginstate"),
    }
    return sum(n for bit, n in sizes.items() if spec & bit)


class Rollout:
    """Rollout object (reference: a thread pool; here: a handle on the GPU library)."""

    def __init__(self, *, nthread: Optional[int] = None):
        self.nthread = 0 if nthread is None else nthread
        self.rollout_ = _lib()

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.close()

    def close(self):
        self.rollout_ = None

    def _call(self, model, data, nstep, control_spec, initial_state, initial_warmstart, control,
              state, sensordata):
        lib = self.rollout_
        nbatch = initial_state.shape[0]
        models = list(model) if not _is_model(model) else [model] * nbatch
        datas = list(data) if isinstance(data, (list, tuple)) else [data]
        if len(models) != nbatch:
            raise ValueError(f"model list has length {len(models)}, expected nbatch={nbatch}")
        mp = (C.c_void_p * nbatch)(*[_capi._address(m) for m in models])
        dp = (C.c_void_p * len(datas))(*[_capi._address(d) for d in datas])

        def ptr(a):
            return None if a is None else a.ctypes.data
        rc = lib.c.mjhip_rollout(mp, dp, nbatch, int(nstep), int(control_spec), ptr(initial_state),
                                 ptr(initial_warmstart), ptr(control), ptr(state), ptr(sensordata))
        if rc < 0:
            msg = lib.error()
            # size/shape/feature problems are the caller's: same exception type as the reference
            raise ValueError(msg) if rc in (-1, -2) else RuntimeError(msg)
        if rc == 2:
            # an environment reached a collider the GPU path lacks: its trajectory is not the reference's
            raise RuntimeError(lib.error())
        if rc == 1:
            # contact / constraint capacity overflow: the affected rollouts were frozen (back-filled)
            import warnings
            warnings.warn(lib.error(), RuntimeWarning, stacklevel=3)

    def rollout(self, model, data, initial_state, control=None, *, control_spec: int = mjSTATE_CTRL,
                skip_checks: bool = False, nstep: Optional[int] = None, initial_warmstart=None,
                state=None, sensordata=None, chunk_size: Optional[int] = None):
        """See `mujoco.rollout.Rollout.rollout`; returns (state, sensordata)."""
        if self.rollout_ is None:
            raise RuntimeError("rollout requested after thread pool shutdown")

        if skip_checks:
            self._call(model, data, nstep, control_spec, initial_state, initial_warmstart, control,
                       state, sensordata)
            return state, sensordata

        if not _is_model(model):
            model = list(model)
        if control_spec & ~mjSTATE_USER:
            raise ValueError("control_spec can only contain bits in mjSTATE_USER")
        if nstep and not isinstance(nstep, int):
            raise ValueError("nstep must be an integer")
        if chunk_size and not isinstance(chunk_size, int):
            raise ValueError("chunk_size must be an integer")
        _check_must_be_numeric(initial_state=initial_state, initial_warmstart=initial_warmstart,
                               control=control, state=state, sensordata=sensordata)
        _check_number_of_dimensions(2, initial_state=initial_state, initial_warmstart=initial_warmstart)
        _check_number_of_dimensions(3, control=control, state=state, sensordata=sensordata)

        initial_state = _ensure_2d(initial_state)
        initial_warmstart = _ensure_2d(initial_warmstart)
        control = _ensure_3d(control)
        state = _ensure_3d(state)
        sensordata = _ensure_3d(sensordata)

        nbatch = _infer_dimension(0, 1, initial_state=initial_state, initial_warmstart=initial_warmstart,
                                  control=control, state=state, sensordata=sensordata)
        if isinstance(model, list) and nbatch == 1:
            nbatch = len(model)
        if isinstance(model, list) and len(model) > 1 and len(model) != nbatch:
            raise ValueError(f"nbatch inferred as {nbatch} but model is length {len(model)}")
        elif not isinstance(model, list):
            model = [model]
        if not isinstance(data, list):
            data = [data]

        nstep = _infer_dimension(1, nstep or 1, control=control, state=state, sensordata=sensordata)

        nstate = state_size(model[0], mjSTATE_FULLPHYSICS)
        ncontrol = state_size(model[0], control_spec)
        nv = int(model[0].nv)
        nsensordata = int(getattr(model[0], "nsensordata", 0) or 0)
        for m in model[1:]:
            if (nstate != state_size(m, mjSTATE_FULLPHYSICS) or ncontrol != state_size(m, control_spec)
                    or nv != int(m.nv) or nsensordata != int(getattr(m, "nsensordata", 0) or 0)):
                raise ValueError("models are not compatible")

        _check_trailing_dimension(nstate, initial_state=initial_state, state=state)
        _check_trailing_dimension(ncontrol, control=control)
        _check_trailing_dimension(nv, initial_warmstart=initial_warmstart)
        _check_trailing_dimension(nsensordata, sensordata=sensordata)

        model = model * nbatch if len(model) == 1 else model
        initial_state = _tile_if_required(initial_state, nbatch)
        initial_warmstart = _tile_if_required(initial_warmstart, nbatch)
        control = _tile_if_required(control, nbatch, nstep)

        if state is None:
            state = np.empty((nbatch, nstep, nstate), dtype=np.float64)
        if sensordata is None:
            sensordata = np.empty((nbatch, nstep, nsensordata), dtype=np.float64)

        self._call(model, data, nstep, control_spec, initial_state, initial_warmstart, control,
                   state, sensordata)
        return state, sensordata


persistent_rollout = None


def shutdown_persistent_pool():
    """Shut down the persistent Rollout object optionally created by `rollout`."""
    global persistent_rollout
    if persistent_rollout is not None:
        persistent_rollout.close()
    persistent_rollout = None


atexit.register(shutdown_persistent_pool)


def rollout(model, data, initial_state, control=None, *, control_spec: int = mjSTATE_CTRL,
            skip_checks: bool = False, nstep: Optional[int] = None, initial_warmstart=None,
            state=None, sensordata=None, chunk_size: Optional[int] = None,
            persistent_pool: bool = False):
    """See `mujoco.rollout.rollout` (python/mujoco/rollout.py:261); returns (state, sensordata)."""
    if not isinstance(data, list):
        data = [data]
    nthread = len(data) if len(data) > 1 else 0
    global persistent_rollout
    if persistent_pool:
        if persistent_rollout is None:
            persistent_rollout = Rollout(nthread=nthread)
        if persistent_rollout.nthread != nthread:
            persistent_rollout.close()
            persistent_rollout = Rollout(nthread=nthread)
        rollout_ = persistent_rollout
    else:
        rollout_ = Rollout(nthread=nthread)
    try:
        return rollout_.rollout(model, data, initial_state, control, control_spec=control_spec,
                                skip_checks=skip_checks, nstep=nstep,
                                initial_warmstart=initial_warmstart, state=state,
                                sensordata=sensordata, chunk_size=chunk_size)
    finally:
        if not persistent_pool:
            rollout_.close()


def _check_must_be_numeric(**kwargs):
    for key, value in kwargs.items():
        if value is None:
            continue
        if not isinstance(value, np.ndarray) and not isinstance(value, float):
            raise ValueError(f"{key} must be a numpy array or float")


def _check_number_of_dimensions(ndim, **kwargs):
    for key, value in kwargs.items():
        if value is None:
            continue
        if np.ndim(value) > ndim:
            raise ValueError(f"{key} can have at most {ndim} dimensions")


def _check_trailing_dimension(dim, **kwargs):
    for key, value in kwargs.items():
        if value is None:
            continue
        if value.shape[-1] != dim:
            raise ValueError(f"trailing dimension of {key} must be {dim}, got {value.shape[-1]}")


def _ensure_2d(arg):
    if arg is None:
        return None
    return np.ascontiguousarray(np.atleast_2d(arg), dtype=np.float64)


def _ensure_3d(arg):
    if arg is None:
        return None
    arg = np.asarray(arg)
    while arg.ndim < 3:
        arg = arg[np.newaxis, ...]   # leading singleton dims only
    return np.ascontiguousarray(arg, dtype=np.float64)


def _infer_dimension(dim, value, **kwargs):
    for name, array in kwargs.items():
        if array is None:
            continue
        if array.shape[dim] != value:
            if value == 1:
                value = array.shape[dim]
            elif array.shape[dim] != 1:
                raise ValueError(f"dimension {dim} inferred as {value} but {name} has {array.shape[dim]}")
    return value


def _tile_if_required(array, dim0, dim1=None):
    if array is None:
        return None
    reps = np.ones(array.ndim, dtype=int)
    if array.shape[0] == 1:
        reps[0] = dim0
    if dim1 is not None and array.shape[1] == 1:
        reps[1] = dim1
    return np.tile(array, reps)
