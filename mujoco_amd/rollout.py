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
        """Same contract as `mujoco.rollout.Rollout.rollout` (python/mujoco/rollout.py:87-241): returns
        (state[nbatch, nstep, nstate], sensordata[nbatch, nstep, nsensordata]).  `chunk_size` is accepted and has
        no effect (it tunes the reference's thread pool)."""
        if self.rollout_ is None:
            raise RuntimeError("rollout requested after thread pool shutdown")
        args = {"initial_state": initial_state, "initial_warmstart": initial_warmstart, "control": control,
                "state": state, "sensordata": sensordata}
        if not skip_checks:
            model, data, nstep, args = _prepare(model, data, nstep, chunk_size, control_spec, args)
        self._call(model, data, nstep, control_spec, args["initial_state"], args["initial_warmstart"],
                   args["control"], args["state"], args["sensordata"])
        return args["state"], args["sensordata"]


# One row per array argument of a rollout: its rank once the leading singleton axes are added back, and which model
# size its last axis must have.  The reference checks the same things argument by argument (rollout.py:129-212).
_ARGS = (("initial_state", 2, "nstate"), ("initial_warmstart", 2, "nv"), ("control", 3, "ncontrol"),
         ("state", 3, "nstate"), ("sensordata", 3, "nsensordata"))


def _model_sizes(m, control_spec):
    return {"nstate": state_size(m, mjSTATE_FULLPHYSICS), "ncontrol": state_size(m, control_spec), "nv": int(m.nv),
            "nsensordata": int(getattr(m, "nsensordata", 0) or 0)}


def _agree(axis, current, arrays):
    """the common extent of `axis` over the given (name, array) pairs; 1 stretches to anything"""
    for name, a in arrays:
        n = a.shape[axis]
        if n == current or n == 1:
            continue
        if current != 1:
            raise ValueError(f"dimension {axis} inferred as {current} but {name} has {n}")
        current = n
    return current


def _prepare(model, data, nstep, chunk_size, control_spec, args):
    """validation, shape inference and singleton expansion of a rollout call; returns what `_call` needs"""
    if control_spec & ~mjSTATE_USER:
        raise ValueError("control_spec can only contain bits in mjSTATE_USER")
    for label, v in (("nstep", nstep), ("chunk_size", chunk_size)):
        if v and not isinstance(v, int):
            raise ValueError(f"{label} must be an integer")
    models = [model] if _is_model(model) else list(model)
    given = []
    for name, rank, _ in _ARGS:
        v = args[name]
        if v is None:
            continue
        if not isinstance(v, (np.ndarray, float)):
            raise ValueError(f"{name} must be a numpy array or float")
        if np.ndim(v) > rank:
            raise ValueError(f"{name} can have at most {rank} dimensions")
        a = np.asarray(v, dtype=np.float64)
        a = np.ascontiguousarray(a.reshape((1,) * (rank - a.ndim) + a.shape))
        args[name] = a
        given.append((name, a))
    nbatch = _agree(0, 1, given)
    if nbatch == 1 and len(models) > 1:
        nbatch = len(models)
    if len(models) > 1 and len(models) != nbatch:
        raise ValueError(f"nbatch inferred as {nbatch} but model is length {len(models)}")
    nstep = _agree(1, nstep or 1, [(n, a) for n, a in given if a.ndim == 3])
    sizes = _model_sizes(models[0], control_spec)
    if any(_model_sizes(m, control_spec) != sizes for m in models[1:]):
        raise ValueError("models are not compatible")
    for name, _, key in _ARGS:
        a = args[name]
        if a is not None and a.shape[-1] != sizes[key]:
            raise ValueError(f"trailing dimension of {name} must be {sizes[key]}, got {a.shape[-1]}")
    # inputs given once are repeated for every rollout (and step); outputs are allocated when absent
    for name, rank, key in _ARGS:
        a = args[name]
        if name in ("state", "sensordata"):
            if a is None:
                args[name] = np.empty((nbatch, nstep, sizes[key]), dtype=np.float64)
        elif a is not None:
            full = (nbatch,) + ((nstep,) if rank == 3 else ()) + (a.shape[-1],)
            if a.shape != full:
                args[name] = np.ascontiguousarray(np.broadcast_to(a, full))
    if len(models) == 1:
        models = models * nbatch
    return models, (data if isinstance(data, list) else [data]), nstep, args


_shared = None          # the Rollout object kept alive by rollout(..., persistent_pool=True)


def shutdown_persistent_pool():
    """drop the persistent Rollout object that `rollout(..., persistent_pool=True)` may have created"""
    global _shared
    if _shared is not None:
        _shared.close()
        _shared = None


atexit.register(shutdown_persistent_pool)


def rollout(model, data, initial_state, control=None, *, control_spec: int = mjSTATE_CTRL,
            skip_checks: bool = False, nstep: Optional[int] = None, initial_warmstart=None,
            state=None, sensordata=None, chunk_size: Optional[int] = None,
            persistent_pool: bool = False):
    """Module-level entry point with the signature of `mujoco.rollout.rollout` (python/mujoco/rollout.py:261);
    returns (state, sensordata).  The number of mjData objects stands for the reference's thread count and only
    decides whether a persistent object is rebuilt."""
    global _shared
    nthread = len(data) if isinstance(data, list) and len(data) > 1 else 0
    if persistent_pool:
        if _shared is None or _shared.nthread != nthread:
            shutdown_persistent_pool()
            _shared = Rollout(nthread=nthread)
        runner = _shared
    else:
        runner = Rollout(nthread=nthread)
    try:
        return runner.rollout(model, data, initial_state, control, control_spec=control_spec, skip_checks=skip_checks,
                              nstep=nstep, initial_warmstart=initial_warmstart, state=state, sensordata=sensordata,
                              chunk_size=chunk_size)
    finally:
        if runner is not _shared:
            runner.close()
