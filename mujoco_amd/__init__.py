"""mujoco_amd -- MI355X-native batched mj_step behind MuJoCo's rollout API.

Only the hot path lives here: `mujoco_amd.rollout` mirrors `mujoco.rollout` (python/mujoco/rollout.py
of the reference) and `mujoco_amd.Batch` exposes device-resident closed-loop stepping.  The physics
runs in hand-written HIP kernels (mujoco_amd/csrc, libmjhip.so, C ABI in include/mjhip.h); there is
no CPU fallback -- importing works without a GPU, computing does not.
"""
from __future__ import annotations

import os

from . import _capi
from ._capi import (Batch, DeviceModel, Lib, MjbModel, MjhipError, STAGE_ALL, mjSTATE_CTRL,
                    mjSTATE_FULLPHYSICS, mjSTATE_QFRC_APPLIED)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MJHIP_LIB", os.path.join(_HERE, "csrc", "libmjhip.so"))
_lib = None


def lib() -> Lib:
    """the product library (HIP, gfx950); raises MjhipError if it has not been built."""
    global _lib
    if _lib is None:
        _lib = Lib(LIB_PATH)
    return _lib


__all__ = ["Batch", "DeviceModel", "Lib", "MjbModel", "MjhipError", "lib", "LIB_PATH", "STAGE_ALL",
           "mjSTATE_CTRL", "mjSTATE_FULLPHYSICS", "mjSTATE_QFRC_APPLIED"]
