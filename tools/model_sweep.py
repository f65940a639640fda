#!/usr/bin/env python3
"""Sweep over the reference's own model files: which does mjhip accept, and do the accepted ones
reproduce the reference?

Runs HERE (the build container), not on the GPU box: it reads model files from the reference tree
($MUJOCO_REF, default /root/reference), steps each with the compiled-reference oracle (oracle/_ref)
and with the mjhip kernels on the host wavefront emulation (tests/hostsim: same sources, same C ABI,
lanes scheduled on the CPU).  Test infrastructure: nothing in the product imports this.

  python tools/model_sweep.py [--steps 15] [--nvmax 80] [--out profiles/r02_sweep] [--only substr]

For each model that the reference's compiler loads (nv <= nvmax, nv > 0):
  * upload through mjhip_model_upload: accepted, or rejected with the reason mjhip gives;
  * accepted: `steps` mj_step calls from the first keyframe (or the reset state), ctrl = 0, as shipped;
    then again with the solver forced to PGS, to Newton with elliptic cones, and with the RK4 and
    implicitfast integrators (when the model's features allow the variation);
  * the worst relative deviation of the FULLPHYSICS state (and sensordata) over the trajectory, the
    integer observables (ncon, nefc) and mjhip's warnings are recorded.
Output: <out>/sweep.txt (one line per model and variation + census of rejection reasons), also printed.
"""
import argparse
import collections
import glob
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from mujoco_amd import _capi as K          # noqa: E402
from oracle import refbind as rb           # noqa: E402

REF = os.environ.get("MUJOCO_REF", "/root/reference")
HOSTSIM_LIB = os.path.join(ROOT, "tests", "hostsim", "libmjhip_hostsim.so")

mjSOL_PGS, mjSOL_CG, mjSOL_NEWTON = 0, 1, 2
mjINT_EULER, mjINT_RK4, mjINT_IMPLICIT, mjINT_IMPLICITFAST = 0, 1, 2, 3
mjCONE_PYRAMIDAL, mjCONE_ELLIPTIC = 0, 1

VARIATIONS = [
    ("as-shipped", {}),
    ("pgs", {"solver": mjSOL_PGS}),
    ("newton+elliptic", {"solver": mjSOL_NEWTON, "cone": mjCONE_ELLIPTIC}),
    ("rk4", {"integrator": mjINT_RK4}),
    ("implicitfast", {"integrator": mjINT_IMPLICITFAST}),
]


def model_files(only):
    pats = ["model/**/*.xml", "test/**/testdata/**/*.xml", "test/**/testdata/*.xml"]
    seen = []
    for p in pats:
        for f in sorted(glob.glob(os.path.join(REF, p), recursive=True)):
            if f not in seen and (not only or only in f):
                seen.append(f)
    return seen


def rel(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    if a.size == 0:
        return 0.0
    if not (np.all(np.isfinite(a)) and np.all(np.isfinite(b))):
        return float("inf") if not np.array_equal(np.isfinite(a), np.isfinite(b)) else 0.0
    return float(np.max(np.abs(a - b) / (1.0 + np.abs(b))))


def run_one(lib, path, steps, lds, changes, progress=None):
    """returns (status, detail).  status in {'ok', 'deviates', 'warning', 'rejected', 'noload', 'skip'}"""
    try:
        m = rb.MjModel.from_xml_path(path)
    except Exception as ex:                                   # the reference's compiler refuses it (or a stub dependency)
        return "noload", str(ex).splitlines()[0][:100] if str(ex) else "compile error"
    if m.nv == 0:
        return "skip", "nv = 0"
    for k, v in changes.items():
        setattr(m.opt, k, v)
    try:
        dm = K.DeviceModel(lib, m)
    except K.MjhipError as ex:
        return "rejected", str(ex)
    d = rb.MjData(m)
    if m.nkey > 0:
        rb.mj_resetDataKeyframe(m, d, 0)
    else:
        rb.mj_resetData(m, d)
    spec = rb.mjSTATE_FULLPHYSICS
    s0 = rb.mj_getState(m, d, spec).copy()
    b = K.Batch(dm, 1)
    b.plan_lds(lds)
    nstate = s0.size
    ctrl = np.zeros((1, steps, max(m.nu, 0)))
    # the reference trajectory
    ref = np.zeros((steps, nstate)); ref_int = np.zeros((steps, 2), int); ref_sens = np.zeros((steps, m.nsensordata))
    ctrl0 = np.array(d.ctrl).copy() if m.nu > 0 else None      # (a keyframe may carry controls: pulley.xml)
    mocap = None
    if m.nmocap > 0:
        mocap = (np.array(d.mocap_pos).copy(), np.array(d.mocap_quat).copy())
    for t in range(steps):
        rb.mj_step(m, d)
        ref[t] = rb.mj_getState(m, d, spec)
        ref_int[t] = (d.ncon, d.nefc)
        if m.nsensordata:
            ref_sens[t] = d.sensordata
    ref_warn = sum(d.warning_number(i) for i in range(7))
    if progress: progress()                                    # (the reference's trajectory is done)
    # mjhip: closed-loop steps so that the integer observables can be read after each
    b.reset()
    b.set("time", s0[None, :1]); b.set("qpos", s0[None, 1:1 + m.nq]); b.set("qvel", s0[None, 1 + m.nq:1 + m.nq + m.nv])
    if m.na > 0:
        b.set("act", s0[None, 1 + m.nq + m.nv:1 + m.nq + m.nv + m.na])
    if mocap is not None:
        b.set("mocap_pos", mocap[0].reshape(1, -1)); b.set("mocap_quat", mocap[1].reshape(1, -1))
    if ctrl0 is not None:
        b.set("ctrl", ctrl0[None, :])
    worst = 0.0; worst_sens = 0.0; int_ok = True
    for t in range(steps):
        b.step(1)
        got = np.concatenate([b.get("time")[0, :1], b.get("qpos")[0], b.get("qvel")[0]] + ([b.get("act")[0]] if m.na > 0 else []))
        worst = max(worst, rel(got, ref[t, :got.size]))
        c = b.get("counts")[0]
        if (int(c[0]), int(c[1])) != tuple(ref_int[t]):
            int_ok = False
        if m.nsensordata:
            worst_sens = max(worst_sens, rel(b.get("sensordata")[0], ref_sens[t]))
    warn = b.get("warning")[0]
    detail = f"state {worst:.1e}" + (f" sensor {worst_sens:.1e}" if m.nsensordata else "") + \
             f" counts {'exact' if int_ok else 'differ'} ncon {int(ref_int[-1, 0])} nefc {int(ref_int[-1, 1])} nv {m.nv}"
    if warn.sum() > 0:
        names = ["inertia", "contactfull", "cnstrfull", "vgeomfull", "badqpos", "badqvel", "badqacc", "unsupported"]
        which = ",".join(names[i] if i < len(names) else str(i) for i in np.nonzero(warn)[0])
        return ("warning" if ref_warn == 0 else "ok"), detail + f" mjhip warnings: {which}" + (" (reference warns too)" if ref_warn else "")
    if worst > 1e-6 or worst_sens > 1e-6:
        return "deviates", detail
    return "ok", detail


def run_isolated(lib, path, steps, lds, changes):
    """run_one in a forked child: a crash of the compiled reference (sphere_radial.xml forced to PGS overflows its 10 MB
    arena and the reference's own mj_step then dies) or of the emulation is a status of that model, not the end of the sweep"""
    import pickle
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        os.close(r)
        try:
            res = run_one(lib, path, steps, lds, changes, progress=lambda: os.write(w, b"R"))
        except Exception as ex:                                # a failure of the harness itself: reported, not hidden
            res = ("error", f"{type(ex).__name__}: {ex}")
        os.write(w, b"D" + pickle.dumps(res))
        os._exit(0)
    os.close(w)
    with os.fdopen(r, "rb") as fh:
        data = fh.read()
    _, status = os.waitpid(pid, 0)
    if b"D" in data[:2]:
        return pickle.loads(data[data.index(b"D") + 1:])
    sig = status & 0x7f
    who = "the emulation of the kernels" if data[:1] == b"R" else "the compiled REFERENCE (before mjhip ran a step)"
    return "crash", f"the process died (signal {sig}) inside {who}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=15)
    ap.add_argument("--nvmax", type=int, default=80)
    ap.add_argument("--lds", type=int, default=40960)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_sweep"))
    ap.add_argument("--only", default="")
    ap.add_argument("--variations", default="all")
    args = ap.parse_args()
    if not os.path.isdir(os.path.join(REF, "model")):
        sys.exit(f"reference tree not found at {REF} (this tool runs in the build container)")
    lib = K.Lib(HOSTSIM_LIB)
    files = model_files(args.only)
    lines = []
    census = collections.Counter()
    status_count = collections.Counter()
    t0 = time.time()
    for f in files:
        short = os.path.relpath(f, REF)
        try:
            m = rb.MjModel.from_xml_path(f)
            nv = m.nv
            del m
        except Exception as ex:
            status_count["noload"] += 1
            lines.append(f"noload     {short}: {str(ex).splitlines()[0][:90] if str(ex) else ''}")
            continue
        if nv == 0 or nv > args.nvmax:
            status_count["skip"] += 1
            lines.append(f"skip       {short}: nv = {nv}")
            continue
        for name, changes in VARIATIONS:
            if args.variations != "all" and name not in args.variations.split(","):
                continue
            st, detail = run_isolated(lib, f, args.steps, args.lds, changes)
            if name == "as-shipped":
                status_count[st] += 1
                if st == "rejected":
                    reason = re.sub(r"\s*\(.*", "", detail.split(":")[-1].strip())[:70]
                    census[reason] += 1
            else:
                status_count[f"{name}:{st}"] += 1
            lines.append(f"{st:10s} {short} [{name}]: {detail}")
            if st in ("rejected", "noload", "skip") and name == "as-shipped":
                break
        print(lines[-1], flush=True)
    out = []
    out.append(f"# model sweep: {len(files)} files under {REF}/model and test/**/testdata, {args.steps} steps, nv <= {args.nvmax}, "
               f"hostsim with a {args.lds} B LDS plan; {time.time() - t0:.0f} s")
    out.append("# as-shipped status counts: " + ", ".join(f"{k} {v}" for k, v in sorted(status_count.items()) if ":" not in k))
    out.append("# variations: " + ", ".join(f"{k} {v}" for k, v in sorted(status_count.items()) if ":" in k))
    out.append("# rejection census (as shipped):")
    for k, v in census.most_common():
        out.append(f"#   {v:3d}  {k}")
    out += lines
    os.makedirs(args.out, exist_ok=True)
    with open(os.path.join(args.out, "sweep.txt"), "w") as fh:
        fh.write("\n".join(out) + "\n")
    print("\n".join(out[:4 + len(census)]))


if __name__ == "__main__":
    main()
