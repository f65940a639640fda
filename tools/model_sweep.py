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

The same sweep ON THE DEVICE (round 6): the reference tree does not exist on the GPU box, the compiled reference
(oracle/_ref) does.  So
  python tools/model_sweep.py --export tools/sweep_export [--nvmax 320]
writes every model the emulation accepts as <name>.mjb.gz (mj_saveModel of the compiled model: bit-identical constants)
plus an index, here; and on the box
  python tools/model_sweep.py --from-mjb tools/sweep_export --device --out gpurun_out/sweep_gpu
loads each .mjb with the oracle (reference trajectory, live) and with libmjhip.so (the product, on cuda:0) and applies the
same comparison.  `--fixtures DIR --subset file` additionally stores the reference trajectories of a subset as .npz
next to the .mjb files (tests/golden/sweep: what tests/test_gpu_sweep.py replays without the live oracle).
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
    ("implicit", {"integrator": mjINT_IMPLICIT}),
]


def model_files(only):
    pats = ["model/**/*.xml", "test/**/testdata/**/*.xml", "test/**/testdata/*.xml"]
    seen = []
    for p in pats:
        for f in sorted(glob.glob(os.path.join(REF, p), recursive=True)):
            if f not in seen and (not only or only in f):
                seen.append(f)
    return seen


def load_model(path):
    """the oracle's mjModel of an .xml, .mjb or .mjb.gz file (textures make an .mjb large: the exports are gzipped)"""
    if path.endswith(".mjb.gz"):
        import gzip
        import tempfile
        with tempfile.NamedTemporaryFile(suffix=".mjb") as tmp:
            tmp.write(gzip.open(path, "rb").read())
            tmp.flush()
            return rb.MjModel.from_binary_path(tmp.name)
    return rb.MjModel.from_binary_path(path) if path.endswith(".mjb") else rb.MjModel.from_xml_path(path)


def save_gz(m, dst_stem):
    import gzip
    m.save_binary(dst_stem + ".mjb")
    with open(dst_stem + ".mjb", "rb") as fh, gzip.GzipFile(dst_stem + ".mjb.gz", "wb", mtime=0) as out:
        out.write(fh.read())
    os.remove(dst_stem + ".mjb")


def rel(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    if a.size == 0:
        return 0.0
    if not (np.all(np.isfinite(a)) and np.all(np.isfinite(b))):
        return float("inf") if not np.array_equal(np.isfinite(a), np.isfinite(b)) else 0.0
    return float(np.max(np.abs(a - b) / (1.0 + np.abs(b))))


def run_one(lib, path, steps, lds, changes, progress=None, record=None):
    """returns (status, detail).  status in {'ok', 'deviates', 'warning', 'rejected', 'noload', 'skip'}
    record: a dict that receives the inputs and the reference trajectory (fixtures for tests/test_gpu_sweep.py)"""
    try:
        m = load_model(path)
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
    if lds:
        b.plan_lds(lds)          # (the emulation: the device's default plan; on the device the batch has planned already)
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
    if record is not None:
        record.update(state0=s0, ref_state=ref, ref_counts=ref_int, ref_sensordata=ref_sens, ref_warnings=np.array(ref_warn),
                      ctrl0=ctrl0 if ctrl0 is not None else np.zeros(0),
                      mocap_pos=mocap[0].ravel() if mocap else np.zeros(0), mocap_quat=mocap[1].ravel() if mocap else np.zeros(0))
        # the same trajectory from the reference linked with the KERNELS' sin / cos / atan2 / exp (liboracle_dm.so,
        # oracle/devmath_shim.cc): what the device is expected to reproduce to the bit -- the emulation calls the host's
        # libm, as the reference as built does, the device evaluates its own routines
        if rb.available("devmath"):
            prev = rb.use("devmath")
            try:
                # (the SAME compiled model: saved by the build that compiled it and re-loaded here -- compiling the XML
                #  again under this build would round the model's own constants, e.g. quaternions of euler angles, with
                #  the other libm, and a pile of boxes amplifies that last bit to centimetres in 15 steps: planks.xml)
                import tempfile
                with tempfile.NamedTemporaryFile(suffix=".mjb") as tmpf:
                    prev2 = rb.use(prev)
                    m.save_binary(tmpf.name)
                    rb.use(prev2)
                    m2 = rb.MjModel.from_binary_path(tmpf.name)
                for k, v in changes.items():
                    setattr(m2.opt, k, v)
                d2 = rb.MjData(m2)
                if m2.nkey > 0:
                    rb.mj_resetDataKeyframe(m2, d2, 0)
                else:
                    rb.mj_resetData(m2, d2)
                r2 = np.zeros_like(ref); i2 = np.zeros_like(ref_int); s2 = np.zeros_like(ref_sens)
                for t in range(steps):
                    rb.mj_step(m2, d2)
                    r2[t] = rb.mj_getState(m2, d2, spec)
                    i2[t] = (d2.ncon, d2.nefc)
                    if m2.nsensordata:
                        s2[t] = d2.sensordata
                record.update(ref_state_dm=r2, ref_counts_dm=i2, ref_sensordata_dm=s2)
                del d2, m2
            finally:
                rb.use(prev)
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


def run_isolated(lib, path, steps, lds, variations, want_record=False, stop_after_first=False):
    """run_one for each (name, changes) of `variations` in a forked child: a crash of the compiled reference
    (sphere_radial.xml forced to PGS overflows its 10 MB arena and the reference's own mj_step then dies) or of the
    kernels / their emulation is a status of that model and variation, not the end of the sweep.  One child runs all the
    variations of a model (on the device: one initialisation of the HIP runtime per model -- the parent never touches it);
    after a crash the remaining variations run in a new child.  Returns [(name, status, detail, record or None)]; the
    variations after a rejected / unloadable / skipped first one are not run."""
    import pickle
    import struct
    results = []
    todo = list(variations)
    while todo:
        r, w = os.pipe()
        pid = os.fork()
        if pid == 0:
            os.close(r)
            for name, changes in todo:
                os.write(w, b"S")
                try:
                    rec = {} if want_record else None
                    st, detail = run_one(lib, path, steps, lds, changes, progress=lambda: os.write(w, b"R"), record=rec)
                except Exception as ex:                            # a failure of the harness itself: reported, not hidden
                    st, detail, rec = "error", f"{type(ex).__name__}: {ex}", None
                blob = pickle.dumps((name, st, detail, rec))
                os.write(w, b"D" + struct.pack("<Q", len(blob)) + blob)
                if (st in ("rejected", "noload", "skip") and name == variations[0][0]) or stop_after_first:
                    break
            os._exit(0)
        os.close(w)
        with os.fdopen(r, "rb") as fh:
            data = fh.read()
        _, status = os.waitpid(pid, 0)
        pos, started, in_mjhip = 0, False, False
        while pos < len(data):
            tag = data[pos:pos + 1]
            if tag == b"S":
                started, in_mjhip, pos = True, False, pos + 1
            elif tag == b"R":
                in_mjhip, pos = True, pos + 1
            else:
                n = struct.unpack("<Q", data[pos + 1:pos + 9])[0]
                results.append(pickle.loads(data[pos + 9:pos + 9 + n]))
                todo.pop(0)
                started, pos = False, pos + 9 + n
        if started:                                               # the child died inside this variation
            sig = status & 0x7f
            who = "the kernels (or their emulation)" if in_mjhip else "the compiled REFERENCE (before mjhip ran a step)"
            results.append((todo.pop(0)[0], "crash", f"the process died (signal {sig}) inside {who}", None))
        elif results and ((results[-1][1] in ("rejected", "noload", "skip") and results[-1][0] == variations[0][0]) or stop_after_first):
            break
        elif not started and todo and os.WIFSIGNALED(status):
            results.append((todo.pop(0)[0], "crash", f"the process died (signal {status & 0x7f}) between variations", None))
    return results


def export_name(short):
    """file stem of an exported model: the path under the reference tree, flattened"""
    return re.sub(r"[^A-Za-z0-9_.-]", "_", short[:-4].replace("/", "__"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=15)
    ap.add_argument("--nvmax", type=int, default=80)
    ap.add_argument("--lds", type=int, default=40960)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_sweep"))
    ap.add_argument("--only", default="")
    ap.add_argument("--variations", default="all")
    ap.add_argument("--export", default="", help="write every accepted model as <dir>/<name>.mjb + index.txt (build container)")
    ap.add_argument("--from-mjb", default="", help="sweep the exported .mjb files of this directory instead of the reference tree")
    ap.add_argument("--device", action="store_true", help="step with libmjhip.so on the GPU instead of the host emulation")
    ap.add_argument("--fixtures", default="", help="with --subset: store <name>.mjb + <name>.npz (inputs + reference trajectories of every variation) here")
    ap.add_argument("--oracle", choices=["parity", "devmath"], default="",
                    help="the reference build to compare with: parity = as built (glibc libm), devmath = linked with the kernels' "
                         "sin / cos / atan2 / exp; default: parity on the emulation (which calls the host's libm), devmath on the device")
    ap.add_argument("--subset", default="", help="file with one model path (as in sweep.txt) per line")
    args = ap.parse_args()
    if not args.from_mjb and not os.path.isdir(os.path.join(REF, "model")):
        sys.exit(f"reference tree not found at {REF} (this tool runs in the build container)")
    lib = K.Lib(os.path.join(ROOT, 'mujoco_amd', 'csrc', 'libmjhip.so')) if args.device else K.Lib(HOSTSIM_LIB)
    if args.device:
        args.lds = 0
    oracle = args.oracle or ("devmath" if args.device else "parity")
    rb.use(oracle)
    if args.from_mjb:
        index = [ln.split() for ln in open(os.path.join(args.from_mjb, "index.txt")) if ln.strip() and not ln.startswith("#")]
        files = [os.path.join(args.from_mjb, stem + ".mjb.gz") for stem, _ in index if not args.only or args.only in _]
        shorts = {os.path.join(args.from_mjb, stem + ".mjb.gz"): short for stem, short in index}
    else:
        files = model_files(args.only)
        shorts = {f: os.path.relpath(f, REF) for f in files}
    subset = set(ln.strip() for ln in open(args.subset) if ln.strip() and not ln.startswith("#")) if args.subset else None
    if subset is not None:
        files = [f for f in files if shorts[f] in subset]
    exported = []
    lines = []
    census = collections.Counter()
    status_count = collections.Counter()
    t0 = time.time()
    for f in files:
        short = shorts[f]
        try:
            m = load_model(f)
            nv = m.nv
            if (args.export or args.fixtures) and 0 < nv <= args.nvmax:
                # (saved before anything runs; kept only if the model is accepted)
                dst = args.export or args.fixtures
                os.makedirs(dst, exist_ok=True)
                save_gz(m, os.path.join(dst, export_name(short)))
            del m
        except Exception as ex:
            status_count["noload"] += 1
            lines.append(f"noload     {short}: {str(ex).splitlines()[0][:90] if str(ex) else ''}")
            continue
        if nv == 0 or nv > args.nvmax:
            status_count["skip"] += 1
            lines.append(f"skip       {short}: nv = {nv}")
            continue
        todo = [(n, c) for n, c in VARIATIONS if args.variations == "all" or n in args.variations.split(",")]
        res = run_isolated(lib, f, args.steps, args.lds, todo, want_record=bool(args.fixtures),
                           stop_after_first=bool(args.export and not args.fixtures))   # (exporting: the as-shipped run decides)
        for name, st, detail, rec in res:
            if args.fixtures and st == "ok" and rec:
                fx = os.path.join(args.fixtures, export_name(short) + ".npz")
                old = dict(np.load(fx)) if os.path.exists(fx) and name != todo[0][0] else {}
                for k, v in rec.items():
                    old[f"{name}:{k}" if k.startswith("ref_") else k] = v
                old[f"{name}:changes"] = np.array([f"{k}={v}" for k, v in dict(todo)[name].items()], dtype="U32")
                np.savez_compressed(fx, **old)
            if name == todo[0][0]:
                status_count[st] += 1
                if st == "rejected":
                    reason = re.sub(r"\s*\(.*", "", detail.split(":")[-1].strip())[:70]
                    census[reason] += 1
                if args.export or args.fixtures:
                    if st in ("ok", "deviates", "warning"):
                        exported.append((export_name(short), short))
                    else:
                        try: os.remove(os.path.join(args.export or args.fixtures, export_name(short) + ".mjb.gz"))
                        except OSError: pass
            else:
                status_count[f"{name}:{st}"] += 1
            lines.append(f"{st:10s} {short} [{name}]: {detail}")
            print(lines[-1], flush=True)
    out = []
    if args.export or args.fixtures:
        with open(os.path.join(args.export or args.fixtures, "index.txt"), "w") as fh:
            fh.write("# <file stem> <model path under the reference tree>: models the emulation accepted, saved by mj_saveModel\n")
            fh.write("".join(f"{a} {b}\n" for a, b in exported))
    where = (f"{len(files)} exported .mjb files of {args.from_mjb}" if args.from_mjb else
             f"{len(files)} files under {REF}/model and test/**/testdata")
    how = (f"libmjhip.so on {lib.backend()}" if args.device else f"hostsim with a {args.lds} B LDS plan")
    out.append(f"# model sweep: {where}, {args.steps} steps, nv <= {args.nvmax}, {how}; oracle build: {oracle}; {time.time() - t0:.0f} s")
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
