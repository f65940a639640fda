#!/usr/bin/env python3
"""Merge fixture directories written by `tools/model_sweep.py --fixtures <dir>` into tests/golden/sweep and fold their
sweep.txt lines into profiles/<round>_sweep/sweep.txt (re-run models replace their earlier lines; the header counts are
recomputed).  Build-container tool (test infrastructure).

  python tools/merge_sweep_fixtures.py <sweep.txt to update> <fixture dir>:<its sweep.txt> ...
"""
import collections
import glob
import os
import re
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DST = os.path.join(ROOT, "tests", "golden", "sweep")


def short(line):
    m = re.match(r"^\S+\s+(\S+)", line)
    return m.group(1).rstrip(":") if m else None


def main():
    target, pairs = sys.argv[1], [a.split(":") for a in sys.argv[2:]]
    entries = dict(l.split() for l in open(os.path.join(DST, "index.txt")) if l.strip() and not l.startswith("#"))
    repl = {}
    for src, txt in pairs:
        for l in open(os.path.join(src, "index.txt")):
            if l.strip() and not l.startswith("#"):
                s, sh = l.split()
                if os.path.exists(os.path.join(src, s + ".npz")):
                    shutil.copy(os.path.join(src, s + ".mjb.gz"), DST)
                    shutil.copy(os.path.join(src, s + ".npz"), DST)
                    entries[s] = sh
        for l in open(txt).read().splitlines():
            if not l.startswith("#"):
                repl.setdefault(short(l), []).append(l)
    # a device-libm trajectory is stored only where it differs from the reference as built
    for f in glob.glob(os.path.join(DST, "*.npz")):
        d = dict(np.load(f))
        drop = [k for k in d if k.endswith("_dm") and k[:-3] in d and d[k].shape == d[k[:-3]].shape and np.array_equal(d[k], d[k[:-3]])]
        if drop:
            for k in drop:
                del d[k]
            np.savez_compressed(f, **d)
    with open(os.path.join(DST, "index.txt"), "w") as f:
        f.write("# <file stem> <model path under the reference tree>: models the emulation accepted, saved by mj_saveModel\n")
        f.write("# (nv <= 320: all five variations; model/flex/*.xml with 320 < nv <= 1600: as shipped only)\n")
        for a, b in sorted(entries.items(), key=lambda x: x[1]):
            f.write(f"{a} {b}\n")
    lines = open(target).read().splitlines()
    first = next(l for l in lines if l.startswith("# model sweep"))
    out, done = [], set()
    for l in lines:
        if l.startswith("#"):
            continue
        sh = short(l)
        if sh in repl:
            if sh not in done:
                out += repl[sh]
                done.add(sh)
            continue
        out.append(l)
    for sh, ls in repl.items():
        if sh not in done:
            out += ls
    out.sort(key=lambda l: (short(l) or ""))
    st, var, census = collections.Counter(), collections.Counter(), collections.Counter()
    for l in out:
        status = l.split()[0]
        m = re.search(r"\[(.*?)\]", l)
        if m is None:
            st[status] += 1
        elif m.group(1) == "as-shipped":
            st[status] += 1
            if status == "rejected":
                census[re.sub(r"\s*\(.*", "", l.split(":")[-1].strip())[:70]] += 1
        else:
            var[f"{m.group(1)}:{status}"] += 1
    hdr = [first, "# as-shipped status counts: " + ", ".join(f"{k} {v}" for k, v in sorted(st.items())),
           "# variations: " + ", ".join(f"{k} {v}" for k, v in sorted(var.items())), "# rejection census (as shipped):"] + \
          [f"#   {v:3d}  {k}" for k, v in census.most_common()]
    open(target, "w").write("\n".join(hdr + out) + "\n")
    print("\n".join(hdr[:3]), f"\n{len(entries)} fixtures")


if __name__ == "__main__":
    main()
