#!/usr/bin/env python3
"""Register / scratch / spill report of the shipped code objects: summarises the -Rpass-analysis=kernel-resource-usage
remarks hipcc left in mujoco_amd/csrc/build/<unit>.log (written by __graft_entry__.build()).  Runs anywhere.

  python tools/kernel_resources.py [substring ...]     default: the kernels and the out-of-line collision routines
"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
want = sys.argv[1:] or ["mjh_k_", "ccd_", "rc_farthest", "stage_collision", "solve_pgs", "factor_ld", "solve_ld"]
rows = []
for log in sorted(glob.glob(os.path.join(ROOT, "mujoco_amd", "csrc", "build", "*.log"))):
    cur = None
    for line in open(log, errors="replace"):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"unit": os.path.basename(log)[:-4], "name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark: .*?\s{2,}(\w[\w /\[\]]*?): (\S+)", line) or re.search(r"^\s+(\w[\w /\[\]]*?): (\S+)\s*$", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
names = [r["name"] for r in rows]
try:
    dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
except Exception:
    dem = names
print("%-14s %-64s %5s %5s %8s %6s %6s %4s" % ("unit", "function", "VGPR", "SGPR", "scratch", "vspill", "sspill", "occ"))
for r, d in zip(rows, dem):
    if not any(w in d for w in want):
        continue
    short = re.sub(r"\(.*", "", d)[-64:]
    print("%-14s %-64s %5s %5s %8s %6s %6s %4s" % (r["unit"], short, r.get("VGPRs", "?"), r.get("TotalSGPRs", "?"),
          r.get("ScratchSize [bytes/lane]", "?"), r.get("VGPRs Spill", "?"), r.get("SGPRs Spill", "?"), r.get("Occupancy [waves/SIMD]", "?")))
