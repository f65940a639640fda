"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes for the rollout kernel.

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  Correction for gfx950 per
/opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE under-reports a wide coalesced read
stream by exactly 2x (128-B requests tallied at 64 B); other access widths and WRITE_SIZE are
uncalibrated.  Both the raw and the x2-corrected read figure are printed.  Only the launches of the
TIMED region (the last ceil(K/C) rollout dispatches of a `bench.py --no-extra` run) are averaged.
usage: pmc_summary.py <dir with bench.json, pmc_fetch/, pmc_write/>
"""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
_b = json.loads(open(os.path.join(out, "bench.json")).read().strip().splitlines()[-1])
STEPS_PER_LAUNCH = int(_b["roofline"]["steps_per_launch"]); NENV = int(_b["config"]["envs_per_gpu"])
NTIMED = -(-int(_b["steps"]) // STEPS_PER_LAUNCH)


def load(sub, counter):
    vals = []
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "rollout" in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter]
        rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
        vals += [float(r["Counter_Value"]) for r in rows]
    return vals[-NTIMED:]


fetch = load("pmc_fetch", "FETCH_SIZE")
write = load("pmc_write", "WRITE_SIZE")
print(f"workload: {_b['config']['workload']}; kernel variant {_b['config'].get('kernel_variant')}; steps {_b['steps']} warmup {_b['warmup']}")
for name, v in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
    if v:
        print(f"{name}: {len(v)} timed dispatches, mean {sum(v)/len(v):.1f} KiB/dispatch, min {min(v):.1f}, max {max(v):.1f}")
    else:
        print(f"{name}: no samples")
if fetch and write:
    f, w = sum(fetch) / len(fetch) * 1024, sum(write) / len(write) * 1024
    print(f"per launch ({STEPS_PER_LAUNCH} steps x {NENV} envs): read {f/1e6:.2f} MB raw / {2*f/1e6:.2f} MB with the gfx950 x2 correction, "
          f"written {w/1e6:.2f} MB; per env-step: {(2*f+w)/(STEPS_PER_LAUNCH*NENV):.0f} B (corrected)")
