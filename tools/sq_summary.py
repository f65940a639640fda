"""Summarise the SQ counter passes of tools/gpu_sq.sh.
usage: sq_summary.py <dir>    (dir holds stages_<regime>_p<k>/ and rollout_<regime>_p<k>/ rocprofv3 outputs + their .log)
Per-stage figures come from single-stage dispatches of the forward kernel (tools/sq_stages.py) and are
per env-step; rollout figures are per env-step over the timed launches of bench.py --no-extra."""
import collections, csv, glob, json, os, sys

out = sys.argv[1]


def rows(d, kernel_substr):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rs = [r for r in csv.DictReader(open(f)) if kernel_substr in r.get("Kernel_Name", "")]
        rs.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
        for r in rs:
            acc[r["Counter_Name"]].append((int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"])))
    return acc


for regime in ("uniform", "testspeed"):
    logs = sorted(glob.glob(os.path.join(out, f"stages_{regime}_p*.log")))
    if not logs:
        continue
    meta = None
    table = collections.OrderedDict()
    for lg in logs:
        try:
            meta = json.loads(open(lg).read().strip().splitlines()[-1])
        except Exception:
            continue
        acc = rows(lg[:-4], "forward")
        n, reps, nenv = len(meta["stages"]), meta["reps"], meta["nenv"]
        for cname, vals in acc.items():
            vals = [v for _, v in vals][-n*reps:]
            if len(vals) != n*reps:
                continue
            per = [sum(vals[i*reps:(i + 1)*reps])/reps/nenv for i in range(n)]
            for i, st in enumerate(meta["stages"]):
                if meta.get("prefix") and st != "all":
                    # dispatches are prefixes of the stage list: a stage is the difference of two consecutive prefixes
                    table.setdefault(st, {})[cname] = per[i] - (per[i - 1] if i else 0.0)
                    if i == n - 2: table.setdefault("sum (last prefix)", {})[cname] = per[i]
                else:
                    table.setdefault(st, {})[cname] = per[i]
    if meta:
        print(f"== per stage, regime {regime}: {meta['variant']} kernel, {meta['nenv']} envs, mean ncon {meta['mean_ncon']:.1f} nefc {meta['mean_nefc']:.1f} "
              f"solver iterations {meta['mean_niter']:.1f}; counters per env-step (one wavefront)")
        names = sorted({c for t in table.values() for c in t})
        print("stage".ljust(14) + "".join(c.replace("SQ_", "")[:15].rjust(16) for c in names))
        for st, t in table.items():
            print(st.ljust(14) + "".join(("%.0f" % t.get(c, float("nan"))).rjust(16) for c in names))
        # one table for the "issuing" question: SQ_WAVE_CYCLES, SQ_WAIT_INST_* and SQ_ACTIVE_INST_* all count QUAD-cycles
        # (MI355X_MICROARCH.md, "s_memtime tick vs SQ PMC units"), so their ratios need no conversion; SQ_BUSY_CYCLES counts
        # cycles of the SQ (per XCD-level unit), reported as is
        print("stage".ljust(22) + "VALU insts".rjust(12) + "wave qcyc".rjust(12) + "issuing any".rjust(12) + "issuing VALU".rjust(13) + "waiting".rjust(10) + "lanes/VALU".rjust(12))
        for st, t in table.items():
            wc = t.get("SQ_WAVE_CYCLES")
            if not wc:
                continue
            lanes = t["SQ_THREAD_CYCLES_VALU"]/t["SQ_ACTIVE_INST_VALU"] if t.get("SQ_THREAD_CYCLES_VALU") and t.get("SQ_ACTIVE_INST_VALU") else float("nan")
            print(st.ljust(22) + ("%.0f" % t.get("SQ_INSTS_VALU", float("nan"))).rjust(12) + ("%.0f" % wc).rjust(12)
                  + ("%.3f" % (t.get("SQ_ACTIVE_INST_ANY", float("nan"))/wc)).rjust(12) + ("%.3f" % (t.get("SQ_ACTIVE_INST_VALU", float("nan"))/wc)).rjust(13)
                  + ("%.3f" % (t.get("SQ_WAIT_INST_ANY", float("nan"))/wc)).rjust(10) + ("%.1f" % lanes).rjust(12))
    # whole rollout kernel
    for lg in sorted(glob.glob(os.path.join(out, f"rollout_{regime}_p*.log"))):
        try:
            bj = json.loads(open(lg).read().strip().splitlines()[-1])
        except Exception:
            continue
        acc = rows(lg[:-4], "rollout")
        spl, nenv = int(bj["roofline"]["steps_per_launch"]), int(bj["config"]["envs_per_gpu"])
        ntimed = -(-int(bj["steps"])//spl)
        print(f"-- rollout kernel, regime {regime} ({os.path.basename(lg)}): per env-step over the {ntimed} timed launches "
              f"({bj['value']/1e6:.2f} M env-steps/s under the profiler)")
        for cname, vals in sorted(acc.items()):
            v = [x for _, x in vals][-ntimed:]
            print(f"   {cname:28s} {sum(v)/len(v)/(spl*nenv):14.1f}")
