#!/bin/bash
# Instruction-fetch evidence for low-occupancy launches (what are the ~127 K cycles of a slider-crank step?).
#   bash tools/gpu_icache.sh <tag>
# (a) slider_crank at 64 / 256 / 1024 / 4096 environments, generic and lean kernels: env-steps/s and us per env-step
# (b) rocprofv3 --pmc passes on the 64-environment rollout: instruction-cache requests / hits / misses, fetch count and level
TAG=${1:-r05}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/icache_$TAG
mkdir -p $OUT
B="python $PWD/bench.py --config slider_crank --no-legs --no-extra --no-cpu-baseline --steps 1000 --warmup 100"
val() { python - "$1" <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    n = j["config"]["envs_per_gpu"]
    print("  %6d envs  %-10s %8.3f M env-steps/s   %7.2f us per env-step of one wavefront   parity %s" % (n, j["roofline"].get("kernel", "?").replace("mjh_k_rollout_", ""), j["value"]/1e6, n/j["value"]*1e6, j.get("parity_sample", {}).get("ok")))
except Exception as exc:
    print("  (no line:", exc, ")")
PY
}
{
echo "== slider_crank.xml: rate against launch size and kernel (one wavefront = one environment = one workgroup)"
for n in 64 256 1024 4096; do
  for v in "" lean; do
    MJHIP_VARIANT=$v timeout 300 $B --envs-per-gpu $n > $OUT/b_${n}_${v:-default}.json 2> $OUT/b_${n}_${v:-default}.err; val $OUT/b_${n}_${v:-default}.json
  done
done
} > $OUT/icache_summary.txt 2>&1
cd /tmp
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"; do
  i=$((i+1))
  for v in "" lean; do
    MJHIP_VARIANT=$v timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_${v:-default}_p$i -o pmc -- $B > $OUT/pmc_${v:-default}_p$i.log 2> $OUT/pmc_${v:-default}_p$i.err
  done
done
cd $OLDPWD
python - $OUT >> $OUT/icache_summary.txt 2>&1 <<'PY'
import collections, csv, glob, json, os, sys
out = sys.argv[1]
for v in ("default", "lean"):
    print("== counters of the rollout kernel, 64 environments, %s variant: per env-step" % v)
    for lg in sorted(glob.glob(os.path.join(out, "pmc_%s_p*.log" % v))):
        try:
            bj = json.loads([l for l in open(lg) if l.startswith("{")][-1])
        except Exception as exc:
            print("  ", os.path.basename(lg), "no bench line", exc); continue
        spl, nenv = int(bj["roofline"]["steps_per_launch"]), int(bj["config"]["envs_per_gpu"])
        ntimed = -(-int(bj["steps"])//spl)
        acc = collections.defaultdict(list)
        for f in glob.glob(os.path.join(lg[:-4], "**", "*counter_collection.csv"), recursive=True):
            rs = [r for r in csv.DictReader(open(f)) if "rollout" in r.get("Kernel_Name", "")]
            rs.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
            for r in rs: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, vals in sorted(acc.items()):
            vals = vals[-ntimed:]
            print("   %-30s %14.1f" % (c, sum(vals)/len(vals)/(spl*nenv)))
PY
cat $OUT/icache_summary.txt
rm -rf $OUT/*/*/*.db 2>/dev/null
find $OUT -name "*.csv" -size +2M -delete
