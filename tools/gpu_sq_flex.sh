#!/bin/bash
# one SQ counter pass of the flex bench (multi-wavefront kernel): instruction mix per env-step -> gpurun_out/sq_flex/summary.txt
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/sq_flex
mkdir -p $OUT
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/p1 -o pmc -- python $OLDPWD/bench.py --config flex --no-extra --steps 200 > $OUT/p1.log 2> $OUT/p1.err
cd $OLDPWD
python - <<'PY' > $OUT/summary.txt
import csv, glob, collections
f = glob.glob("gpurun_out/sq_flex/p1/**/*counter_collection.csv", recursive=True)
tot = collections.defaultdict(float); nd = 0
for path in f:
    for r in csv.DictReader(open(path)):
        if "rollout_wn" not in r.get("Kernel_Name", ""): continue
        tot[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
disp = sorted({d for d, _ in tot}, key=int)
last = disp[-1] if disp else None      # the timed launch: 200 steps x 256 environments
print("flex bench, kernel mjh_k_rollout_wn, last dispatch (200 steps x 256 envs, 8 wavefronts per environment); per env-step:")
for (d, c), v in sorted(tot.items()):
    if d == last: print("  %-22s %14.0f   per env-step %10.1f" % (c, v, v/(200*256)))
PY
cat $OUT/summary.txt
rm -rf $OUT/p1/*/*.db 2>/dev/null; find $OUT -name "*.csv" -size +2M -delete
