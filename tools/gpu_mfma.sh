# A/B of the matrix-core AR build against the exact vector path, testspeed regime (nefc ~ 44)
set -x
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/mfma
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mfma or step1" 2>&1 | tail -5
ARGS="--no-extra --ctrl ou-halton --settle 1000 --steps 200 --warmup 20"
for on in 0 1; do
  MJHIP_MFMA=$on python bench.py $ARGS > $OUT/bench_mfma$on.json 2> $OUT/err
  python -c "import json;d=json.load(open('$OUT/bench_mfma$on.json'));print('mfma=$on testspeed regime %.3fM launch %.2f ms' % (d['value']/1e6, d['roofline']['launch_ms']), d['end_state'])"
  MJHIP_MFMA=$on python bench.py --no-extra --steps 500 --warmup 100 > $OUT/bench_rand_mfma$on.json 2> $OUT/err
  python -c "import json;d=json.load(open('$OUT/bench_rand_mfma$on.json'));print('mfma=$on random actions %.3fM launch %.2f ms' % (d['value']/1e6, d['roofline']['launch_ms']), d['end_state'])"
done
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_BUSY_CYCLES\|SQ_INSTS_VALU\b\|SQ_VALU_MFMA[A-Z_]*" | sort -u > $OUT/avail_mfma_counters.txt; cat $OUT/avail_mfma_counters.txt
for c in SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU; do
  MJHIP_MFMA=1 timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- python $OLDPWD/bench.py $ARGS > /dev/null 2> $OUT/pmc_$c.err
  python - <<PY
import csv,glob
v=[float(r["Counter_Value"]) for f in glob.glob("$OUT/pmc_$c/**/*counter_collection.csv",recursive=True) for r in csv.DictReader(open(f)) if "rollout" in r.get("Kernel_Name","") and r.get("Counter_Name")=="$c"]
print("$c", "dispatches", len(v), "last two (timed 100-step launches):", v[-2:])
PY
done
rm -rf $OUT/pmc_*/*/*.db
