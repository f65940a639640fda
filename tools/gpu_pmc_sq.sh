set -x
mkdir -p gpurun_out/pmc_sq
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_sq
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -o -E "\b(SQ|SQC)_[A-Z0-9_]+" $OUT/counters.txt | sort -u > $OUT/sq_names.txt
wc -l $OUT/sq_names.txt
cd /tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_IFETCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" \
           "SQ_INSTS_FLAT_LDS_ONLY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o pmc -- python $OLDPWD/bench.py --no-cpu-baseline --steps 100 --warmup 100 > $OUT/p$i.log 2>&1
  tail -2 $OUT/p$i.log
done
cd $OLDPWD
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc_sq/p*/')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'rollout' in r.get('Kernel_Name', ''):
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in acc.items():
            print(d, k, 'n=%d mean=%.4g' % (len(v), sum(v)/len(v)))
PY
rm -rf $OUT/p*/*/*.db
du -sh $OUT
