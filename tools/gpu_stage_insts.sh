export TMPDIR=/tmp
OUT=$PWD/gpurun_out/stage_insts
mkdir -p $OUT
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc -o pmc -- python $OLDPWD/tools/stage_insts.py > $OUT/run.log 2>&1
cd $OLDPWD
python - <<'PY'
import csv, glob, collections
rows = collections.OrderedDict()
for f in glob.glob('gpurun_out/stage_insts/pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'mjh_k_forward' in r['Kernel_Name']:
            rows.setdefault(int(r['Dispatch_Id']), {})[r['Counter_Name']] = float(r['Counter_Value'])
order = [l.split(' ', 1)[1].strip().split(',') for l in open('gpurun_out/stage_insts/run.log') if l.startswith('STAGE_ORDER')][0]
ids = sorted(rows)[-len(order):]
print("%-18s %10s %10s %10s %10s %10s   (per env, wave-level instructions)" % ("stage", "VALU", "SALU", "FLAT", "LDS", "SMEM"))
for name, i in zip(order, ids):
    r = rows[i]
    print("%-18s %10.0f %10.0f %10.0f %10.0f %10.0f" % (name, *[r.get(k, 0)/4096 for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_FLAT", "SQ_INSTS_LDS", "SQ_INSTS_SMEM")]))
PY
tail -2 $OUT/run.log
rm -rf $OUT/pmc/*/*.db
