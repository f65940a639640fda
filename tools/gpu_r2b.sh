set -x
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r2b
mkdir -p $OUT
for v in generic lean; do
  MJHIP_VARIANT=$v timeout 300 python bench.py --no-extra --steps 500 --warmup 100 > $OUT/bench500_$v.json 2> $OUT/err
  python -c "import json;d=json.load(open('$OUT/bench500_$v.json'));print('$v 500/100', d['value'], d['roofline']['launch_ms'], d['end_state'])"
  MJHIP_VARIANT=$v timeout 300 python bench.py --no-extra --steps 20 --warmup 5 > $OUT/bench20_$v.json 2> $OUT/err
  python -c "import json;d=json.load(open('$OUT/bench20_$v.json'));print('$v 20/5', d['value'], d['roofline']['launch_ms'], d['end_state'])"
done
