# 1 vs 2 environments per wavefront at batch sizes beyond the machine's 4096 one-wave slots
set -x
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/bigbatch
mkdir -p $OUT
for n in 4096 8192 16384 32768; do
  for v in lean generic; do
    MJHIP_VARIANT=$v timeout 300 python bench.py --no-extra --envs-per-gpu $n --steps 200 --warmup 100 > $OUT/b_${v}_$n.json 2> $OUT/err || tail -2 $OUT/err
    python -c "import json;d=json.load(open('$OUT/b_${v}_$n.json'));print('$v nenv=$n %.3fM env-steps/s launch %.2f ms' % (d['value']/1e6, d['roofline']['launch_ms']))"
  done
done
