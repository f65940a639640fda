set -x
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r2d
mkdir -p $OUT
python -m pytest tests -m gpu -x -q 2>&1 | tail -6
for bp in 1 0; do
  MJHIP_BROADPHASE=$bp python bench.py --no-extra --steps 500 --warmup 100 > $OUT/b500_bp$bp.json 2> $OUT/err
  python -c "import json;d=json.load(open('$OUT/b500_bp$bp.json'));print('broadphase=$bp 500/100 %.3fM launch %.2f ms' % (d['value']/1e6, d['roofline']['launch_ms']))"
  MJHIP_BROADPHASE=$bp python bench.py --no-extra --steps 20 --warmup 5 > $OUT/b20_bp$bp.json 2> $OUT/err
  python -c "import json;d=json.load(open('$OUT/b20_bp$bp.json'));print('broadphase=$bp 20/5 %.3fM launch %.2f ms' % (d['value']/1e6, d['roofline']['launch_ms']))"
done
