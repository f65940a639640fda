set -x
export TMPDIR=/tmp
for sh in -1 3 2; do
  for rep in 1 2; do
  MJHIP_LIB=$PWD/tools/variants/libmjhip_prio$sh.so python bench.py --no-extra --steps 500 --warmup 100 > /tmp/b.json 2> /tmp/err
  python -c "import json;d=json.load(open('/tmp/b.json'));print('prio_shift=$sh 500/100 %.3fM launch %.2f ms' % (d['value']/1e6, d['roofline']['launch_ms']))"
  MJHIP_LIB=$PWD/tools/variants/libmjhip_prio$sh.so python bench.py --no-extra --steps 20 --warmup 5 > /tmp/b.json 2> /tmp/err
  python -c "import json;d=json.load(open('/tmp/b.json'));print('prio_shift=$sh 20/5 %.3fM launch %.2f ms' % (d['value']/1e6, d['roofline']['launch_ms']))"
  done
done
