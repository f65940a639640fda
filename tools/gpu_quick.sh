# quick A/B of the committed library: two repetitions of the default and of the driver's configuration
export TMPDIR=/tmp
for rep in 1 2; do
python bench.py --no-extra --steps 500 --warmup 100 > /tmp/b.json 2> /tmp/err
python -c "import json;d=json.load(open('/tmp/b.json'));print('${TAG:-lib} 500/100 %.3fM launch %.2f ms' % (d['value']/1e6, d['roofline']['launch_ms']))"
python bench.py --no-extra --steps 20 --warmup 5 > /tmp/b.json 2> /tmp/err
python -c "import json;d=json.load(open('/tmp/b.json'));print('${TAG:-lib} 20/5 %.3fM launch %.2f ms' % (d['value']/1e6, d['roofline']['launch_ms']))"
done
