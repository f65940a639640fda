# dealing policies A/B (MJHIP_BALANCE_SNAKE: 0 descending, 1 serpentine, 2 heaviest + three lightest)
export TMPDIR=/tmp
for rep in 1 2; do
for s in 1 2; do
MJHIP_BALANCE_SNAKE=$s python bench.py --no-extra --steps 500 --warmup 100 > /tmp/b.json 2> /tmp/err
python -c "import json;d=json.load(open('/tmp/b.json'));print('mode=$s 500/100 %.3fM launch %.2f ms' % (d['value']/1e6, d['roofline']['launch_ms']))"
MJHIP_BALANCE_SNAKE=$s python bench.py --no-extra --steps 20 --warmup 5 > /tmp/b.json 2> /tmp/err
python -c "import json;d=json.load(open('/tmp/b.json'));print('mode=$s 20/5 %.3fM launch %.2f ms' % (d['value']/1e6, d['roofline']['launch_ms']))"
done; done
MJHIP_BALANCE_SNAKE=1 python tools/regime_stats.py 1000 2>&1 | grep "^launch"
MJHIP_BALANCE_SNAKE=2 python tools/regime_stats.py 1000 2>&1 | grep "^launch"
