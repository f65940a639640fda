# A/B: PGS with the chain length as a compile-time constant; tests; testspeed regime
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
bash tools/gpu_quick.sh
python bench.py --steps 100 --warmup 20 --regime-steps 200 > gpurun_out/r2f_bench.json 2> /tmp/err; tail -3 /tmp/err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2f_bench.json'))
print('100/20 %.3fM' % (d['value']/1e6), 'parity', d.get('parity_sample'))
print('regime', d.get('testspeed_regime'))
PY
