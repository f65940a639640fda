set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
for lay in soa aos; do
MJHIP_LAYOUT=$lay timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$lay.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$lay.log
tail -6 gpurun_out/pytest_gpu_$lay.log
done
: > gpurun_out/sweep_soa.txt
for epw in 64 32 16 8; do
  for lds in 0 16384 20480 32768; do
    echo "== soa epw=$epw lds=$lds" >> gpurun_out/sweep_soa.txt
    MJHIP_LAYOUT=soa MJHIP_EPW=$epw MJHIP_LDS_BYTES=$lds timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
try:
    r = json.loads(sys.stdin.readline()); print(r['value'], r['ms_per_step'], r['end_state'])
except Exception as ex: print('FAILED', ex)
" >> gpurun_out/sweep_soa.txt
  done
done
cat gpurun_out/sweep_soa.txt
