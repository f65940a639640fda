# usage: bash tools/gpu_sweep.sh   (on the GPU box): correctness first, then waves/SIMD x LDS budget sweep
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
tail -4 gpurun_out/smoke.log
timeout 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
: > gpurun_out/sweep.txt
for w in 2 3 4; do
  for lds in 0 10240 13312 16384 20480 26624 32768; do
    echo "== waves_per_eu=$w lds=$lds" >> gpurun_out/sweep.txt
    MJHIP_LIB=$PWD/tools/variants/gpurun_out_libs_w$w.so MJHIP_LDS_BYTES=$lds timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
try:
    r = json.loads(sys.stdin.readline()); print(r['value'], r['ms_per_step'], r['end_state'])
except Exception as ex: print('FAILED', ex)
" >> gpurun_out/sweep.txt
  done
done
cat gpurun_out/sweep.txt
