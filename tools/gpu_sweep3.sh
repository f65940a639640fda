mkdir -p gpurun_out
: > gpurun_out/sweep3.txt
for cfg in "2 20480" "2 16384" "3 13312" "3 12288" "4 10240" "4 9216"; do
  set -- $cfg
  echo "== waves_per_eu=$1 lds=$2" >> gpurun_out/sweep3.txt
  MJHIP_LIB=$PWD/tools/variants/gpurun_out_libs_w$1.so MJHIP_LDS_BYTES=$2 timeout 300 python bench.py --steps 200 --warmup 50 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
try:
    r = json.loads(sys.stdin.readline()); print(r['value'], r['ms_per_step'])
except Exception as ex: print('FAILED', ex)
" >> gpurun_out/sweep3.txt
done
cat gpurun_out/sweep3.txt
