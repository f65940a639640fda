mkdir -p gpurun_out
: > gpurun_out/sweep4.txt
for cfg in "4 7168 50" "4 8192 50" "4 9216 50" "4 9728 50" "4 10240 50" "4 9216 25" "4 9216 100" "4 9216 10"; do
  set -- $cfg
  echo "== waves_per_eu=$1 lds=$2 chunk=$3" >> gpurun_out/sweep4.txt
  MJHIP_LIB=$PWD/tools/variants/libmjhip_w$1.so MJHIP_LDS_BYTES=$2 timeout 300 python bench.py --steps 400 --warmup 100 --chunk $3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
try:
    r = json.loads(sys.stdin.readline()); print(r['value'], r['ms_per_step'])
except Exception as ex: print('FAILED', ex)
" >> gpurun_out/sweep4.txt
done
cat gpurun_out/sweep4.txt
