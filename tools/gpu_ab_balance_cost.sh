export TMPDIR=/tmp
# humanoid driver line, three alternating pairs: launch dealing by the work estimate vs the measured wall time of the previous launch
mkdir -p gpurun_out/r04aj
for i in 1 2 3; do
  for v in work wall; do
    MJHIP_BALANCE_COST=$v timeout 120 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-legs > gpurun_out/r04aj/${v}_$i.json 2>/dev/null
    echo "$v $i $(python -c "import json;print(round(json.loads(open('gpurun_out/r04aj/${v}_$i.json').read().splitlines()[-1])['value']))")"
  done
done
