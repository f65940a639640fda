#!/bin/bash
# driver configuration (--steps 20 --warmup 5): steps per launch
for c in 20 10 5 4 2; do
  for rep in 1 2; do
    timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --chunk $c --no-extra > gpurun_out/r3s_c$c.json 2> gpurun_out/r3s.err
    python -c "
import json,sys; j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('chunk', sys.argv[2], round(j['value']), round(j['ms_per_step'],4), round(j['roofline']['kernel_ms_total'],3), round(j['roofline']['wall_ms_total'],3))" gpurun_out/r3s_c$c.json $c
  done
done
