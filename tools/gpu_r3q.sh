#!/bin/bash
# batched warm-start passes + pipelined serial cost sum + two elements per lane in the stretch pass: flex tests, lines, profile
timeout 900 python -m pytest tests/test_flex_gpu.py -x -q > gpurun_out/r3q_tests.log 2>&1; tail -2 gpurun_out/r3q_tests.log
line() { python -c "
import json,sys; j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(j['value']), round(j['ms_per_step'],3), j['roofline']['kernel'])" $1 "$2"; }
timeout 600 python bench.py --config flex --steps 200 --no-extra > gpurun_out/r3q_flex256.json 2> gpurun_out/r3q.err; line gpurun_out/r3q_flex256.json "flex 256:"
timeout 600 python bench.py --config flex --steps 100 --no-extra --envs-per-gpu 4096 > gpurun_out/r3q_flex4096.json 2>> gpurun_out/r3q.err; line gpurun_out/r3q_flex4096.json "flex 4096:"
if [ -f tools/variants/libmjhip_prof.so ]; then bash tools/gpu_flex2.sh | grep -v Warning | head -30; fi
grep -v amdgpu.ids gpurun_out/r3q.err | tail -3
