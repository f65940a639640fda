#!/bin/bash
# gradient formed in the J'f pass: flex tests + flex lines (256 / 4096 environments)
timeout 900 python -m pytest tests/test_flex_gpu.py -x -q > gpurun_out/r3p_tests.log 2>&1; tail -2 gpurun_out/r3p_tests.log
line() { python -c "
import json,sys; j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(j['value']), round(j['ms_per_step'],3), j['roofline']['kernel'])" $1 "$2"; }
timeout 600 python bench.py --config flex --steps 200 --no-extra > gpurun_out/r3p_flex256.json 2> gpurun_out/r3p.err; line gpurun_out/r3p_flex256.json "flex 256:"
timeout 600 python bench.py --config flex --steps 100 --no-extra --envs-per-gpu 4096 > gpurun_out/r3p_flex4096.json 2>> gpurun_out/r3p.err; line gpurun_out/r3p_flex4096.json "flex 4096:"
grep -v amdgpu.ids gpurun_out/r3p.err | tail -3
