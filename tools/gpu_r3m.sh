#!/bin/bash
# merged ordered-sum passes + ping-pong chains + big short-lived fields overlaying the solver region: suite, flex A/B, profile
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3m_tests.log 2>&1; tail -2 gpurun_out/r3m_tests.log
line() { python -c "
import json,sys; j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(j['value']), round(j['ms_per_step'],3), j['roofline']['kernel'])" $1 "$2"; }
timeout 600 python bench.py --config flex --steps 200 --no-extra > gpurun_out/r3m_flex256.json 2> gpurun_out/r3m.err; line gpurun_out/r3m_flex256.json "flex 256:"
MJHIP_NO_BIG_OVERLAY=1 timeout 600 python bench.py --config flex --steps 200 --no-extra > gpurun_out/r3m_flex256_nooverlay.json 2>> gpurun_out/r3m.err; line gpurun_out/r3m_flex256_nooverlay.json "flex 256 (no overlay of big fields):"
if [ -f tools/variants/libmjhip_prof.so ]; then bash tools/gpu_flex2.sh | grep -v Warning | head -30; fi
grep -v amdgpu.ids gpurun_out/r3m.err | tail -5
