#!/bin/bash
# wide (256-VGPR) generic kernels + direct LDS chains: suite, A/B lines for flex and cube, flex stage profile
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3k_tests.log 2>&1; tail -3 gpurun_out/r3k_tests.log
line() { python -c "
import json,sys; j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(j['value']), round(j['ms_per_step'],3), j['roofline']['kernel'])" $1 "$2"; }
timeout 600 python bench.py --config flex --steps 200 --no-extra > gpurun_out/r3k_flex256.json 2> gpurun_out/r3k.err; line gpurun_out/r3k_flex256.json "flex 256:"
MJHIP_WIDE_REGS=0 timeout 600 python bench.py --config flex --steps 200 --no-extra > gpurun_out/r3k_flex256_narrow.json 2>> gpurun_out/r3k.err; line gpurun_out/r3k_flex256_narrow.json "flex 256 (128-VGPR kernel):"
timeout 300 python bench.py --config cube --steps 100 --warmup 20 --no-extra > gpurun_out/r3k_cube.json 2>> gpurun_out/r3k.err; line gpurun_out/r3k_cube.json "cube:"
MJHIP_WIDE_REGS=0 timeout 300 python bench.py --config cube --steps 100 --warmup 20 --no-extra > gpurun_out/r3k_cube_narrow.json 2>> gpurun_out/r3k.err; line gpurun_out/r3k_cube_narrow.json "cube (128-VGPR kernel):"
MJHIP_LDS_BYTES=40960 timeout 300 python bench.py --config cube --steps 100 --warmup 20 --no-extra > gpurun_out/r3k_cube_40k.json 2>> gpurun_out/r3k.err; line gpurun_out/r3k_cube_40k.json "cube (40 KB LDS):"
if [ -f tools/variants/libmjhip_prof.so ]; then bash tools/gpu_flex2.sh | grep -v Warning | head -32; fi
timeout 300 python bench.py --steps 20 --warmup 5 --no-extra > gpurun_out/r3k_humanoid20.json 2>> gpurun_out/r3k.err; line gpurun_out/r3k_humanoid20.json "humanoid 20:"
timeout 300 python bench.py --solver newton --steps 100 --no-extra > gpurun_out/r3k_humanoid_newton.json 2>> gpurun_out/r3k.err; line gpurun_out/r3k_humanoid_newton.json "humanoid newton:"
grep -v amdgpu.ids gpurun_out/r3k.err | tail -5
