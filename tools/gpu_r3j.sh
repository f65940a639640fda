#!/bin/bash
# 160 KB LDS blocks for launches of one workgroup per CU (flex at 256 environments): suite, flex lines, stage profile
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3j_tests.log 2>&1; tail -3 gpurun_out/r3j_tests.log
timeout 600 python bench.py --config flex --steps 200 --no-extra > gpurun_out/r3j_flex256.json 2> gpurun_out/r3j.err; python -c "
import json; j=json.loads(open('gpurun_out/r3j_flex256.json').read().strip().splitlines()[-1]); print('flex 256:', j['value'], j['ms_per_step'], j['config']['mapping'])"
MJHIP_MAX_LDS=65536 timeout 600 python bench.py --config flex --steps 200 --no-extra > gpurun_out/r3j_flex256_64k.json 2>> gpurun_out/r3j.err; python -c "
import json; j=json.loads(open('gpurun_out/r3j_flex256_64k.json').read().strip().splitlines()[-1]); print('flex 256 (64 KB cap):', j['value'], j['ms_per_step'])"
timeout 900 python bench.py --config flex --steps 200 > gpurun_out/r3j_flex256_full.json 2>> gpurun_out/r3j.err; python -c "
import json; j=json.loads(open('gpurun_out/r3j_flex256_full.json').read().strip().splitlines()[-1]); print('flex 256 full:', j['value'], j['parity_sample']['ok'], j['parity_sample']['reference_glibc']['identical_input_steps']['bit_exact_steps'], j['cpu_baseline']['rollout_regime']['value'])"
if [ -f tools/variants/libmjhip_prof.so ]; then bash tools/gpu_flex2.sh | head -30; fi
timeout 300 python bench.py --steps 20 --warmup 5 --no-extra > gpurun_out/r3j_humanoid20.json 2>> gpurun_out/r3j.err; tail -c 300 gpurun_out/r3j_humanoid20.json
timeout 300 python bench.py --config cube --steps 100 --warmup 20 --no-extra > gpurun_out/r3j_cube.json 2>> gpurun_out/r3j.err; tail -c 300 gpurun_out/r3j_cube.json
tail -5 gpurun_out/r3j.err
