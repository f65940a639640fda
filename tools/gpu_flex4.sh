#!/bin/bash
# flex: tests, bench at 256 (config 5) / 1024 / 4096 environments per GPU, stage profile
timeout 900 python -m pytest tests/test_flex_gpu.py -x -q > gpurun_out/flex_tests.log 2>&1; tail -2 gpurun_out/flex_tests.log
timeout 1200 python bench.py --config flex --steps 200 --no-cpu-baseline > gpurun_out/flex_bench_256.json 2> gpurun_out/flex_bench.err; python -c "
import json; j=json.loads(open('gpurun_out/flex_bench_256.json').read().strip().splitlines()[-1]); print('256:', j['value'], j['ms_per_step'], j['parity_sample']['ok'], j['parity_sample']['reference_glibc']['identical_input_steps']['bit_exact_steps'])"
for n in 1024 4096; do timeout 1200 python bench.py --config flex --steps 100 --no-extra --envs-per-gpu $n > gpurun_out/flex_bench_$n.json 2>> gpurun_out/flex_bench.err; python -c "
import json; j=json.loads(open('gpurun_out/flex_bench_$n.json').read().strip().splitlines()[-1]); print('$n:', j['value'], j['ms_per_step'])"; done
tail -3 gpurun_out/flex_bench.err
bash tools/gpu_flex2.sh | head -28
