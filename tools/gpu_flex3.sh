#!/bin/bash
# flex on the device: tests, bench line (with the CPU legs), stage profile
timeout 900 python -m pytest tests/test_flex_gpu.py -x -q > gpurun_out/flex_tests.log 2>&1; tail -3 gpurun_out/flex_tests.log
timeout 1200 python bench.py --config flex --steps 200 > gpurun_out/flex_bench.json 2> gpurun_out/flex_bench.err; tail -c 1200 gpurun_out/flex_bench.json; tail -3 gpurun_out/flex_bench.err
bash tools/gpu_flex2.sh | head -30
