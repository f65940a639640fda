#!/bin/bash
# flex on the device: its tests, then the bench line of the flex configuration (short) and a quick timing split
timeout 900 python -m pytest tests/test_flex_gpu.py -x -q > gpurun_out/flex_tests.log 2>&1; tail -5 gpurun_out/flex_tests.log
timeout 900 python bench.py --config flex --steps 200 --no-cpu-baseline > gpurun_out/flex_bench.json 2> gpurun_out/flex_bench.err; tail -c 2500 gpurun_out/flex_bench.json; tail -5 gpurun_out/flex_bench.err
