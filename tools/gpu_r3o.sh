#!/bin/bash
# flex edge equalities on the device + SQ counters of the final build
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3o_tests.log 2>&1; tail -2 gpurun_out/r3o_tests.log
timeout 300 python bench.py --config flex --steps 200 --no-extra > gpurun_out/r3o_flex256.json 2> gpurun_out/r3o.err; tail -c 200 gpurun_out/r3o_flex256.json
bash tools/gpu_sq.sh r03d > gpurun_out/r3o_sq.log 2>&1; tail -40 gpurun_out/sq_r03d/sq_summary.txt
