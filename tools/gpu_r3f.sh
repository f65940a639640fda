#!/bin/bash
mkdir -p gpurun_out/r3f
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r3f/gpu_all.log 2>&1; echo "rc=$?" >> gpurun_out/r3f/gpu_all.log
tail -4 gpurun_out/r3f/gpu_all.log
python bench.py --config cube --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r3f/bench_cube.json 2> gpurun_out/r3f/bench_cube.err
cut -c1-260 gpurun_out/r3f/bench_cube.json
MODEL=cube K=40 W=20 MJHIP_LIB=$PWD/tools/variants/libmjhip_prof.so timeout 300 python tools/stage_profile.py 2>/dev/null | head -30 > gpurun_out/r3f/stageprof_cube.txt
cat gpurun_out/r3f/stageprof_cube.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3f/bench_driver.json 2> gpurun_out/r3f/bench_driver.err
cut -c1-200 gpurun_out/r3f/bench_driver.json
