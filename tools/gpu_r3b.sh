#!/bin/bash
# round 3, second GPU pass: tests with the custom sincos build, cube bench, stage profiles, SQ counters
mkdir -p gpurun_out/r3b
python -m pytest tests -m gpu -x -q > gpurun_out/r3b/gpu_all.log 2>&1; echo "rc=$?" >> gpurun_out/r3b/gpu_all.log
tail -3 gpurun_out/r3b/gpu_all.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3b/bench_driver.json 2> gpurun_out/r3b/bench_driver.err
python bench.py --steps 500 --warmup 100 --no-extra > gpurun_out/r3b/bench_500.json 2> gpurun_out/r3b/bench_500.err
python bench.py --config cube --steps 100 --warmup 20 > gpurun_out/r3b/bench_cube.json 2> gpurun_out/r3b/bench_cube.err
cut -c1-600 gpurun_out/r3b/bench_driver.json; cut -c1-300 gpurun_out/r3b/bench_500.json; cat gpurun_out/r3b/bench_cube.json | cut -c1-1500; tail -3 gpurun_out/r3b/bench_cube.err
MODEL=cube K=40 W=20 MJHIP_LIB=$PWD/tools/variants/libmjhip_prof.so python tools/stage_profile.py > gpurun_out/r3b/stageprof_cube.txt 2>&1
head -40 gpurun_out/r3b/stageprof_cube.txt
bash tools/gpu_sq.sh r3b > gpurun_out/r3b/sq.log 2>&1
tail -60 gpurun_out/sq_r3b/sq_summary.txt
