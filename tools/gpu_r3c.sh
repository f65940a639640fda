#!/bin/bash
# round 3, sparse constraint path: GPU parity of the new tests, cube bench, cube stage profile
mkdir -p gpurun_out/r3c
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sparse or cube or newton" > gpurun_out/r3c/gpu_sparse.log 2>&1; echo "rc=$?" >> gpurun_out/r3c/gpu_sparse.log
tail -5 gpurun_out/r3c/gpu_sparse.log
timeout 600 python bench.py --config cube --steps 100 --warmup 20 > gpurun_out/r3c/bench_cube.json 2> gpurun_out/r3c/bench_cube.err
cut -c1-1200 gpurun_out/r3c/bench_cube.json; tail -3 gpurun_out/r3c/bench_cube.err
MODEL=cube K=40 W=20 MJHIP_LIB=$PWD/tools/variants/libmjhip_prof.so timeout 600 python tools/stage_profile.py > gpurun_out/r3c/stageprof_cube.txt 2>&1
head -32 gpurun_out/r3c/stageprof_cube.txt
