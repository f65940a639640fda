#!/bin/bash
# round 3, first GPU pass: the new convex / cube tests, then the whole -m gpu suite, then a quick bench
mkdir -p gpurun_out/r3a
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "convex or cube" -s > gpurun_out/r3a/convex.log 2>&1
echo "convex rc=$?" >> gpurun_out/r3a/convex.log
python -m pytest tests -m gpu -x -q > gpurun_out/r3a/gpu_all.log 2>&1
echo "all rc=$?" >> gpurun_out/r3a/gpu_all.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r3a/bench_driver.json 2> gpurun_out/r3a/bench_driver.err
python bench.py --steps 500 --warmup 100 > gpurun_out/r3a/bench_500.json 2> gpurun_out/r3a/bench_500.err
tail -5 gpurun_out/r3a/convex.log; tail -5 gpurun_out/r3a/gpu_all.log; cat gpurun_out/r3a/bench_driver.json | cut -c1-400
