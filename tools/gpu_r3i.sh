#!/bin/bash
# round 3, re-entry check of the committed build: the whole GPU suite, the flex profile set
# (BASELINE config 5), quick humanoid / cube lines, flex stage profile
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3i_tests.log 2>&1; tail -3 gpurun_out/r3i_tests.log
bash tools/gpu_profile.sh r03_flex --config flex --steps 200 > gpurun_out/r03_flex.log 2>&1
tail -3 gpurun_out/prof_r03_flex/pmc_summary.txt; head -4 gpurun_out/prof_r03_flex/kernel_stats.csv
for n in 1024 4096; do timeout 600 python bench.py --config flex --steps 100 --no-extra --envs-per-gpu $n > gpurun_out/prof_r03_flex/bench_$n.json 2>> gpurun_out/prof_r03_flex/bench.err; tail -c 400 gpurun_out/prof_r03_flex/bench_$n.json; done
timeout 300 python bench.py --steps 20 --warmup 5 --no-extra > gpurun_out/r3i_humanoid20.json 2> gpurun_out/r3i_h.err; tail -c 600 gpurun_out/r3i_humanoid20.json
timeout 300 python bench.py --steps 500 --warmup 100 --no-extra > gpurun_out/r3i_humanoid500.json 2>> gpurun_out/r3i_h.err; tail -c 600 gpurun_out/r3i_humanoid500.json
timeout 300 python bench.py --config cube --steps 100 --warmup 20 --no-extra > gpurun_out/r3i_cube.json 2>> gpurun_out/r3i_h.err; tail -c 600 gpurun_out/r3i_cube.json
if [ -f tools/variants/libmjhip_prof.so ]; then bash tools/gpu_flex2.sh | head -34; fi
