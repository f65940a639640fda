#!/bin/bash
# round 3 profile set of the committed build: humanoid (driver configuration and 500 steps) and cube
bash tools/gpu_profile.sh r03_steps20 --steps 20 --warmup 5 > gpurun_out/r03_steps20.log 2>&1
bash tools/gpu_profile.sh r03_steps500 --steps 500 --warmup 100 > gpurun_out/r03_steps500.log 2>&1
bash tools/gpu_profile.sh r03_cube --config cube --steps 100 --warmup 20 > gpurun_out/r03_cube.log 2>&1
MODEL=humanoid K=100 W=20 MJHIP_LIB=$PWD/tools/variants/libmjhip_prof.so timeout 300 python tools/stage_profile.py > gpurun_out/prof_r03_steps500/stage_profile_lean.txt 2>&1
MODEL=cube K=40 W=20 MJHIP_LIB=$PWD/tools/variants/libmjhip_prof.so timeout 300 python tools/stage_profile.py > gpurun_out/prof_r03_cube/stage_profile_cube.txt 2>&1
python tools/tail_stats.py > gpurun_out/prof_r03_steps500/tail_stats.txt 2>&1
tail -3 gpurun_out/prof_r03_steps20/pmc_summary.txt gpurun_out/prof_r03_steps500/pmc_summary.txt gpurun_out/prof_r03_cube/pmc_summary.txt
cat gpurun_out/prof_r03_steps20/kernel_stats.csv | head -5; cat gpurun_out/prof_r03_cube/kernel_stats.csv | head -5
