#!/bin/bash
# round 3 final profile set: suite, flex (config 5) / cube (config 4) / humanoid (config 2: driver configuration and 500 steps)
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3n_tests.log 2>&1; tail -2 gpurun_out/r3n_tests.log
bash tools/gpu_profile.sh r03_flex --config flex --steps 200 > gpurun_out/r03_flex.log 2>&1
for n in 1024 4096; do timeout 600 python bench.py --config flex --steps 100 --no-extra --envs-per-gpu $n > gpurun_out/prof_r03_flex/bench_$n.json 2>> gpurun_out/prof_r03_flex/bench.err; done
bash tools/gpu_r3g.sh > gpurun_out/r3n_r3g.log 2>&1
MODEL=flex NENV=256 K=50 W=400 MJHIP_LIB=$PWD/tools/variants/libmjhip_prof.so timeout 600 python tools/stage_profile.py > gpurun_out/prof_r03_flex/stage_profile_flex.txt 2>&1
line() { python -c "
import json,sys; j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(j['value']), round(j['ms_per_step'],3), j['roofline']['kernel'], j.get('parity_sample',{}).get('ok'))" $1 "$2"; }
for t in flex steps20 steps500 cube; do line gpurun_out/prof_r03_$t/bench_full.json "$t:"; tail -1 gpurun_out/prof_r03_$t/pmc_summary.txt; done
line gpurun_out/prof_r03_flex/bench_1024.json "flex 1024:"; line gpurun_out/prof_r03_flex/bench_4096.json "flex 4096:"
