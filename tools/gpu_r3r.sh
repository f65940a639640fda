#!/bin/bash
# end of round 3: suite, smoke, the driver's bench invocations (default and --steps 20 --warmup 5), flex profile set
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3r_tests.log 2>&1; tail -2 gpurun_out/r3r_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3r_smoke.log 2>&1; tail -2 gpurun_out/r3r_smoke.log
( time timeout 900 python bench.py > gpurun_out/r3r_bench_default.json 2> gpurun_out/r3r_bench_default.err ) 2>&1 | grep real
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3r_bench_driver.json 2> gpurun_out/r3r_bench_driver.err ) 2>&1 | grep real
python - <<'PY'
import json
for f in ("default", "driver"):
    j = json.loads(open("gpurun_out/r3r_bench_%s.json" % f).read().strip().splitlines()[-1])
    print(f, round(j["value"]), j["ms_per_step"], j["steps"], j["roofline"]["frac"], j["roofline"].get("valu_insts_per_env_step"), j["cpu_baseline"]["value"], j["parity_sample"]["ok"])
PY
bash tools/gpu_profile.sh r03_flex --config flex --steps 200 > gpurun_out/r03_flex.log 2>&1
for n in 1024 4096; do timeout 600 python bench.py --config flex --steps 100 --no-extra --envs-per-gpu $n > gpurun_out/prof_r03_flex/bench_$n.json 2>> gpurun_out/prof_r03_flex/bench.err; done
MODEL=flex NENV=256 K=50 W=400 MJHIP_LIB=$PWD/tools/variants/libmjhip_prof.so timeout 600 python tools/stage_profile.py > gpurun_out/prof_r03_flex/stage_profile_flex.txt 2>&1
python - <<'PY'
import json
j = json.loads(open("gpurun_out/prof_r03_flex/bench_full.json").read().strip().splitlines()[-1])
print("flex", round(j["value"]), j["ms_per_step"], j["parity_sample"]["ok"], j["parity_sample"]["reference_glibc"]["identical_input_steps"]["bit_exact_steps"], j["cpu_baseline"]["rollout_regime"]["value"])
PY
tail -1 gpurun_out/prof_r03_flex/pmc_summary.txt; head -3 gpurun_out/prof_r03_flex/kernel_stats.csv
