#!/bin/bash
# round 3 checkpoint: all GPU tests, humanoid bench (driver config, 500 steps), cube bench
mkdir -p gpurun_out/r3e
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r3e/gpu_all.log 2>&1; echo "rc=$?" >> gpurun_out/r3e/gpu_all.log
tail -6 gpurun_out/r3e/gpu_all.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r3e/bench_driver.json 2> gpurun_out/r3e/bench_driver.err
python bench.py --steps 500 --warmup 100 --no-extra > gpurun_out/r3e/bench_500.json 2> gpurun_out/r3e/bench_500.err
python bench.py --config cube --steps 100 --warmup 20 > gpurun_out/r3e/bench_cube.json 2> gpurun_out/r3e/bench_cube.err
cut -c1-400 gpurun_out/r3e/bench_driver.json; cut -c1-300 gpurun_out/r3e/bench_500.json; cut -c1-300 gpurun_out/r3e/bench_cube.json
python - <<'PY'
import json
for f in ("bench_driver", "bench_cube"):
    d = json.load(open("gpurun_out/r3e/%s.json" % f))
    print(f, "parity_sample:", json.dumps(d.get("parity_sample"))[:700])
    print(f, "cpu_baseline:", json.dumps(d.get("cpu_baseline"))[:900])
PY
