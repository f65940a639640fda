set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -E "rel err|passed|failed|rc=" gpurun_out/pytest_gpu.log | tail -20
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
timeout 300 python bench.py --no-cpu-baseline --solver newton --steps 200 --warmup 50 2>&1 | tail -1 | cut -c1-200
