set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200; done
timeout 300 python bench.py --no-cpu-baseline --solver newton --steps 200 --warmup 50 2>&1 | tail -1 | cut -c1-200
