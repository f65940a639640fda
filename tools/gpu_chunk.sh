set -x
export TMPDIR=/tmp
for c in 50 100 250 500; do timeout 300 python bench.py --no-cpu-baseline --chunk $c 2>&1 | tail -1 | cut -c1-160; done
timeout 300 python bench.py --no-cpu-baseline --steps 1000 --warmup 100 --chunk 1000 2>&1 | tail -1 | cut -c1-160
