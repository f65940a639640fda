# round 2, GPU run A: tests, per-variant throughput, default bench
set -x
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r2a
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest.txt; cat $OUT/pytest.txt
for v in generic lean lean2 lean4; do
  MJHIP_VARIANT=$v timeout 300 python bench.py --no-extra --steps 200 --warmup 100 > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  tail -c 1500 $OUT/bench_$v.json; tail -3 $OUT/bench_$v.err
done
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; cat $OUT/bench_driver.json; tail -5 $OUT/bench_driver.err
