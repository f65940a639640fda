# HBM traffic attribution of the lean rollout kernel (runs on the GPU box): the same bench command under
#   base     : product build (128 VGPRs / 4 waves per SIMD, 10 KB LDS plan)
#   lds40k   : everything LDS resident (MJHIP_LDS_BYTES=40960): removes the constraint arrays that overflow the plan
#   wpe2     : 256-VGPR build (tools/variants/libmjhip_wpe2.so): removes register spills
#   both     : neither -> floor = algorithmic bytes + model constants + contact arrays
# two PMC passes each (FETCH_SIZE, WRITE_SIZE; counters only, no other tracing)
export TMPDIR=/tmp
ARGS="--no-extra --no-cpu-baseline --steps 100 --warmup 20"
run() {
  tag=$1; shift
  OUT=$PWD/gpurun_out/traffic_$tag
  mkdir -p $OUT
  env "$@" python bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err
  (cd /tmp && env "$@" rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python $OLDPWD/bench.py $ARGS > /dev/null 2> $OUT/pmc_fetch.err)
  (cd /tmp && env "$@" rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python $OLDPWD/bench.py $ARGS > /dev/null 2> $OUT/pmc_write.err)
  python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
  echo "== $tag"; python -c "import json;d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]);print('%.3f M env-steps/s' % (d['value']/1e6), d['config']['mapping'])"; tail -1 $OUT/pmc_summary.txt
  rm -rf $OUT/pmc_fetch $OUT/pmc_write
}
run base X=1
run lds40k MJHIP_LDS_BYTES=40960
run wpe2 MJHIP_LIB=$PWD/tools/variants/libmjhip_wpe2.so
run both MJHIP_LIB=$PWD/tools/variants/libmjhip_wpe2.so MJHIP_LDS_BYTES=40960
