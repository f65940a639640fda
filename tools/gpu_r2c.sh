set -x
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r2c
mkdir -p $OUT
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --no-extra --steps 500 --warmup 100 > $OUT/b500_$tag.json 2> $OUT/err || tail -3 $OUT/err
  python -c "import json;d=json.load(open('$OUT/b500_$tag.json'));print('$tag 500/100 %.3fM launch %.2f ms' % (d['value']/1e6, d['roofline']['launch_ms']), d['config']['kernel_variant'])"
  env "$@" timeout 300 python bench.py --no-extra --steps 20 --warmup 5 > $OUT/b20_$tag.json 2> $OUT/err || tail -3 $OUT/err
  python -c "import json;d=json.load(open('$OUT/b20_$tag.json'));print('$tag 20/5 %.3fM launch %.2f ms' % (d['value']/1e6, d['roofline']['launch_ms']))"
}
run lean_bal MJHIP_VARIANT=lean
run lean_nobal MJHIP_VARIANT=lean MJHIP_BALANCE=0
run lean2_bal MJHIP_VARIANT=lean2
run lean2_nobal MJHIP_VARIANT=lean2 MJHIP_BALANCE=0
run generic_bal MJHIP_VARIANT=generic
run lean_dpp32 MJHIP_VARIANT=lean MJHIP_LIB=$PWD/tools/variants/libmjhip_dpp32.so
