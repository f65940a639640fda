set -x
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/inl
mkdir -p $OUT
export MJHIP_LIB=$PWD/tools/variants/libmjhip_inl2.so
python bench.py --no-extra --steps 500 --warmup 100 > $OUT/bench.json 2> $OUT/err
python -c "import json;d=json.load(open('$OUT/bench.json'));print('inlined 500/100 %.3fM launch %.2f ms' % (d['value']/1e6, d['roofline']['launch_ms']))"
python bench.py --no-extra --steps 20 --warmup 5 > $OUT/bench20.json 2> $OUT/err
python -c "import json;d=json.load(open('$OUT/bench20.json'));print('inlined 20/5 %.3fM launch %.2f ms' % (d['value']/1e6, d['roofline']['launch_ms']))"
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python $OLDPWD/bench.py --no-extra --steps 500 --warmup 100 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python $OLDPWD/bench.py --no-extra --steps 500 --warmup 100 > /dev/null 2> $OUT/pmc_write.err
cd $OLDPWD
python tools/pmc_summary.py $OUT
rm -rf $OUT/pmc_*/*/*.db
