#!/bin/bash
# PMC FETCH_SIZE / WRITE_SIZE passes of the humanoid driver line on the product build and on the -DMJH_NO_LAUNDER lean kernel
# (tools/build_variants.sh builds tools/variants/libmjhip_nolaunder.so); summary: profiles/r04/launder_traffic_ab.txt
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04n
mkdir -p $OUT
cd /tmp
for v in product nolaunder; do
  if [ $v = nolaunder ]; then export MJHIP_LIB=$OLDPWD/tools/variants/libmjhip_nolaunder.so; else unset MJHIP_LIB; fi
  python $OLDPWD/bench.py --no-extra --gpus 1 --steps 20 --warmup 5 > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$v/pmc_fetch -o pmc -- python $OLDPWD/bench.py --no-extra --gpus 1 --steps 20 --warmup 5 > /dev/null 2> $OUT/$v.fetch.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/$v/pmc_write -o pmc -- python $OLDPWD/bench.py --no-extra --gpus 1 --steps 20 --warmup 5 > /dev/null 2> $OUT/$v.write.err
  cp $OUT/bench_$v.json $OUT/$v/bench.json
  python $OLDPWD/tools/pmc_summary.py $OUT/$v > $OUT/pmc_summary_$v.txt 2>&1
  echo "== $v"; tail -1 $OUT/pmc_summary_$v.txt; python -c "import json; j=json.loads(open('$OUT/bench_$v.json').read().splitlines()[-1]); print(round(j['value']))"
done
rm -rf $OUT/*/pmc_*/*/*.db 2>/dev/null; find $OUT -name "*.csv" -size +2M -delete
