# usage: bash tools/gpu_profile.sh <tag> [bench args...]   (runs on the GPU box via gpurun)
# bench line (with every leg), rocprofv3 kernel trace + stats, and the two PMC passes of the SAME
# command (timed region only: --no-extra), summarised by tools/pmc_summary.py
set -x
TAG=${1:-r02}; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
python bench.py "$@" > $OUT/bench_full.json 2> $OUT/bench.err; tail -c 3000 $OUT/bench_full.json
python bench.py --no-extra "$@" > $OUT/bench.json 2>> $OUT/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $OLDPWD/bench.py --no-extra "$@" > $OUT/bench_traced.json 2> $OUT/trace.err
cd $OLDPWD
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
cat $OUT/kernel_stats.csv
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python $OLDPWD/bench.py --no-extra "$@" > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python $OLDPWD/bench.py --no-extra "$@" > /dev/null 2> $OUT/pmc_write.err
cd $OLDPWD
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1; cat $OUT/pmc_summary.txt
# keep the merged payload small
rm -rf $OUT/trace/*/*.db $OUT/pmc_fetch/*/*.db $OUT/pmc_write/*/*.db 2>/dev/null
find $OUT -name "*.csv" -size +2M -delete
du -sh $OUT
