# usage: bash tools/gpu_ktrace.sh <tag> [env assignments...]  -- kernel-trace stats of a short bench
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/ktrace_$TAG
mkdir -p $OUT
cd /tmp
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $OLDPWD/bench.py --no-cpu-baseline --steps 50 --warmup 10 > $OUT/bench.json 2> $OUT/trace.err
cd $OLDPWD
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
cut -c1-60,1000- $OUT/kernel_stats.csv | head -3 >/dev/null
python - <<PY
import csv
for r in csv.DictReader(open("$OUT/kernel_stats.csv")):
    print("%-40s calls=%6s avg_us=%10.1f total_ms=%9.2f  %5s%%" % (r["Name"][:40], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, r["Percentage"][:5]))
PY
rm -rf $OUT/trace
