#!/bin/bash
# SQ counters of the committed build: per stage (forward-kernel dispatches) and for the rollout kernel, both regimes
# usage: bash tools/gpu_sq.sh <tag> [config] [regimes]      (config: a bench.py configuration, default humanoid; regimes: "uniform testspeed")
TAG=${1:-r03}
CONFIG=${2:-humanoid}
REGIMES=${3:-uniform testspeed}
export MODEL=$CONFIG
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/sq_$TAG
mkdir -p $OUT
SETS=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
      "SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT" \
      "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_THREAD_CYCLES_VALU")
cd /tmp
for regime in $REGIMES; do
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    REGIME=$regime timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/stages_${regime}_p$i -o pmc -- python $OLDPWD/tools/sq_stages.py > $OUT/stages_${regime}_p$i.log 2> $OUT/stages_${regime}_p$i.err
    if [ $regime = uniform ]; then ARGS="--steps 100 --warmup 20"; else ARGS="--steps 100 --warmup 20 --settle 1000 --ctrl ou-halton"; fi
    timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/rollout_${regime}_p$i -o pmc -- python $OLDPWD/bench.py --config $CONFIG --no-extra --no-legs --no-cpu-baseline $ARGS > $OUT/rollout_${regime}_p$i.log 2> $OUT/rollout_${regime}_p$i.err
  done
done
cd $OLDPWD
python tools/sq_summary.py $OUT > $OUT/sq_summary.txt 2>&1
cat $OUT/sq_summary.txt
rm -rf $OUT/*/*/*.db 2>/dev/null
find $OUT -name "*.csv" -size +2M -delete
du -sh $OUT
