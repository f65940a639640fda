#!/bin/bash
# SQ counters per stage for the cube (generic kernel, Newton on the sparse path)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/sq_cube
mkdir -p $OUT
SETS=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
      "SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT")
cd /tmp
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  MJHIP_LDS_BYTES=20480 MODEL=cube REGIME=uniform SETTLE=60 REPS=2 timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/stages_uniform_p$i -o pmc -- python $OLDPWD/tools/sq_stages.py > $OUT/stages_uniform_p$i.log 2> $OUT/stages_uniform_p$i.err
done
cd $OLDPWD
python tools/sq_summary.py $OUT > $OUT/sq_summary.txt 2>&1
cat $OUT/sq_summary.txt | cut -c1-400
rm -rf $OUT/*/*/*.db 2>/dev/null
find $OUT -name "*.csv" -size +2M -delete
