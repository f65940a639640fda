set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in ${VARIANTS:-lean generic}; do
MJHIP_VARIANT=$v MJHIP_LIB=$PWD/tools/variants/libmjhip_prof.so python tools/stage_profile.py > gpurun_out/stageprof_$v.txt 2>&1
head -30 gpurun_out/stageprof_$v.txt
done
