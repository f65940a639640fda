set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
for lds in 20480 0; do
MJHIP_LIB=$PWD/tools/variants/libmjhip_prof.so MJHIP_LDS_BYTES=$lds python tools/stage_profile.py > gpurun_out/stageprof_$lds.txt 2>&1
cat gpurun_out/stageprof_$lds.txt
done
