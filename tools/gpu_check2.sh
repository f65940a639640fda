set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
for lds in 10240 9216; do
echo "== lds=$lds"
MJHIP_LDS_BYTES=$lds timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-230
done
MJHIP_LIB=$PWD/tools/variants/libmjhip_prof.so python tools/stage_profile.py > gpurun_out/stageprof.txt 2>&1
cat gpurun_out/stageprof.txt
