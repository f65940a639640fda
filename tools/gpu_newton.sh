set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/newton_tests.txt
cat gpurun_out/newton_tests.txt
python bench.py --solver newton --steps 100 --warmup 20 --no-extra > gpurun_out/newton_bench.txt 2>&1
tail -2 gpurun_out/newton_bench.txt
python bench.py --solver cg --steps 100 --warmup 20 --no-extra > gpurun_out/cg_bench.txt 2>&1
tail -2 gpurun_out/cg_bench.txt
SOLVER=2 MJHIP_VARIANT=generic MJHIP_LIB=$PWD/tools/variants/libmjhip_prof.so python tools/stage_profile.py > gpurun_out/stageprof_newton.txt 2>&1
head -30 gpurun_out/stageprof_newton.txt
python bench.py --steps 100 --warmup 20 --no-extra > gpurun_out/pgs_bench.txt 2>&1
tail -1 gpurun_out/pgs_bench.txt
