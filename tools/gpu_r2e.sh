set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r2e_tests.txt
cat gpurun_out/r2e_tests.txt
SOLVER=2 MJHIP_VARIANT=generic MJHIP_LIB=$PWD/tools/variants/libmjhip_prof.so python tools/stage_profile.py > gpurun_out/stageprof_newton.txt 2>&1
head -24 gpurun_out/stageprof_newton.txt
MJHIP_VARIANT=generic MJHIP_LIB=$PWD/tools/variants/libmjhip_prof.so python tools/stage_profile.py > gpurun_out/stageprof_generic.txt 2>&1
head -24 gpurun_out/stageprof_generic.txt
MJHIP_VARIANT=lean MJHIP_LIB=$PWD/tools/variants/libmjhip_prof.so python tools/stage_profile.py > gpurun_out/stageprof_lean.txt 2>&1
head -24 gpurun_out/stageprof_lean.txt
for s in newton cg; do python bench.py --solver $s --steps 100 --warmup 20 --no-extra 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['config']['solver'], d['value'], d['config']['kernel_variant'])"; done
MJHIP_VARIANT=generic python bench.py --steps 100 --warmup 20 --no-extra 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['config']['solver'], d['value'], d['config']['kernel_variant'])"
