#!/bin/bash
# cube: LDS budget sweep (stage profile + bench) and the list of golden steps that are not bit-exact on the GPU
mkdir -p gpurun_out/r3d
for L in 10240 20480 40960; do
  MJHIP_LDS_BYTES=$L MODEL=cube K=40 W=20 MJHIP_LIB=$PWD/tools/variants/libmjhip_prof.so timeout 300 python tools/stage_profile.py 2>/dev/null | head -28 > gpurun_out/r3d/stageprof_cube_$L.txt
  echo "== LDS $L"; grep -E "wall|collision|make|constraint|euler|inside|LDS plan" gpurun_out/r3d/stageprof_cube_$L.txt
  MJHIP_LDS_BYTES=$L timeout 300 python bench.py --config cube --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | cut -c1-240
done
