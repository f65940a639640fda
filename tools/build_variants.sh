# Builds the library variants the measurement scripts use into tools/variants/ (git-ignored; they
# travel to the GPU box with the snapshot):
#   libmjhip_w{2,3,4}.so  register budget for 2/3/4 waves per SIMD (tools/gpu_sweep.sh)
#   libmjhip_prof.so      -DMJH_PROFILE: per-stage wall-clock accumulators (tools/stage_profile.py)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/variants
INC=${MUJOCO_INCLUDE:-/root/reference/include}
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared -Wno-enum-compare -I$INC"
for w in 2 3 4; do
  /opt/rocm/bin/hipcc $FLAGS -DMJH_WAVES_PER_EU=$w mujoco_amd/csrc/mjh_hip.hip -o tools/variants/libmjhip_w$w.so
done
/opt/rocm/bin/hipcc $FLAGS -DMJH_PROFILE mujoco_amd/csrc/mjh_hip.hip -o tools/variants/libmjhip_prof.so
ls -la tools/variants
