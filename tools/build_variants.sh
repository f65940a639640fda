# Builds the library variants the measurement scripts use into tools/variants/ (git-ignored; they
# travel to the GPU box with the snapshot):
#   libmjhip_prof.so      -DMJH_PROFILE: per-stage wall-clock accumulators (tools/stage_profile.py)
#   libmjhip_wpe2.so      lean kernel compiled for 2 waves per SIMD (256 VGPRs, no spills): traffic attribution
#   libmjhip_nolaunder.so lean kernel without the per-step descriptor laundering (tools/gpu_pmc_ab_launder.sh)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/variants
python - <<'PY'
import __graft_entry__ as g
g._build_lib(extra_flags=["-DMJH_PROFILE"], lib="tools/variants/libmjhip_prof.so", objdir="tools/variants/obj_prof")
# the lean kernel with a 256-VGPR budget (2 waves per SIMD): no spills; the other units are reused
g._build_lib(extra_flags=["-DMJH_LEAN_WPE=2"], lib="tools/variants/libmjhip_wpe2.so", objdir="tools/variants/obj_wpe2",
             units=["mjh_kern_lean.hip"], reuse="mujoco_amd/csrc/build")
g._build_lib(extra_flags=["-DMJH_NO_LAUNDER"], lib="tools/variants/libmjhip_nolaunder.so", objdir="tools/variants/obj_nolaunder",
             units=["mjh_kern_lean.hip"], reuse="mujoco_amd/csrc/build")
PY
ls -la tools/variants
