# Builds the library variants the measurement scripts use into tools/variants/ (git-ignored; they
# travel to the GPU box with the snapshot):
#   libmjhip_prof.so      -DMJH_PROFILE: per-stage wall-clock accumulators (tools/stage_profile.py)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/variants
python - <<'PY'
import __graft_entry__ as g
g._build_lib(extra_flags=["-DMJH_PROFILE"], lib="tools/variants/libmjhip_prof.so", objdir="tools/variants/obj_prof")
PY
ls -la tools/variants
