#!/bin/bash
# Flex (BASELINE config 5: jelly.xml, 256 envs, CG, settled contact regime): the current build against earlier trees
# (tools/variants/<name>_tree = `git archive <commit>`, its library built by its own __graft_entry__) on ONE box, alternating
# rounds, timed region only (--no-extra).  Round 5's driver-format lines read 0.324-0.335 M against round 4's 0.350 M; its own
# A/B only went back to a mid-round-5 commit.
#   bash tools/gpu_flex_ab.sh <outdir> [tree names ...]      default: r04 (the round-4 final commit 07854f7)
set -u
OUT=${1:-gpurun_out/flex_ab}; shift || true
TREES=${*:-r04}
mkdir -p "$OUT"
HERE=$PWD
val() { python - "$1" <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    print("%.1f env-steps/s  ms/step %.4f  kernel %s" % (j["value"], j["ms_per_step"], j["roofline"].get("kernel")))
except Exception as exc:
    print("no line:", exc)
PY
}
clk() { /opt/rocm/bin/rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -2 | tr '\n' ' '; echo; }
for i in 1 2 3; do
  echo "-- round $i"; clk
  for t in $TREES; do
    ( cd tools/variants/${t}_tree && timeout 600 python bench.py --config flex --steps 200 --no-extra > "$HERE/$OUT/${t}_$i.json" 2> "$HERE/$OUT/${t}_$i.err" )
    printf "  %-24s " "$t:"; val "$OUT/${t}_$i.json"
  done
  timeout 600 python bench.py --config flex --steps 200 --no-extra > "$OUT/cur_$i.json" 2> "$OUT/cur_$i.err"
  printf "  %-24s " "current:"; val "$OUT/cur_$i.json"
done
clk
