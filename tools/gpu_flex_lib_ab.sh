#!/bin/bash
# flex bench (--no-extra) of a variant library ($MJHIP_LIB) against the shipped one and earlier trees, alternating rounds
#   bash tools/gpu_flex_lib_ab.sh <outdir> <variant.so> [tree names ...]
set -u
OUT=${1:?outdir}; VAR=${2:?variant}; shift 2
mkdir -p "$OUT"; HERE=$PWD
val() { python - "$1" <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    print("%.1f env-steps/s  ms/step %.4f" % (j["value"], j["ms_per_step"]))
except Exception as exc:
    print("no line:", exc)
PY
}
for i in 1 2 3; do
  echo "-- round $i"
  for t in "$@"; do
    ( cd tools/variants/${t}_tree && timeout 600 python bench.py --config flex --steps 200 --no-extra > "$HERE/$OUT/${t}_$i.json" 2> /dev/null )
    printf "  %-28s " "$t:"; val "$OUT/${t}_$i.json"
  done
  MJHIP_LIB=$PWD/$VAR timeout 600 python bench.py --config flex --steps 200 --no-extra > "$OUT/var_$i.json" 2> /dev/null
  printf "  %-28s " "$(basename $VAR):"; val "$OUT/var_$i.json"
  timeout 600 python bench.py --config flex --steps 200 --no-extra > "$OUT/cur_$i.json" 2> /dev/null
  printf "  %-28s " "shipped:"; val "$OUT/cur_$i.json"
done
