#!/bin/bash
# A/B of two builds of the library on ONE box, alternating pairs: the shipped libmjhip.so against a variant
# (tools/variants/*.so, selected through $MJHIP_LIB) on the humanoid metric (driver configuration), its testspeed regime
# and the opt-in residual PGS sweep.
#   bash tools/gpu_ab_lib.sh <outdir> <variant.so> [pairs]
set -u
OUT=${1:?outdir}; VAR=${2:?variant library}; N=${3:-3}
mkdir -p "$OUT"
ARGS="--gpus 1 --steps 20 --warmup 5 --no-legs --no-cpu-baseline --no-newton-regime --api-steps 0 --parity-envs 0"
val() { python - "$1" <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    print("value %.4g  testspeed_regime %.4g  pgs_residual %s / testspeed %s" % (j["value"], j.get("testspeed_regime_value") or float("nan"),
          "%.4g" % j["pgs_residual_value"] if j.get("pgs_residual_value") else None,
          "%.4g" % j["pgs_residual_testspeed_regime_value"] if j.get("pgs_residual_testspeed_regime_value") else None))
except Exception as exc:
    print("no line:", exc)
PY
}
for i in $(seq 1 $N); do
  echo "-- pair $i"
  MJHIP_LIB=$PWD/$VAR timeout 600 python bench.py $ARGS > "$OUT/var_$i.json" 2> "$OUT/var_$i.err"
  echo -n "  variant $(basename $VAR): "; val "$OUT/var_$i.json"
  timeout 600 python bench.py $ARGS > "$OUT/cur_$i.json" 2> "$OUT/cur_$i.err"
  echo -n "  shipped:                 "; val "$OUT/cur_$i.json"
done
