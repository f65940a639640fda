#!/bin/bash
# One parameterised GPU job script (runs on the GPU box through gpurun; everything lands under gpurun_out/<tag>/).
#   bash tools/gpu_job.sh <tag> <step> [<step> ...]
# steps:
#   suite            pytest -m gpu + smoke()
#   driver           bench.py exactly as the driver runs it (--gpus 1 --steps 20 --warmup 5; carries the `configs` legs)
#   default          bench.py with its defaults (500 steps), --no-legs
#   bench:<args>     bench.py <args> (use , for spaces: bench:--config,cube,--steps,100)  -> bench_<n>.json
#   ab:<VAR=VAL>:<args>   the same bench line with and without an environment variable (A/B)
#   profile:<args>   tools/gpu_profile.sh <tag>_<n> <args>: bench + rocprofv3 kernel stats + PMC FETCH/WRITE passes
#   stages:<MODEL>   per-stage profile (tools/stage_profile.py on tools/variants/libmjhip_prof.so)
#   sq:<config>      SQ instruction counters of the rollout kernel (tools/gpu_sq.sh)
#   settled          per-stage profile of the humanoid in its settled regime (1000 warm-up steps), exact and residual PGS
#   regime           tools/regime_stats.py (testspeed regime: per-environment cost against the constraint count), both sweeps
#   tail             tools/tail_stats.py
#   sweep            tools/model_sweep.py over tests/golden/sweep with libmjhip.so on the device, against the live oracle
#   sweeplong[:N]    the same over N steps (default 100) instead of the fixtures' 15
#   flexab[:t1,t2]   tools/gpu_flex_ab.sh: flex bench, current build against earlier trees (tools/variants/<t>_tree; default r04)
#   ablib:<variant.so>   tools/gpu_ab_lib.sh: shipped library against tools/variants/<variant.so>, three alternating pairs
#   resources        kernel resource usage (VGPRs, scratch, spills) of the shipped code object
# (the one-shot scripts of rounds 2-3 -- gpu_r2*.sh, gpu_r3[a-s].sh, gpu_flex*.sh -- were sequences of these steps;
#  their outputs are quoted in profiles/r02* and profiles/r03*)
set -u
TAG=${1:?tag}; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
n=0
summ() { python - "$1" <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
except Exception as exc:
    print("  (no JSON line:", exc, ")"); sys.exit(0)
print("  value %.4g %s  ms/step %.4g  frac %.3g  parity_ok %s  kernel %s  line %d B" % (j["value"], j["unit"], j["ms_per_step"], j["roofline"]["frac"], j.get("parity_ok"), j["roofline"].get("kernel"), len(json.dumps(j))))
cb = j.get("cpu_baseline") or {}
print("  cpu_baseline %s (%s s, x%s) testspeed %s | testspeed_regime %s newton_regime %s api %s" % (cb.get("value"), cb.get("seconds"), cb.get("repeats"), cb.get("testspeed_value"), j.get("testspeed_regime_value"), j.get("newton_regime_value"), j.get("api_regime_value")))
print("  pgs_residual %s (parity_ok %s, max rel err %s, niter differs on %s of %s steps) testspeed %s" % (j.get("pgs_residual_value"), j.get("pgs_residual_parity_ok"), j.get("pgs_residual_max_rel_err"), j.get("pgs_residual_iter_differs"), j.get("pgs_residual_steps"), j.get("pgs_residual_testspeed_regime_value")))
for name in ("cube", "flex", "slider_crank"):
    if ("leg_%s_value" % name) in j or ("leg_%s_error" % name) in j:
        print("  leg %-12s value %s frac %s parity_ok %s cpu %s wall %s %s" % (name, j.get("leg_%s_value" % name), j.get("leg_%s_roofline_frac" % name), j.get("leg_%s_parity_ok" % name), j.get("leg_%s_cpu_like_for_like" % name), j.get("leg_%s_wall_s" % name), j.get("leg_%s_error" % name, "")))
PY
}
for step in "$@"; do
  n=$((n+1))
  kind=${step%%:*}; rest=${step#*:}; [ "$rest" = "$step" ] && rest=""
  args=${rest//,/ }
  echo "== [$TAG] $step"
  case $kind in
    suite)
      timeout 1700 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; tail -3 "$OUT/pytest_gpu.log"
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; tail -2 "$OUT/smoke.log" ;;
    driver)
      ( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver.json" 2> "$OUT/bench_driver.err" ) 2>&1 | grep real
      summ "$OUT/bench_driver.json"; cp gpurun_out/bench_full_humanoid_n1.json "$OUT/bench_driver_full.json" 2>/dev/null ;;
    default)
      ( time timeout 900 python bench.py --no-legs > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" ) 2>&1 | grep real
      summ "$OUT/bench_default.json" ;;
    bench)
      timeout 900 python bench.py $args > "$OUT/bench_$n.json" 2> "$OUT/bench_$n.err"; echo "  args: $args"; summ "$OUT/bench_$n.json" ;;
    ab)
      var=${rest%%:*}; a2=${rest#*:}; a2=${a2//,/ }
      timeout 900 python bench.py $a2 > "$OUT/ab_${n}_base.json" 2> "$OUT/ab_$n.err"; echo "  base: $a2"; summ "$OUT/ab_${n}_base.json"
      env "$var" timeout 900 python bench.py $a2 > "$OUT/ab_${n}_var.json" 2>> "$OUT/ab_$n.err"; echo "  $var"; summ "$OUT/ab_${n}_var.json" ;;
    profile)
      bash tools/gpu_profile.sh "${TAG}_$n" $args > "$OUT/profile_$n.log" 2>&1; tail -4 "$OUT/profile_$n.log"
      mv "gpurun_out/prof_${TAG}_$n" "$OUT/prof_$n" 2>/dev/null ;;
    stages)
      MODEL=${rest:-humanoid} MJHIP_LIB=$PWD/tools/variants/libmjhip_prof.so timeout 900 python tools/stage_profile.py > "$OUT/stage_profile_${rest:-humanoid}.txt" 2>&1
      head -40 "$OUT/stage_profile_${rest:-humanoid}.txt" ;;
    settled)
      # the humanoid after 1000 steps of random actions (lying on the floor: nefc ~ 40+), exact and residual PGS sweeps
      W=1000 K=100 MODEL=humanoid MJHIP_LIB=$PWD/tools/variants/libmjhip_prof.so timeout 900 python tools/stage_profile.py > "$OUT/stage_profile_humanoid_settled.txt" 2>&1
      head -32 "$OUT/stage_profile_humanoid_settled.txt"
      MJHIP_PGS=residual W=1000 K=100 MODEL=humanoid MJHIP_LIB=$PWD/tools/variants/libmjhip_prof.so timeout 900 python tools/stage_profile.py > "$OUT/stage_profile_humanoid_settled_pgs_residual.txt" 2>&1
      head -32 "$OUT/stage_profile_humanoid_settled_pgs_residual.txt" ;;
    regime)
      timeout 900 python tools/regime_stats.py > "$OUT/regime_stats.txt" 2>&1; tail -14 "$OUT/regime_stats.txt"
      MJHIP_PGS=residual timeout 900 python tools/regime_stats.py > "$OUT/regime_stats_pgs_residual.txt" 2>&1; tail -14 "$OUT/regime_stats_pgs_residual.txt" ;;
    sq)
      bash tools/gpu_sq.sh ${TAG}_${rest:-humanoid} ${rest:-humanoid} uniform > "$OUT/sq_${rest:-humanoid}.log" 2>&1; tail -5 "$OUT/sq_${rest:-humanoid}.log"
      cp gpurun_out/sq_${TAG}_${rest:-humanoid}/sq_summary.txt "$OUT/sq_summary_${rest:-humanoid}.txt" 2>/dev/null ;;
    tail)
      timeout 600 python tools/tail_stats.py > "$OUT/tail_stats.txt" 2>&1; tail -12 "$OUT/tail_stats.txt" ;;
    sweep)
      # against the reference linked with the kernels' own sin / cos (what the device reproduces to the bit), then against
      # the reference as built (glibc libm): the difference between the two columns is libm's last bit
      ( time timeout 2400 python tools/model_sweep.py --from-mjb tests/golden/sweep --device --nvmax 1600 --out "$OUT/sweep_gpu" > "$OUT/sweep_gpu.log" 2>&1 ) 2>&1 | grep real
      head -3 "$OUT/sweep_gpu/sweep.txt"; grep -v "^ok\|^#" "$OUT/sweep_gpu/sweep.txt" | head -40
      ( time timeout 2400 python tools/model_sweep.py --from-mjb tests/golden/sweep --device --nvmax 1600 --oracle parity --out "$OUT/sweep_gpu_glibc" > "$OUT/sweep_gpu_glibc.log" 2>&1 ) 2>&1 | grep real
      head -3 "$OUT/sweep_gpu_glibc/sweep.txt"; grep -v "^ok\|^#" "$OUT/sweep_gpu_glibc/sweep.txt" | grep -v rejected | head -20 ;;
    sweeplong)
      # the same sweep over a longer horizon (default 100 steps; sweeplong:<steps>), against the reference linked with the kernels' libm
      ( time timeout 3000 python tools/model_sweep.py --from-mjb tests/golden/sweep --device --nvmax 1600 --steps ${rest:-100} --out "$OUT/sweep_gpu_long" > "$OUT/sweep_gpu_long.log" 2>&1 ) 2>&1 | grep real
      head -3 "$OUT/sweep_gpu_long/sweep.txt"; grep -v "^ok\|^#" "$OUT/sweep_gpu_long/sweep.txt" | grep -v rejected | head -40 ;;
    flexab)
      bash tools/gpu_flex_ab.sh "gpurun_out/$TAG/flex_ab" ${args:-r04} 2>&1 | tee "$OUT/flex_ab.txt" ;;
    ablib)
      bash tools/gpu_ab_lib.sh "gpurun_out/$TAG/ab_${rest%.so}" "tools/variants/$rest" 3 2>&1 | tee "$OUT/ab_${rest%.so}.txt" ;;
    resources)
      python tools/kernel_resources.py > "$OUT/kernel_resource_usage.txt" 2>&1; cat "$OUT/kernel_resource_usage.txt" ;;
    *) echo "unknown step $step" ;;
  esac
done
