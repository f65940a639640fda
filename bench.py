#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched mj_step hot path (BASELINE.json: humanoid.xml, 4096
envs/GPU, PGS solver, Euler, fp64) on N MI355X of one node.

  python bench.py --gpus N --steps K --warmup W
  (N>1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

A "step" is one mj_step of every environment of the rank's batch.  Controls are synthetic
random actions U(ctrlrange) drawn once per (env, step) and resident in HBM before the timed region
(SURVEY.md 8d mode B); the state after every step is written to a device array
[nenv][K][nstate] (the rollout API's `state` output, worst-case I/O).  The K timed steps run as
launches of the rollout kernel of --chunk steps each (one wavefront per environment loops over
the steps of a launch), bracketed by barrier + device synchronisation; the time is the max over
ranks.  The W warm-up steps exercise EVERY code path of the timed region (launch with a state
output, the strided final-state copy, the gather), so nothing is loaded or compiled inside the
timer and wall time == kernel time at any --steps.  Environments shard across ranks with no
per-step exchange (weak scaling: 4096 envs on every GPU); the only collective is the end-of-run
gather of the final states to rank 0 over RCCL, inside the timed region.

Prints ONE JSON line (rank 0) with the throughput and
  roofline       the dominant (only) kernel against the HBM bound,
  parity_sample  64 of the timed environments re-stepped on the compiled reference (oracle/_ref),
                 every step of warm-up + timed region: max relative state error per step and the
                 integer counts (ncon, nefc, solver iterations) of the last step,
  testspeed_regime  the same kernel in the reference testspeed's control regime (OU-filtered
                 Halton noise, `sample/testspeed.cc:72-112`, after the humanoids have settled on
                 the floor: nefc ~ 46) so that CPU and GPU legs run the same contact regime,
  cpu_baseline   (N=1) the reference engine's own `testspeed` on all host cores, bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import re
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NENV_PER_GPU = 4096
HBM_PEAK_GBS = 8000.0

# --config: the BASELINE.json configurations that fit one GPU.  `humanoid` (configs[1]) is the metric;
# the others are additional lines (`python bench.py --config cube`), never the driver's default.
#   solver / integrator None = as the model file ships them; ctrl range and dt from the XML cited
CONFIGS = {
    "humanoid": dict(mjb="humanoid.mjb", xml="model/humanoid/humanoid.xml", nenv=4096, solver="pgs", integrator="euler",
                     ctrl=(-1.0, 1.0), dt=0.005, free_root=True,
                     metric="env-steps/sec on humanoid.xml, {nenv} envs/GPU"),
    # model/cube/cube_3x3x3.xml: Newton (default solver), implicitfast (:4), motors ctrlrange +-0.05 (:16), dt 0.01
    "cube": dict(mjb="cube_3x3x3.mjb", xml="model/cube/cube_3x3x3.xml", nenv=2048, solver=None, integrator=None,
                 ctrl=(-0.05, 0.05), dt=0.01, free_root=False, glibc_slack=0.01,
                 metric="env-steps/sec on cube_3x3x3.xml (convex mesh contacts, Newton), {nenv} envs/GPU"),
    # model/slider_crank/slider_crank.xml: position actuators ctrlrange +-0.1 (:10), default dt 0.002
    "slider_crank": dict(mjb="slider_crank.mjb", xml="model/slider_crank/slider_crank.xml", nenv=64, solver="pgs",
                         integrator="euler", ctrl=(-0.1, 0.1), dt=0.002, free_root=False,
                         metric="env-steps/sec on slider_crank.xml, {nenv} envs"),
    # model/flex/jelly.xml (BASELINE configs[4]: 1024 envs over 4 GPUs = 256 per GPU): 512-vertex solid flex, nv 1536, CG,
    # Euler, dt 1 ms, no actuators -- the environments differ by their initial vertex velocities.  The jelly reaches its
    # capsule after ~340 steps and has settled on it after ~1000 (mean nefc ~ 80, ~17 CG iterations per step): the
    # metric is timed in THAT regime (settle 1000 untimed steps); the nearly contact-free fall from the reset state is
    # reported separately as `free_fall_regime`, never under the metric's name
    "flex": dict(mjb="jelly.mjb", xml="model/flex/jelly.xml", nenv=256, solver=None, integrator=None,
                 ctrl=(0.0, 0.0), dt=0.001, free_root=False, settle=1000, warmup=20, solver_label="cg", integ_label="euler",
                 parity_envs=8, free_fall_steps=100,
                 metric="env-steps/sec on flex/jelly.xml (512-vertex solid flex, CG, settled contact regime), {nenv} envs/GPU"),
}


def initial_states(qpos0: np.ndarray, nv: int, nenv: int, seed: int, free_root: bool = True) -> np.ndarray:
    """reset state + hinge perturbation N(0,0.05^2), qvel ~ N(0,0.1^2), env-major draws.
    humanoid: qpos[0:7] is the free joint (left untouched), the rest are hinges.  Models whose
    joints are not all hinges (cube: ball joints, unit quaternions) keep qpos0 and draw qvel only."""
    rng = np.random.Generator(np.random.PCG64(seed))
    nq = qpos0.size
    s0 = np.zeros((nenv, 1 + nq + nv))
    for e in range(nenv):
        s0[e, 1:1 + nq] = qpos0
        if free_root:
            s0[e, 8:1 + nq] += rng.normal(0, 0.05, size=nq - 7)
        s0[e, 1 + nq:] = rng.normal(0, 0.1, size=nv)
    return s0


def halton(index: int, base: int) -> float:
    """mju_Halton (src/engine/engine_util_misc.c:2283)"""
    n0, f, hn = index, 1.0 / base, 0.0
    while n0 > 0:
        n1 = n0 // base
        hn += f * (n0 - n1 * base)
        f /= base
        n0 = n1
    return hn


def ctrl_noise(nstep: int, nu: int, dt: float, lo: np.ndarray, hi: np.ndarray,
               noise_std: float = 0.01, noise_rate: float = 0.1) -> np.ndarray:
    """CtrlNoise of the reference's sample/testspeed.cc:72-112 (all actuators ctrl-limited, no
    keyframe): an Ornstein-Uhlenbeck filtered Halton sequence around the range midpoint."""
    rate = np.exp(-dt / noise_rate)
    scale = noise_std * np.sqrt(1 - rate * rate)
    mid, half = 0.5 * (hi + lo), 0.5 * (hi - lo)
    out = np.zeros((nstep, nu))
    for t in range(nstep):
        for i in range(nu):
            c = rate * out[t - 1, i] + (1 - rate) * mid[i] if t > 0 else mid[i]
            c += scale * half[i] * (2 * halton(t, i + 2) - 1)
            out[t, i] = min(max(c, lo[i]), hi[i])
    return out


def cpu_baseline(nthread: int, budget_s: float = 15.0, mjb_name: str = "humanoid.mjb", solver_flag: str | None = "PGS") -> dict | None:
    """reference CPU engine timed by the reference's own sample/testspeed.cc (compiled from the
    reference sources into oracle/_ref by oracle/Makefile) on the host cores of this box."""
    exe = os.path.join(ROOT, "oracle", "_ref", "testspeed")
    mjb = os.path.join(ROOT, "tests", "golden", mjb_name)
    if not os.path.exists(exe):
        return None
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref"))
    extra = [f"--solver={solver_flag}"] if solver_flag else []

    def run(nstep):
        out = subprocess.run([exe, mjb, f"--nstep={nstep}", f"--nthread={nthread}", *extra],
                             capture_output=True, text=True, env=env, timeout=600).stdout
        m = re.search(r"Total steps per second\s*:\s*([0-9.]+)", out)
        it = re.search(r"(?:PGS|Newton|CG) iters / step\s*:\s*([0-9.]+)", out)
        nc = re.search(r"Contacts / step\s*:\s*([0-9.]+)", out)
        ne = re.search(r"Constraints / step\s*:\s*([0-9.]+)", out)
        g = lambda x: float(x.group(1)) if x else None
        return g(m), g(it), g(nc), g(ne)

    sps, *_ = run(2000)                          # calibration
    if not sps:
        return None
    nstep = int(max(2000, min(400000, budget_s * sps / nthread)))
    sps, iters, ncon, nefc = run(nstep)
    return {"value": sps, "unit": "env-steps/s", "cores": nthread, "kind": "reference",
            "mean_ncon": ncon, "mean_nefc": nefc, "mean_pgs_iter": iters,
            "sample": f"reference sample/testspeed.cc on liboracle_fast (-O3 -mavx), {mjb_name} {' '.join(extra)} "
                      f"--nthread={nthread} --nstep={nstep} (OU-Halton ctrl noise, its default regime; "
                      f"{iters} solver iters/step)"}


def cpu_rollout_leg(mjb_name, solver_id, integ_id, s0, ctrl, nthread, warm0=None, min_seconds=3.0):
    """The like-for-like CPU number for `value`: the reference engine (oracle/_ref/liboracle_fast.so) stepping
    the bench's OWN initial states and control stream -- rows [0, R) of the GPU batch, warm-up + timed steps
    -- on all host cores, through oracle/rollout_bench.cc (the work of _unsafe_rollout_threaded,
    python/mujoco/rollout.cc:181-216).  TEST INFRASTRUCTURE used as a baseline, never as the product."""
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "rollout_bench")
    if not os.path.exists(exe):
        return None
    # (the batch is repeated inside one thread team until the timed work lasts min_seconds: round 5's single 60 ms pass
    # read 0.53, 1.68 and 1.96 M env-steps/s in three runs of one build)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref"), ROLLOUT_BENCH_MIN_S=str(min_seconds))
    R, T = ctrl.shape[0], ctrl.shape[1]
    with tempfile.TemporaryDirectory() as td:
        np.ascontiguousarray(s0[:R], dtype=np.float64).tofile(os.path.join(td, "s0.bin"))
        np.ascontiguousarray(ctrl, dtype=np.float64).tofile(os.path.join(td, "ctrl.bin"))
        extra = []
        if warm0 is not None:
            np.ascontiguousarray(warm0[:R], dtype=np.float64).tofile(os.path.join(td, "warm0.bin"))
            extra = [os.path.join(td, "warm0.bin")]
        out = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", mjb_name), os.path.join(td, "s0.bin"),
                              os.path.join(td, "ctrl.bin"), str(R), str(T), str(nthread),
                              str(-1 if solver_id is None else solver_id), str(-1 if integ_id is None else integ_id),
                              os.path.join(td, "final.bin"), *extra], capture_output=True, text=True, env=env, timeout=900).stdout
        final = np.fromfile(os.path.join(td, "final.bin")) if os.path.exists(os.path.join(td, "final.bin")) else None
    kv = dict(x.split("=") for x in out.split() if "=" in x)
    if "env_steps_per_s" not in kv:
        return None
    return {"value": float(kv["env_steps_per_s"]), "unit": "env-steps/s", "cores": nthread, "kind": "reference",
            "rollouts": R, "nstep": T, "seconds": float(kv["seconds"]), "repeats": int(kv.get("repeats", 1)), "mean_ncon": float(kv["mean_ncon"]),
            "mean_nefc": float(kv["mean_nefc"]), "mean_solver_iter": float(kv["mean_niter"]),
            "sample": f"oracle/rollout_bench (liboracle_fast, -O3 -mavx): rows [0,{R}) of the GPU batch, the same state0 and "
                      f"U(ctrlrange) control stream, {T} steps (warm-up + timed) from "
                      + ("reset" if warm0 is None else "the GPU batch's settled state and warm start")
                      + f", {nthread} threads, batch repeated x{kv.get('repeats', 1)} = {kv['seconds']} s timed"}, final


def measured_traffic(steps_per_launch: int, nenv: int, model_xml: str = ""):
    """HBM bytes per launch from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same workload
    (tools/gpu_profile.sh -> profiles/<round>/pmc_summary*.txt; FETCH_SIZE x2 per the gfx950 correction of
    MI355X_MICROARCH.md).  Only the summaries of profiles/CURRENT -- the directory measured on the committed build --
    are consulted, for the same model and environment count; the counters are normalised per env-step and scaled to
    this run's launch, so a summary taken at another steps-per-launch still applies (the kernel's traffic is
    proportional to the steps it runs).  Returns (bytes per launch, source) or (None, None): never an older round's number."""
    import glob
    cur = os.path.join(ROOT, "profiles", "CURRENT")
    if not os.path.exists(cur):
        return None, None
    pref = os.path.join(ROOT, "profiles", open(cur).read().strip())
    best = (None, None)
    for f in sorted(glob.glob(os.path.join(pref, "pmc_summary*.txt"))):
        text = open(f).read()
        if model_xml and model_xml not in text:
            continue
        m = re.search(r"per launch \((\d+) steps x (\d+) envs\): read ([0-9.]+) MB raw / ([0-9.]+) MB with .*?written ([0-9.]+) MB", text)
        if m and int(m.group(2)) == nenv:
            per_env_step = (float(m.group(4)) + float(m.group(5))) * 1e6 / (int(m.group(1)) * nenv)
            exact = int(m.group(1)) == steps_per_launch
            if best[0] is None or exact:
                best = (per_env_step * steps_per_launch * nenv,
                        os.path.relpath(f, ROOT) + ("" if exact else f" (counters taken at {m.group(1)} steps per launch, scaled per env-step to this run's {steps_per_launch})"))
    return best


def measured_sq(config: str):
    """Instruction-level evidence for `roofline` from the committed SQ-counter summary of the rollout kernel
    (tools/gpu_sq.sh -> profiles/<CURRENT>/sq_summary_<config>.txt, regime uniform): vector / scalar instructions per
    env-step and the mean fraction of the 64 lanes a vector instruction has active.  None without a summary."""
    cur = os.path.join(ROOT, "profiles", "CURRENT")
    if not os.path.exists(cur):
        return None
    f = os.path.join(ROOT, "profiles", open(cur).read().strip(), "sq_summary_%s.txt" % config)
    if not os.path.exists(f):
        return None
    vals = {}
    take = False
    for line in open(f):
        if line.startswith("-- rollout kernel"):
            take = "regime uniform" in line
        elif line.startswith("=="):
            take = False
        elif take:
            m = re.match(r"\s+(SQ_[A-Z_]+)\s+([0-9.]+)", line)
            if m:
                vals[m.group(1)] = float(m.group(2))
    if "SQ_INSTS_VALU" not in vals:
        return None
    out = {"valu_insts_per_env_step": vals["SQ_INSTS_VALU"], "salu_insts_per_env_step": vals.get("SQ_INSTS_SALU"),
           "source": os.path.relpath(f, ROOT)}
    if "SQ_THREAD_CYCLES_VALU" in vals and vals.get("SQ_ACTIVE_INST_VALU"):
        out["active_lane_frac"] = vals["SQ_THREAD_CYCLES_VALU"] / vals["SQ_ACTIVE_INST_VALU"] / 64.0
    return out


def parity_sample(model_path, solver, integrator, s0, ctrl, gpu_state, envs, step_once, ctrl_range=(-1.0, 1.0), iter_exact=True,
                  warm0=None, glibc_slack=0.0):
    """Re-step the sampled environments on the compiled reference (TEST INFRASTRUCTURE, used here
    as the checker only).  ctrl / gpu_state: [len(envs)][T][...] host arrays of the whole run
    (warm-up + timed).  Two builds of the reference are consulted (oracle/Makefile):
      * `reference_glibc`: the reference as built, against the host's libm -- the oracle;
      * `reference_device_libm`: the same objects with sin / cos / atan2 / exp bound to the routines the kernels evaluate
        (mjh_sincos, mjh_atan2, mjh_exp; oracle/devmath_shim.cc).  glibc's and the kernels' are both ~1 ulp and differ in the
        last bit for a small fraction of arguments; on a stiff, geometrically degenerate model (cube_3x3x3: aligned
        cubelet faces; EPA and face clipping are discontinuous in the poses) one such bit in a hinge's quaternion can
        move a contact point by millimetres, i.e. the next state by far more than 1e-6.  Against this build such
        steps agree again, which separates the platform's libm from the algorithm.
    Per build:
      (1) trajectory: the oracle is stepped from s0 with its own warm start; after every step its state
          is compared with the GPU's and then re-synchronised to it (per-step error, not a chaotic accumulation);
      (2) identical inputs, EVERY step of the run: each (state, the oracle's warm start, control) of (1) is
          handed to the GPU for one mj_step (`step_once`, one batch of len(envs)*T environments); the next
          state must agree within 1e-6 and contact count, constraint count and solver iteration count exactly.
    `ok`: (2) holds for every step against the device-libm build AND against the glibc build -- except for the cube
    (glibc_slack = 0.01: a model whose own reference moves by more than 1e-6 under a one-ulp change of the state on ~1 % of
    its steps, tests/test_oracle_golden.py::test_cube_contact_discontinuity), where 99 % of the steps, floats and counts,
    have to agree with glibc."""
    try:
        from oracle import refbind as rb
        if not rb.available():
            return None
    except Exception:
        return None
    spec = rb.mjSTATE_FULLPHYSICS
    T = ctrl.shape[1]
    out = {"envs": len(envs), "steps_checked": T, "tolerance": 1e-6}
    for label, kind in (("reference_glibc", "parity"), ("reference_device_libm", "devmath")):
        if not rb.available(kind):
            continue
        m = rb.MjModel.from_binary_path(model_path, kind=kind)
        if solver is not None:
            m.opt.solver = solver
        if integrator is not None:
            m.opt.integrator = integrator
        worst, worst_at = 0.0, None
        pre_s, pre_w, pre_u, nxt, ints = [], [], [], [], []
        for k, e in enumerate(envs):
            d = rb.MjData(m)
            rb.mj_setState(m, d, s0[k], spec)
            if warm0 is not None:
                d.qacc_warmstart[:] = warm0[k]     # (run starts from a settled GPU state: its warm start comes along)
            for t in range(T):
                pre_s.append(rb.mj_getState(m, d, spec)); pre_w.append(np.array(d.qacc_warmstart)); pre_u.append(ctrl[k, t])
                d.ctrl[:] = ctrl[k, t]
                rb.mj_step(m, d)
                ref = rb.mj_getState(m, d, spec)
                nxt.append(ref); ints.append((int(d.ncon), int(d.nefc), int(d.solver_niter[0])))
                got = gpu_state[k, t]
                err = float(np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))))
                if not np.isfinite(err):
                    err = float("inf")
                if err > worst:
                    worst, worst_at = err, (int(e), t)
                rb.mj_setState(m, d, got, spec)
        # (2) every step again from identical (state, warm start, control)
        got, counts = step_once(np.array(pre_s), np.array(pre_w), np.array(pre_u)[:, None])
        nxt, ints = np.array(nxt), np.array(ints)
        per_step = np.max(np.abs(got - nxt) / np.maximum(1.0, np.abs(nxt)), axis=1)
        # (iter_exact False -- the flex configuration, whose 1536-dof island is summed in a different order than the
        # reference's island-compressed sparse routines: an iteration count may differ by one on a step whose last
        # improvement sits at the solver tolerance; reported, not counted as a mismatch)
        nbad = int(np.sum((counts[:, 0] != ints[:, 0]) | (counts[:, 1] != ints[:, 1]) |
                          ((counts[:, 5] != ints[:, 2]) if iter_exact else (np.abs(counts[:, 5] - ints[:, 2]) > 1))))
        niter_diff = int(np.sum(counts[:, 5] != ints[:, 2]))
        out[label] = {
            "identical_input_steps": {"steps": int(len(nxt)), "max_rel_err": float(per_step.max()),
                                      "frac_within_tolerance": float(np.mean(per_step <= 1e-6)),
                                      "bit_exact_steps": int(np.sum(np.all(got == nxt, axis=1))), "count_mismatches": nbad,
                                      "solver_iter_differs": niter_diff,
                                      "checked": "next state within 1e-6; ncon, nefc, solver_niter exact" if iter_exact else
                                                 "next state within 1e-6; ncon, nefc exact; solver_niter within one",
                                      "mean_ncon": float(ints[:, 0].mean()), "mean_nefc": float(ints[:, 1].mean()),
                                      "mean_solver_iter": float(ints[:, 2].mean())},
            "trajectory": {"max_rel_err": worst, "worst_env_step": worst_at, "within_tolerance": bool(worst <= 1e-6)}}
    g = out.get("reference_glibc", {}).get("identical_input_steps")
    dmath = out.get("reference_device_libm", {}).get("identical_input_steps")
    if g is not None and dmath is not None:
        # (against glibc a step may differ in its last bit of sin / cos, which on ~1 % of the cube's steps moves a degenerate
        # contact -- floats and counts alike, tests/test_oracle_golden.py::test_cube_contact_discontinuity: the same 99 % gate
        # for both; against the device-libm build everything is exact)
        # glibc_slack: the fraction of steps allowed to differ from the glibc build -- 0 except for the cube (0.01)
        out["ok"] = bool(dmath["max_rel_err"] <= 1e-6 and dmath["count_mismatches"] == 0 and
                         g["frac_within_tolerance"] >= 1.0 - glibc_slack and g["count_mismatches"] <= glibc_slack*g["steps"])
        out["glibc_slack"] = glibc_slack
    elif g is not None:
        out["ok"] = bool(g["max_rel_err"] <= 1e-6 and g["count_mismatches"] == 0)
    out["protocol"] = ("oracle/_ref mj_step from the same state0/controls over warm-up + timed region, re-synchronised to the GPU state "
                       "after every step (trajectory); then every one of those steps re-run on the GPU from identical (state, warm "
                       "start, control) with exact integer observables (identical_input_steps).  reference_glibc = the reference as "
                       "built; reference_device_libm = the same objects with sin / cos / atan2 / exp bound to the kernels' own routines "
                       "(oracle/devmath_shim.cc): ok needs every step within 1e-6 and every count exact against the latter, and >= 99 % of the steps (floats and counts) against the former")
    return out


def api_regime(model_path, solver_id, integ_id, nenv, nstep, s0, cfg, batch, stream, dev):
    """The drop-in entry point as a caller uses it: `mujoco_amd.rollout.rollout(model, data, initial_state, control,
    state=...)` with numpy arrays (python/mujoco/rollout.py's signature; underneath `mjhip_rollout`, include/mjhip.h:192 =
    `_unsafe_rollout`'s contract, rollout.cc:74-178): controls go host -> device and the per-step states come back inside the
    timer: one launch, the controls copied up before it and the states down after it (launching in chunks with overlapped
    copies measured slower -- profiles/r05/api_rate.txt, mjh_runtime.h: rollout_impl, $MJHIP_ROLLOUT_CHUNK).
    Next to it the same work device-resident (one launch of `nstep` steps from the same state0 / controls), the rate
    `value` is the ceiling of.  mjModel / mjData are the CALLER's objects: here made by the compiled reference standing in
    for the caller's MuJoCo (as in tests/); the product only reads mjModel and writes the last state into mjData."""
    import torch
    import mujoco_amd as ma
    from mujoco_amd import rollout as ro
    from oracle import refbind as rb          # the caller's MuJoCo (mj_loadModel / mj_makeData), not part of the timed path
    if not rb.available():
        return {"error": "no MuJoCo library on this box to make the caller's mjModel / mjData"}
    m = rb.MjModel.from_binary_path(model_path)
    if solver_id is not None:
        m.opt.solver = solver_id
    if integ_id is not None:
        m.opt.integrator = integ_id
    d = rb.MjData(m)
    rng = np.random.Generator(np.random.PCG64(97))
    nu = m.nu
    ctrl = rng.uniform(cfg["ctrl"][0], cfg["ctrl"][1], size=(nenv, nstep, nu))
    state = np.empty((nenv, nstep, s0.shape[1]))
    state[:] = 0                                                   # (touch the pages: an RL loop reuses its output array)
    ro.rollout(m, d, s0, ctrl, state=state)                        # warm-up: device model / batch cache, staging buffers
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        out, _ = ro.rollout(m, d, s0, ctrl, state=state)
        times.append(time.perf_counter() - t0)
    t_api = min(times)
    # the same rollout device-resident: arrays already in HBM, one launch
    cd = torch.from_numpy(ctrl).to(dev); sd = torch.from_numpy(s0).to(dev)
    od = torch.empty((nenv, nstep, s0.shape[1]), dtype=torch.float64, device=dev)
    tdev = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        batch.rollout_device(nstep, ma.mjSTATE_CTRL, sd.data_ptr(), 0, cd.data_ptr(), od.data_ptr(), stream, cont=False)
        torch.cuda.synchronize()
        tdev.append(time.perf_counter() - t0)
    t_dev = min(tdev)
    same = bool(np.array_equal(od.cpu().numpy(), out))
    return {"value": nenv*nstep/t_api, "unit": "env-steps/s", "envs": nenv, "steps": nstep, "seconds": t_api, "all_seconds": times,
            "device_resident_value": nenv*nstep/t_dev, "ratio_to_device_resident": t_dev/t_api,
            "host_mb_in": (ctrl.nbytes + s0.nbytes)/1e6, "host_mb_out": state.nbytes/1e6, "identical_to_device_resident": same,
            "entry_point": "mujoco_amd.rollout.rollout(model, data, initial_state, control, state=preallocated) -> mjhip_rollout",
            "note": "H2D of the controls and D2H of every step's state inside the timer (pageable numpy arrays, one launch in between)"}


LEGS = {
    # name: bench arguments of the sub-run (BASELINE configs[3], [4] per GPU, [0])
    "cube": ["--config", "cube", "--steps", "100", "--warmup", "20", "--parity-envs", "16"],
    "flex": ["--config", "flex", "--steps", "200"],
    "slider_crank": ["--config", "slider_crank", "--steps", "1000", "--warmup", "100"],
}


def config_legs() -> dict:
    """BASELINE configs 4 (cube), 5 (flex, one GPU's 256 environments) and 1 (slider-crank) as short sub-runs of this
    script, so that the driver's default command measures them too: each leg is the complete line of
    `python bench.py --config <name> ...` (value, roofline, cpu_baseline with the like-for-like rollout leg,
    parity_sample), with the CPU legs bounded to a few seconds."""
    out = {}
    for name, argv in LEGS.items():
        t0 = time.perf_counter()
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), *argv, "--leg"], capture_output=True, text=True,
                               timeout=420, cwd=ROOT)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            leg = json.loads(lines[-1]) if lines else {"error": (r.stderr or r.stdout)[-800:], "returncode": r.returncode}
        except Exception as exc:      # a failing leg is reported in place, the metric line survives
            leg = {"error": repr(exc)}
        leg["leg_wall_s"] = time.perf_counter() - t0
        leg["command"] = "python bench.py " + " ".join(argv)
        out[name] = leg
    return out


LINE_LIMIT = 8192        # bytes of the driver's line (round 5's 20 KB line was not parsed: BENCH_r05.json "parsed": null)


def _finite(x):
    """the record with every non-finite float replaced by None (strict JSON: no NaN / Infinity tokens)"""
    if isinstance(x, dict):
        return {str(k): _finite(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_finite(v) for v in x]
    if isinstance(x, (float, np.floating)):
        return float(x) if np.isfinite(x) else None
    if isinstance(x, (np.integer,)):
        return int(x)
    if isinstance(x, (np.bool_,)):
        return bool(x)
    return x


def _scalars(d: dict, keys, clip: int = 160) -> dict:
    """d restricted to `keys`, scalar values only, strings clipped"""
    out = {}
    for k in keys:
        v = (d or {}).get(k)
        if isinstance(v, str):
            out[k] = v if len(v) <= clip else v[:clip - 1] + "~"
        elif isinstance(v, (bool, int, float)) or (v is None and k in ("vs_baseline", "traffic")):
            out[k] = v
    return out


def driver_line(res: dict, full_path: str | None) -> dict:
    """The ONE line the driver parses, built from the full record: the contract's keys, `config` with scalar values only,
    `roofline` and `cpu_baseline` as flat objects, the other legs' headline figures as top-level scalars (leg_<name>_*),
    and the path of the full record.  Kept under LINE_LIMIT bytes -- tests/test_bench_line.py holds it to that."""
    res = _finite(res)
    line = _scalars(res, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                          "vs_baseline", "dtype", "data"), clip=200)
    line["config"] = _scalars(res.get("config"), ("workload", "envs_per_gpu", "nstep", "solver", "integrator", "ctrl", "settle",
                                                  "parallelism", "kernel_variant", "layout", "mapping"), clip=200)
    line["roofline"] = _scalars(res.get("roofline"), ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "kernel",
                                                      "steps_per_launch", "launch_ms", "kernel_ms_total", "wall_ms_total",
                                                      "algorithmic_bytes_per_env_step", "algorithmic_bytes_per_launch",
                                                      "valu_insts_per_env_step", "salu_insts_per_env_step", "active_lane_frac"), clip=120)
    cb = res.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = _scalars(cb, ("value", "unit", "cores", "kind", "sample", "seconds", "repeats", "mean_ncon", "mean_nefc",
                                             "mean_solver_iter", "mean_pgs_iter"), clip=260)
        rr, ts = cb.get("rollout_regime") or {}, cb.get("testspeed_regime") or {}
        for k in ("seconds", "repeats", "rollouts", "nstep", "mean_ncon", "mean_nefc", "mean_solver_iter"):
            if k in rr and k not in line["cpu_baseline"]:
                line["cpu_baseline"][k] = rr[k]
        if ts.get("value") is not None:
            line["cpu_baseline"]["testspeed_value"] = ts["value"]
            line["cpu_baseline"]["testspeed_mean_nefc"] = ts.get("mean_nefc")
            line["cpu_baseline"]["testspeed_iters"] = ts.get("mean_pgs_iter")
    es = res.get("end_state") or {}
    line.update({"end_" + k: es[k] for k in ("warnings", "mean_ncon", "mean_nefc", "mean_solver_iter") if k in es})
    ps = res.get("parity_sample")
    if ps is not None:
        line["parity_ok"] = ps.get("ok")
        line["parity_steps_checked"] = ps.get("steps_checked")
        line["parity_envs"] = ps.get("envs")
        for lab, short in (("reference_glibc", "glibc"), ("reference_device_libm", "devlibm")):
            ii = (ps.get(lab) or {}).get("identical_input_steps")
            if ii:
                line[f"parity_{short}_max_rel_err"] = ii.get("max_rel_err")
                line[f"parity_{short}_bit_exact_steps"] = ii.get("bit_exact_steps")
                line[f"parity_{short}_steps"] = ii.get("steps")
                line[f"parity_{short}_count_mismatches"] = ii.get("count_mismatches")
        if "error" in ps:
            line["parity_error"] = str(ps["error"])[:200]
    for name in ("testspeed_regime", "newton_regime", "free_fall_regime"):
        r = res.get(name)
        if r and r.get("value") is not None:
            line[name + "_value"] = r["value"]
            e = r.get("end_state") or {}
            if "mean_nefc" in e:
                line[name + "_mean_nefc"] = e["mean_nefc"]
            it = e.get("mean_pgs_iter", e.get("mean_solver_iter"))
            if it is not None:
                line[name + "_iters"] = it
            if (r.get("cpu_testspeed") or {}).get("value") is not None:
                line[name + "_cpu_testspeed_value"] = r["cpu_testspeed"]["value"]
    pr = res.get("pgs_residual")
    if pr:
        if "value" in pr:
            line["pgs_residual_value"] = pr["value"]
            pps = pr.get("parity_sample") or {}
            line["pgs_residual_parity_ok"] = pps.get("ok")
            ii = (pps.get("reference_device_libm") or {}).get("identical_input_steps") or {}
            line["pgs_residual_max_rel_err"] = ii.get("max_rel_err")
            line["pgs_residual_iter_differs"] = ii.get("solver_iter_differs")
            line["pgs_residual_steps"] = ii.get("steps")
            if (pr.get("testspeed_regime") or {}).get("value") is not None:
                line["pgs_residual_testspeed_regime_value"] = pr["testspeed_regime"]["value"]
        else:
            line["pgs_residual_error"] = str(pr.get("error"))[:200]
    ar = res.get("api_regime")
    if ar:
        if "value" in ar:
            line["api_regime_value"] = ar["value"]
            line["api_regime_ratio_to_device_resident"] = ar.get("ratio_to_device_resident")
            line["api_regime_identical"] = ar.get("identical_to_device_resident")
        else:
            line["api_regime_error"] = str(ar.get("error"))[:200]
    for name, leg in (res.get("configs") or {}).items():
        p = f"leg_{name}_"
        if "error" in leg:
            line[p + "error"] = str(leg["error"])[-200:]
            continue
        rl, lcb, lps = leg.get("roofline") or {}, leg.get("cpu_baseline") or {}, leg.get("parity_sample") or {}
        line[p + "value"] = leg.get("value")
        line[p + "envs"] = (leg.get("config") or {}).get("envs_per_gpu")
        line[p + "steps"] = leg.get("steps")
        line[p + "roofline_frac"] = rl.get("frac")
        line[p + "traffic"] = rl.get("traffic")
        line[p + "kernel"] = rl.get("kernel")
        line[p + "parity_ok"] = lps.get("ok")
        line[p + "cpu_like_for_like"] = lcb.get("value")
        line[p + "cpu_seconds"] = (lcb.get("rollout_regime") or {}).get("seconds")
        line[p + "wall_s"] = leg.get("leg_wall_s")
    if "configs_ok" in res:
        line["configs_ok"] = res["configs_ok"]
    if full_path:
        line["full_record"] = full_path
    # the size gate: drop the optional detail, longest first, until the line fits
    order = [k for k in line if k.startswith("leg_") and k.endswith(("_kernel", "_wall_s", "_cpu_seconds", "_steps", "_envs"))] + \
            [k for k in line if k.startswith("parity_") and k not in ("parity_ok",)] + \
            [k for k in reversed(list(line)) if k.startswith("leg_")]
    while len(json.dumps(line, allow_nan=False)) >= LINE_LIMIT and order:
        line.pop(order.pop(0), None)
    return line


def emit(res: dict, config: str) -> None:
    """write the full record to a side file (gpurun_out/, else the system's temp dir) and print the driver's line"""
    import tempfile
    res = _finite(res)
    full_path = None
    for d in (os.path.join(ROOT, "gpurun_out"), tempfile.gettempdir()):
        try:
            os.makedirs(d, exist_ok=True)
            p = os.path.join(d, f"bench_full_{config}_n{res.get('n_gpus', 1)}.json")
            with open(p, "w") as f:
                json.dump(res, f, indent=1, allow_nan=False)
            full_path = os.path.relpath(p, ROOT) if p.startswith(ROOT) else p
            break
        except OSError:
            continue
    sys.stdout.flush()
    print(json.dumps(driver_line(res, full_path), allow_nan=False), flush=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default 100; flex: 400, the fall onto the capsule)")
    ap.add_argument("--chunk", type=int, default=250,
                    help="steps per rollout-kernel launch (one open-loop rollout call); 250 measured best: longer launches average the per-environment cost, shorter ones re-deal the environments over the SIMDs more often")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="humanoid",
                    help="humanoid = BASELINE configs[1], the metric; cube = configs[3] (many-contact convex meshes, Newton); "
                         "slider_crank = configs[0]")
    ap.add_argument("--envs-per-gpu", type=int, default=0, help="0: the configuration's own count (humanoid 4096, cube 2048, slider_crank 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-state-output", action="store_true", help="do not write the per-step state array")
    ap.add_argument("--solver", choices=["pgs", "newton", "cg"], default=None,
                    help="default: the configuration's (humanoid: pgs = BASELINE config 2, the metric; cube: as shipped = newton)")
    ap.add_argument("--integrator", choices=["euler", "rk4", "implicitfast"], default=None)
    ap.add_argument("--ctrl", choices=["uniform", "ou-halton"], default="uniform",
                    help="uniform = U(ctrlrange) per (env, step) (SURVEY 8d mode B, the metric); ou-halton = "
                         "testspeed's CtrlNoise sequence shared by all envs (mode A)")
    ap.add_argument("--settle", type=int, default=None, help="untimed steps before the warm-up (default 0; flex: 1000, the settled contact regime)")
    ap.add_argument("--no-legs", action="store_true",
                    help="humanoid, N = 1: do not append the `configs` legs (cube, flex, slider_crank; one short sub-run each)")
    ap.add_argument("--leg", action="store_true", help="(internal) this run is a `configs` leg of the default run: short CPU legs")
    ap.add_argument("--no-extra", action="store_true",
                    help="only the timed region (profiling runs): no parity sample, no testspeed-regime leg, no CPU baseline")
    ap.add_argument("--no-pgs-residual", action="store_true", help="skip the pgs_residual leg (the metric's workload with the opt-in tolerance-parity PGS sweep)")
    ap.add_argument("--no-newton-regime", action="store_true", help="skip the newton_regime leg (the testspeed regime with the model's own solver)")
    ap.add_argument("--regime-steps", type=int, default=200, help="timed steps of the testspeed-regime leg")
    ap.add_argument("--regime-settle", type=int, default=1000, help="untimed settling steps of that leg")
    ap.add_argument("--parity-envs", type=int, default=64)
    ap.add_argument("--api-steps", type=int, default=250,
                    help="steps of the api_regime leg: mujoco_amd.rollout.rollout with numpy arrays in and out (0: skip)")
    ap.add_argument("--gather", choices=["per-chunk", "final"], default="per-chunk",
                    help="N > 1: per-chunk = every launch's per-step state array goes to rank 0 over RCCL, overlapped with the "
                         "next launch (north star's observation gather; the default); final = only the end-of-run final states")
    args = ap.parse_args()

    import torch
    import mujoco_amd as ma
    from mujoco_amd.sharding import ChunkGather, gather_to_rank0

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the mjhip path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    lib = ma.lib()
    cfg = CONFIGS[args.config]
    model_path = os.path.join(ROOT, "tests", "golden", cfg["mjb"])
    model = ma.MjbModel(lib, model_path)
    solver_name = args.solver or cfg["solver"]
    integ_name = args.integrator or cfg["integrator"]
    solver_id = {"pgs": 0, "cg": 1, "newton": 2}[solver_name] if solver_name else None
    integ_id = {"euler": 0, "rk4": 1, "implicitfast": 3}[integ_name] if integ_name else None
    if solver_id is not None:
        model.set_option("solver", solver_id)     # PGS = BASELINE config 2
    if integ_id is not None:
        model.set_option("integrator", integ_id)
    solver_name = solver_name or cfg.get("solver_label", "newton")            # (cube_3x3x3.xml ships the default solver)
    integ_name = integ_name or cfg.get("integ_label", "implicitfast")        # (cube_3x3x3.xml:4)
    if args.warmup is None:
        args.warmup = cfg.get("warmup", 100)
    if args.settle is None:
        args.settle = cfg.get("settle", 0)
    if "parity_envs" in cfg:
        args.parity_envs = min(args.parity_envs, cfg["parity_envs"])
    dm = ma.DeviceModel(lib, model)
    nenv, K, W = (args.envs_per_gpu or cfg["nenv"]), args.steps, args.warmup
    nq, nv, nu, nstate = dm.nq, dm.nv, dm.nu, dm.nstate
    batch = ma.Batch(dm, nenv, device=local_rank)
    qpos0 = batch.get("qpos")[0]
    dev = torch.device("cuda", local_rank)
    stream = torch.cuda.current_stream().cuda_stream
    lo, hi = cfg["ctrl"][0]*np.ones(nu), cfg["ctrl"][1]*np.ones(nu)       # ctrlrange of the model's actuators
    dt = cfg["dt"]

    C = max(1, min(args.chunk, K))

    def chunks(n):
        return [C] * (n // C) + ([n % C] if n % C else [])

    def barrier():
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    events = []
    cur = {"batch": batch}                    # the batch the launches step (the newton_regime leg swaps in its own)
    obs = ChunkGather(rank, world, dist)      # per-chunk observation gather to rank 0 (no-op at N = 1)

    def launch(ctrl, out, state0=None):
        """one rollout-kernel launch of ctrl.shape[1] steps; state0 given: load the initial states,
        else continue from the batch's own state, warm start and warning counters"""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        cur["batch"].rollout_device(ctrl.shape[1], ma.mjSTATE_CTRL, 0 if state0 is None else state0.data_ptr(),
                             0, ctrl.data_ptr(), 0 if out is None else out.data_ptr(), stream,
                             cont=state0 is None)
        e1.record()
        events.append((e0, e1, ctrl.shape[1]))

    snap = {}
    # rows of the batch the CPU leg re-steps: as many as ~2 M env-steps allow (all 4096 at the driver's 25 steps -- a leg of
    # 1024 rollouts x 25 steps lasted 12 ms, thread start-up included), never fewer than 4 per host thread
    cpu_rows = min(nenv, max(1024, 4*(os.cpu_count() or 1), 2_000_000 // max(1, W + K)))
    host_ctrl = []                                                   # their control stream, in launch order

    def make_controls(kind, crng, sizes, t_begin):
        """device control arrays [nenv][c][nu] for consecutive launches of the given sizes"""
        out, t = [], t_begin
        if kind == "uniform":
            for c in sizes:
                u = crng.uniform(cfg["ctrl"][0], cfg["ctrl"][1], size=(nenv, c, nu))
                host_ctrl.append(u[:cpu_rows].copy())
                out.append(torch.from_numpy(u).to(dev))
        else:
            seq = ctrl_noise(t_begin + sum(sizes), nu, dt, lo, hi)
            for c in sizes:
                out.append(torch.from_numpy(np.ascontiguousarray(seq[t:t + c])).to(dev)[None].expand(nenv, c, nu).contiguous())
                t += c
        return out

    def timed_region(state0, ctrl_settle, ctrl_w, ctrl_k, want_state):
        """settle (untimed, no output) -> W warm-up steps through every code path of the timed
        region -> barrier -> K timed steps -> final-state copy (+ gather) -> barrier"""
        del events[:]
        first = state0
        snap.clear()
        for i, c in enumerate(ctrl_settle):
            last = i == len(ctrl_settle) - 1
            so = torch.empty((nenv, c.shape[1], nstate), dtype=torch.float64, device=dev) if last else None
            launch(c, so, first)
            first = None
            if last:
                # the settled state and warm start the measured steps begin from: what the parity sample and the CPU
                # rollout leg start from as well (taken before the warm-up, outside the timed region)
                torch.cuda.synchronize()
                snap["state"] = so[:, -1].cpu().numpy().copy()
                snap["warm"] = cur["batch"].get("qacc_warmstart").copy()
                del so
        state_w = [torch.empty((nenv, c.shape[1], nstate), dtype=torch.float64, device=dev) if want_state else None
                   for c in ctrl_w]
        state_k = [torch.empty((nenv, c.shape[1], nstate), dtype=torch.float64, device=dev) if want_state else None
                   for c in ctrl_k]
        final = torch.empty((nenv, nstate), dtype=torch.float64, device=dev)

        def finish(last):
            if last is not None:
                final.copy_(last[:, -1])
            if dist:
                gather_to_rank0(final, rank, world, dist)  # end-of-run state gather (RCCL over xGMI)

        for c, o in zip(ctrl_w, state_w):
            launch(c, o, first)
            if o is not None and args.gather == "per-chunk":
                obs.submit(o)          # (warm-up: the gather path runs once before the timer starts)
            first = None
        obs.wait()
        if not ctrl_w:
            # no warm-up steps asked for: still load the copy / gather code paths (zero simulation steps)
            finish(torch.zeros((nenv, 1, nstate), dtype=torch.float64, device=dev) if want_state else None)
        else:
            finish(state_w[-1])
        barrier()
        n_warm = len(events)
        obs.chunks = obs.bytes_sent = 0
        t0 = time.perf_counter()
        for c, o in zip(ctrl_k, state_k):
            launch(c, o, first)
            if o is not None and args.gather == "per-chunk":
                obs.submit(o)          # observation gather of this chunk, overlapped with the next launch
            first = None
        obs.wait()
        finish(state_k[-1] if state_k else None)
        barrier()
        elapsed = time.perf_counter() - t0
        timed = [(a.elapsed_time(b), n) for a, b, n in events[n_warm:]]
        return elapsed, timed, state_w, state_k

    # ---------------- the metric: random actions U(ctrlrange), BASELINE config 2 -----------------
    s0 = initial_states(qpos0, nv, nenv, seed=1234 + rank, free_root=cfg["free_root"])
    state0 = torch.from_numpy(s0).to(dev)
    crng = np.random.Generator(np.random.PCG64(4321 + rank))
    want_state = not args.no_state_output
    ctrl_s = make_controls(args.ctrl, crng, chunks(args.settle) if args.settle else [], 0)
    n_settle_ctrl = len(host_ctrl)
    ctrl_w = make_controls(args.ctrl, crng, chunks(W) if W else [], args.settle)
    ctrl_k = make_controls(args.ctrl, crng, chunks(K), args.settle + W)
    n_metric_ctrl = len(host_ctrl)          # control chunks of the metric leg (warm-up + timed), in order
    elapsed, timed, state_w, state_k = timed_region(state0, ctrl_s, ctrl_w, ctrl_k, want_state)
    snap_metric = dict(snap)   # (later regions take their own snapshots)
    kernel_ms = sum(t for t, _ in timed)
    launch_ms_timed = float(np.mean([t for t, n in timed if n == C])) if any(n == C for _, n in timed) else kernel_ms
    if dist:
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])
    warn = int(batch.get("warning").sum())
    counts = batch.get("counts")

    res = None
    if rank == 0:
        total_env_steps = nenv * world * K
        value = total_env_steps / elapsed
        # algorithmic HBM bytes per env-step: state read + state written
        # (2*nstate), control read (ncontrol), warmstart read + written (2*nv), 8 bytes each
        bytes_per_env_step = 8 * (2 * nstate + nu + 2 * nv)
        # one launch = C steps of nenv envs; achieved = algorithmic bytes per launch / avg launch time
        achieved = bytes_per_env_step * nenv * C / (launch_ms_timed * 1e-3) / 1e9
        traffic, traffic_src = measured_traffic(C, nenv, cfg["xml"])
        res = {
            "metric": cfg["metric"].format(nenv=nenv),
            "value": value,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed * 1e3 / K,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{cfg['xml']}, {nenv} envs/GPU, {solver_name.upper()} solver, {integ_name}, fp64, "
                                   + ("random actions U(ctrlrange)" if args.ctrl == "uniform" else "testspeed OU-Halton ctrl noise")
                                   + ", per-step state output" + ("" if want_state else " disabled"),
                       "envs_per_gpu": nenv, "nstep": K, "solver": solver_name.upper(), "integrator": integ_name,
                       "ctrl": args.ctrl, "settle": args.settle,
                       "parallelism": f"env-sharded x{world}",
                       "mapping": batch.lds_report().splitlines()[0] if batch.lds_report() else "no LDS plan",
                       "kernel_variant": batch.kernel_variant() if hasattr(batch, "kernel_variant") else "generic",
                       "layout": os.environ.get("MJHIP_LAYOUT", "aos")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         **(measured_sq(args.config) or {}),
                         "kernel": batch.kernel_name(),
                         "steps_per_launch": C,
                         "launch_ms": launch_ms_timed, "kernel_ms_total": kernel_ms,
                         "wall_ms_total": elapsed * 1e3,
                         "algorithmic_bytes_per_env_step": bytes_per_env_step,
                         "algorithmic_bytes_per_launch": bytes_per_env_step * nenv * C},
            "gather": {"mode": args.gather if world > 1 else "none (1 GPU)", "chunks": obs.chunks,
                       "bytes_per_rank": obs.bytes_sent,
                       "note": "per-step state arrays of every rank gathered on rank 0 after each launch (RCCL, async, overlapped "
                               "with the next launch), inside the timed region: its exposed cost is wall_ms_total - kernel_ms_total"},
            "end_state": {"warnings": warn, "mean_ncon": float(counts[:, 0].mean()),
                          "mean_nefc": float(counts[:, 1].mean()), "mean_solver_iter": float(counts[:, 5].mean())},
        }

    # ---------------- parity of the timed workload against the compiled reference ----------------
    if rank == 0 and not args.no_extra and want_state and (args.settle == 0 or snap_metric) and args.parity_envs > 0:
        envs = np.unique(np.linspace(0, nenv - 1, min(args.parity_envs, nenv)).astype(int))
        idx = torch.from_numpy(envs).to(dev)
        gs = torch.cat([x.index_select(0, idx) for x in state_w + state_k], dim=1).cpu().numpy()
        cs = torch.cat([x.index_select(0, idx) for x in ctrl_w + ctrl_k], dim=1).cpu().numpy()
        def step_once(states, warm, u):
            # (in slices of at most `nenv` environments: a batch reserves the reference's arena -- up to 16 MiB of
            # constraint rows -- per environment, and the sample at 500 steps is 38400 environments)
            outs, cnts = [], []
            for a in range(0, len(states), nenv):
                b = min(len(states), a + nenv)
                small = ma.Batch(dm, b - a, device=local_rank)
                outs.append(small.rollout_host(1, ma.mjSTATE_CTRL, states[a:b], warm[a:b], u[a:b])[:, 0])
                cnts.append(small.get("counts"))
                small.close()
            return np.concatenate(outs), np.concatenate(cnts)

        try:
            res["parity_sample"] = parity_sample(model_path, solver_id, integ_id, snap_metric["state"][envs] if snap_metric else s0[envs], cs, gs, envs,
                                                 step_once, cfg["ctrl"], iter_exact=cfg.get("iter_exact", True),
                                                 warm0=snap_metric["warm"][envs] if snap_metric else None,
                                                 glibc_slack=cfg.get("glibc_slack", 0.0))
            if snap_metric:
                res["parity_sample"]["start"] = f"the GPU batch's state and warm start after the {args.settle} settle steps"
        except Exception as exc:  # the bench line must survive a checker problem; it is reported, not hidden
            res["parity_sample"] = {"ok": False, "error": repr(exc)}
    del state_w, state_k

    # ---------------- the metric's workload once more with the OPT-IN residual-update PGS sweep ----------------
    # (mjhip_batch_set_pgs_mode(1), mjh_solver.h: solve_pgs_resid -- tolerance parity instead of bit parity: reported beside
    # `value`, never as it; its own parity sample with the 1e-6 bar, exact contact / constraint counts, solver_niter within one)
    if (not args.no_extra and not args.no_pgs_residual and args.config == "humanoid" and solver_name == "pgs" and args.settle == 0
            and hasattr(batch, "set_pgs_mode")):
        try:
            cur["batch"] = ma.Batch(dm, nenv, device=local_rank)
            cur["batch"].set_pgs_mode(1)
            el_r, timed_r, sw_r, sk_r = timed_region(state0, [], ctrl_w, ctrl_k, want_state)
            k_r = sum(t for t, _ in timed_r)
            if dist:
                t = torch.tensor([el_r, k_r], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el_r, k_r = float(t[0]), float(t[1])
            cn = cur["batch"].get("counts")
            if rank == 0:
                res["pgs_residual"] = {
                    "value": nenv * world * K / el_r, "unit": "env-steps/s", "steps": K, "warmup": W, "ms_per_step": el_r * 1e3 / K,
                    "kernel_ms_total": k_r, "mode": "mjhip_batch_set_pgs_mode(1): residual-update PGS sweep, tolerance parity (opt-in)",
                    "end_state": {"warnings": int(cur["batch"].get("warning").sum()), "mean_ncon": float(cn[:, 0].mean()),
                                  "mean_nefc": float(cn[:, 1].mean()), "mean_solver_iter": float(cn[:, 5].mean())}}
                if want_state and args.parity_envs > 0:
                    envs = np.unique(np.linspace(0, nenv - 1, min(args.parity_envs, nenv)).astype(int))
                    idx = torch.from_numpy(envs).to(dev)
                    gs = torch.cat([x.index_select(0, idx) for x in sw_r + sk_r], dim=1).cpu().numpy()
                    cs = torch.cat([x.index_select(0, idx) for x in ctrl_w + ctrl_k], dim=1).cpu().numpy()

                    def step_once_r(states, warm, u):
                        outs, cnts = [], []
                        for a in range(0, len(states), nenv):
                            b = min(len(states), a + nenv)
                            small = ma.Batch(dm, b - a, device=local_rank)
                            small.set_pgs_mode(1)
                            outs.append(small.rollout_host(1, ma.mjSTATE_CTRL, states[a:b], warm[a:b], u[a:b])[:, 0])
                            cnts.append(small.get("counts"))
                            small.close()
                        return np.concatenate(outs), np.concatenate(cnts)

                    res["pgs_residual"]["parity_sample"] = parity_sample(model_path, solver_id, integ_id, s0[envs], cs, gs, envs, step_once_r,
                                                                         cfg["ctrl"], iter_exact=False)
            del sw_r, sk_r
            cur["batch"].close()
        except Exception as exc:
            if rank == 0:
                res["pgs_residual"] = {"error": repr(exc)}
        cur["batch"] = batch
    del ctrl_w, ctrl_k

    # ---------------- the same kernel in the reference testspeed's control regime ----------------
    if not args.no_extra and args.ctrl == "uniform" and args.regime_steps > 0 and args.config == "humanoid":
        K2, S2, W2 = args.regime_steps, args.regime_settle, 20
        C2 = max(1, min(args.chunk, K2))
        sizes = lambda n: [C2] * (n // C2) + ([n % C2] if n % C2 else [])
        c_s = make_controls("ou-halton", None, sizes(S2), 0)
        c_w = make_controls("ou-halton", None, sizes(W2), S2)
        c_k = make_controls("ou-halton", None, sizes(K2), S2 + W2)
        el2, timed2, _, _ = timed_region(state0, c_s, c_w, c_k, want_state)
        k2 = sum(t for t, _ in timed2)
        if dist:
            t = torch.tensor([el2, k2], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el2, k2 = float(t[0]), float(t[1])
        cn = batch.get("counts")
        if rank == 0:
            res["testspeed_regime"] = {
                "value": nenv * world * K2 / el2, "unit": "env-steps/s", "steps": K2, "settle": S2 + W2,
                "ms_per_step": el2 * 1e3 / K2, "kernel_ms_total": k2, "steps_per_launch": C2,
                "ctrl": "CtrlNoise of sample/testspeed.cc:72-112 (std 0.01, rate 0.1), one sequence shared by all envs; "
                        "initial states as in the metric leg",
                "end_state": {"warnings": int(batch.get("warning").sum()), "mean_ncon": float(cn[:, 0].mean()),
                              "mean_nefc": float(cn[:, 1].mean()), "mean_pgs_iter": float(cn[:, 5].mean())}}

        # ---- the same regime with the opt-in residual-update PGS sweep (tolerance parity)
        if not args.no_pgs_residual and solver_name == "pgs" and hasattr(batch, "set_pgs_mode"):
            try:
                cur["batch"] = ma.Batch(dm, nenv, device=local_rank)
                cur["batch"].set_pgs_mode(1)
                el4, timed4, _, _ = timed_region(state0, c_s, c_w, c_k, want_state)
                if dist:
                    t = torch.tensor([el4], dtype=torch.float64, device=dev)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    el4 = float(t[0])
                cn = cur["batch"].get("counts")
                if rank == 0 and "pgs_residual" in res:
                    res["pgs_residual"]["testspeed_regime"] = {
                        "value": nenv * world * K2 / el4, "unit": "env-steps/s", "steps": K2, "settle": S2 + W2, "ms_per_step": el4 * 1e3 / K2,
                        "end_state": {"warnings": int(cur["batch"].get("warning").sum()), "mean_ncon": float(cn[:, 0].mean()),
                                      "mean_nefc": float(cn[:, 1].mean()), "mean_pgs_iter": float(cn[:, 5].mean())}}
                cur["batch"].close()
            except Exception as exc:
                if rank == 0:
                    res.setdefault("pgs_residual", {})["testspeed_regime"] = {"error": repr(exc)}
            cur["batch"] = batch

        # ---- the same regime with the model's OWN solver (humanoid.xml ships the default: Newton): the configuration the
        # reference's published figures are quoted on (doc/mjx.rst:187-215, :663-676: humanoid, Newton, testspeed), so the
        # number can sit next to them; generic kernel variant (the lean one is PGS-only)
        if not args.no_newton_regime and solver_name == "pgs":
            try:
                model_n = ma.MjbModel(lib, model_path)
                if integ_id is not None:
                    model_n.set_option("integrator", integ_id)
                dm_n = ma.DeviceModel(lib, model_n)
                cur["batch"] = ma.Batch(dm_n, nenv, device=local_rank)
                el3, timed3, _, _ = timed_region(state0, c_s, c_w, c_k, want_state)
                k3 = sum(t for t, _ in timed3)
                if dist:
                    t = torch.tensor([el3, k3], dtype=torch.float64, device=dev)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    el3, k3 = float(t[0]), float(t[1])
                cn = cur["batch"].get("counts")
                if rank == 0:
                    res["newton_regime"] = {
                        "value": nenv * world * K2 / el3, "unit": "env-steps/s", "steps": K2, "settle": S2 + W2, "solver": "NEWTON",
                        "ms_per_step": el3 * 1e3 / K2, "kernel_ms_total": k3, "kernel": cur["batch"].kernel_name(),
                        "ctrl": "as testspeed_regime; humanoid.xml's own solver (Newton), " + integ_name,
                        "end_state": {"warnings": int(cur["batch"].get("warning").sum()), "mean_ncon": float(cn[:, 0].mean()),
                                      "mean_nefc": float(cn[:, 1].mean()), "mean_solver_iter": float(cn[:, 5].mean())}}
                cur["batch"].close()
            except Exception as exc:
                if rank == 0:
                    res["newton_regime"] = {"error": repr(exc)}
            cur["batch"] = batch
        del c_s, c_w, c_k

    # ---------------- the drop-in entry point itself: host arrays in and out through mjhip_rollout ----------------
    if rank == 0 and world == 1 and not args.no_extra and not args.leg and args.config == "humanoid" and args.api_steps > 0:
        try:
            res["api_regime"] = api_regime(model_path, solver_id, integ_id, nenv, args.api_steps, s0, cfg, batch, stream, dev)
        except Exception as exc:
            res["api_regime"] = {"error": repr(exc)}

    # ---------------- flex: the fall from the reset state, reported apart from the metric ----------------
    if not args.no_extra and cfg.get("free_fall_steps") and args.settle > 0:
        K3, W3 = cfg["free_fall_steps"], 20
        c_w = make_controls("uniform", crng, [W3], 0)
        c_k = make_controls("uniform", crng, [K3], W3)
        el3, timed3, _, _ = timed_region(state0, [], c_w, c_k, want_state)
        cn = batch.get("counts")
        if rank == 0:
            res["free_fall_regime"] = {
                "value": nenv * world * K3 / el3, "unit": "env-steps/s", "steps": K3, "warmup": W3, "ms_per_step": el3 * 1e3 / K3,
                "note": "the same kernel from the reset state (no settle): nearly contact-free -- a different regime from `value`, "
                        "reported for completeness and never under the metric's name",
                "end_state": {"mean_ncon": float(cn[:, 0].mean()), "mean_nefc": float(cn[:, 1].mean()),
                              "mean_solver_iter": float(cn[:, 5].mean())}}

    if rank == 0:
        ncpu = os.cpu_count() or 1
        if world == 1 and not args.no_cpu_baseline and not args.no_extra:
            cb = cpu_baseline(ncpu, budget_s=4.0 if args.leg else 15.0, mjb_name=cfg["mjb"],
                              solver_flag=(args.solver or cfg["solver"] or "").upper() or None)
            if cb:
                res["cpu_baseline"] = cb
            # the CPU side of newton_regime: the reference's testspeed with the model's own solver
            if cb and "newton_regime" in res and "value" in res["newton_regime"]:
                cbn = cpu_baseline(ncpu, budget_s=5.0, mjb_name=cfg["mjb"], solver_flag="Newton")
                if cbn:
                    res["newton_regime"]["cpu_testspeed"] = cbn
            # the same workload as `value` (same states, same controls, same steps) on the host cores
            if args.ctrl == "uniform" and (args.settle == 0 or snap_metric) and n_metric_ctrl > n_settle_ctrl:
                try:
                    leg = cpu_rollout_leg(cfg["mjb"], solver_id, integ_id, snap_metric["state"] if snap_metric else s0,
                                          np.concatenate(host_ctrl[n_settle_ctrl:n_metric_ctrl], axis=1), ncpu,
                                          warm0=snap_metric["warm"] if snap_metric else None,
                                          min_seconds=2.0 if args.leg else 3.0)
                except Exception as exc:
                    leg = ({"error": repr(exc)}, None)
                if leg and leg[0].get("value"):
                    # `cpu_baseline.value` is the LIKE-FOR-LIKE figure: the reference stepping this run's own states and
                    # controls; the testspeed binary's own (quieter / noisier) regime sits beside it
                    ts = res.get("cpu_baseline")
                    res["cpu_baseline"] = {"value": leg[0]["value"], "unit": "env-steps/s", "cores": ncpu, "kind": "reference",
                                           "sample": leg[0].get("sample"), "rollout_regime": leg[0]}
                    if ts:
                        res["cpu_baseline"]["testspeed_regime"] = ts
                elif leg:
                    res.setdefault("cpu_baseline", {})["rollout_regime"] = leg[0]
        # ---------------- the other single-GPU BASELINE configurations, one short sub-run each ----------------
        if (args.config == "humanoid" and world == 1 and not args.no_extra and not args.no_legs and not args.leg
                and not args.envs_per_gpu):
            # (the legs are processes of their own on the same GPU: give the parent's batch back first)
            try:
                batch.close()
                del state0
                torch.cuda.empty_cache()
            except Exception:
                pass
            res["configs"] = config_legs()
            res["configs_ok"] = all("error" not in v and (v.get("parity_sample") or {}).get("ok") is True for v in res["configs"].values())
        if args.leg:
            print(json.dumps(_finite(res)), flush=True)       # (read by the parent run, never by the driver)
        else:
            emit(res, args.config)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
