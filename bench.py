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
ranks.  Environments shard across
ranks with no per-step exchange (weak scaling: 4096 envs on every GPU); the only collective is the
end-of-chunk gather of the final states to rank 0 over RCCL, inside the timed region.

Prints ONE JSON line (rank 0) with the throughput, the roofline object for the dominant (only)
kernel and, at N=1, the CPU baseline: the reference engine's own `testspeed` (oracle/_ref, built
from the reference sources) on all host cores for a bounded sample.
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
# algorithmic HBM bytes per env-step: state read + state written
# (2*nstate), control read (ncontrol), warmstart read + written (2*nv), 8 bytes each
HBM_PEAK_GBS = 8000.0


def initial_states(qpos0: np.ndarray, nv: int, nenv: int, seed: int) -> np.ndarray:
    """reset state + hinge perturbation N(0,0.05^2), qvel ~ N(0,0.1^2), env-major draws.
    humanoid: qpos[0:7] is the free joint (left untouched), the rest are hinges."""
    rng = np.random.Generator(np.random.PCG64(seed))
    nq = qpos0.size
    s0 = np.zeros((nenv, 1 + nq + nv))
    for e in range(nenv):
        s0[e, 1:1 + nq] = qpos0
        s0[e, 8:1 + nq] += rng.normal(0, 0.05, size=nq - 7)
        s0[e, 1 + nq:] = rng.normal(0, 0.1, size=nv)
    return s0


def cpu_baseline(nthread: int, budget_s: float = 15.0) -> dict | None:
    """reference CPU engine timed by the reference's own sample/testspeed.cc (compiled from the
    reference sources into oracle/_ref by oracle/Makefile) on the host cores of this box."""
    exe = os.path.join(ROOT, "oracle", "_ref", "testspeed")
    mjb = os.path.join(ROOT, "tests", "golden", "humanoid.mjb")
    if not os.path.exists(exe):
        return None
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref"))

    def run(nstep):
        out = subprocess.run([exe, mjb, f"--nstep={nstep}", f"--nthread={nthread}", "--solver=PGS"],
                             capture_output=True, text=True, env=env, timeout=600).stdout
        m = re.search(r"Total steps per second\s*:\s*([0-9.]+)", out)
        it = re.search(r"PGS iters / step\s*:\s*([0-9.]+)", out)
        return (float(m.group(1)) if m else None), (float(it.group(1)) if it else None)

    sps, _ = run(2000)                          # calibration
    if not sps:
        return None
    nstep = int(max(2000, min(400000, budget_s * sps / nthread)))
    sps, iters = run(nstep)
    return {"value": sps, "unit": "env-steps/s", "cores": nthread, "kind": "reference",
            "sample": f"reference sample/testspeed.cc on liboracle_fast (-O3 -mavx), humanoid.mjb --solver=PGS "
                      f"--nthread={nthread} --nstep={nstep} (OU-Halton ctrl noise, its default regime; "
                      f"{iters} PGS iters/step)"}


def measured_traffic(steps_per_launch: int, nenv: int):
    """HBM bytes per launch from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this
    same command (tools/gpu_profile.sh -> profiles/<round>/pmc_summary.txt; FETCH_SIZE x2 per the
    gfx950 correction of MI355X_MICROARCH.md).  None when no committed summary matches the launch."""
    import glob
    best = None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_summary.txt")))
    # profiles/CURRENT names the directory measured on the committed build; it is read last (wins)
    cur = os.path.join(ROOT, "profiles", "CURRENT")
    if os.path.exists(cur):
        f0 = os.path.join(ROOT, "profiles", open(cur).read().strip(), "pmc_summary.txt")
        if f0 in files:
            files.remove(f0)
            files.append(f0)
    for f in files:
        m = re.search(r"per launch \((\d+) steps x (\d+) envs\): read ([0-9.]+) MB raw / ([0-9.]+) MB with .*?written ([0-9.]+) MB",
                      open(f).read())
        if m and int(m.group(1)) == steps_per_launch and int(m.group(2)) == nenv:
            best = (float(m.group(4)) + float(m.group(5))) * 1e6
    return best


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--chunk", type=int, default=100, help="steps per rollout-kernel launch (one open-loop rollout call)")
    ap.add_argument("--envs-per-gpu", type=int, default=NENV_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-state-output", action="store_true", help="do not write the per-step state array")
    ap.add_argument("--solver", choices=["pgs", "newton"], default="pgs",
                    help="pgs = BASELINE config 2 (the metric); newton = the reference's default solver")
    ap.add_argument("--integrator", choices=["euler", "rk4"], default="euler")
    args = ap.parse_args()

    import torch
    import mujoco_amd as ma
    from mujoco_amd.sharding import gather_to_rank0

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
    model = ma.MjbModel(lib, os.path.join(ROOT, "tests", "golden", "humanoid.mjb"))
    model.set_option("solver", 0 if args.solver == "pgs" else 2)     # PGS = BASELINE config 2
    model.set_option("integrator", 0 if args.integrator == "euler" else 1)
    dm = ma.DeviceModel(lib, model)
    nenv, K, W = args.envs_per_gpu, args.steps, args.warmup
    nq, nv, nu, nstate = dm.nq, dm.nv, dm.nu, dm.nstate
    batch = ma.Batch(dm, nenv, device=local_rank)
    qpos0 = batch.get("qpos")[0]

    # synthetic inputs, resident in HBM before timing.  Steps are issued in launches of C steps
    # (one rollout-kernel launch = C x nenv env-steps) so that every launch is the same unit of
    # work for the profiler; controls / state outputs are laid out per launch.
    C = max(1, min(args.chunk, K))
    def chunks(n):
        return [C] * (n // C) + ([n % C] if n % C else [])
    s0 = initial_states(qpos0, nv, nenv, seed=1234 + rank)
    crng = np.random.Generator(np.random.PCG64(4321 + rank))
    dev = torch.device("cuda", local_rank)
    state0 = torch.from_numpy(s0).to(dev)
    lo, hi = -1.0, 1.0       # humanoid ctrlrange
    ctrl_w = [torch.from_numpy(crng.uniform(lo, hi, size=(nenv, c, nu))).to(dev) for c in chunks(W)]
    ctrl_k = [torch.from_numpy(crng.uniform(lo, hi, size=(nenv, c, nu))).to(dev) for c in chunks(K)]
    state_k = [None if args.no_state_output else torch.empty((nenv, c, nstate), dtype=torch.float64, device=dev)
               for c in chunks(K)]
    final = torch.empty((nenv, nstate), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def barrier():
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    events = []

    def launch(ctrl, out, first):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        # first launch loads the initial states; later ones continue from the batch's own state,
        # warm start and warning counters (MJHIP_ROLLOUT_CONTINUE)
        batch.rollout_device(ctrl.shape[1], ma.mjSTATE_CTRL, state0.data_ptr() if first else 0,
                             0, ctrl.data_ptr(), 0 if out is None else out.data_ptr(), stream,
                             cont=not first)
        e1.record()
        events.append((e0, e1, ctrl.shape[1]))

    # warmup: W untimed steps from the initial states
    first = True
    for c in ctrl_w:
        launch(c, None, first)
        first = False
    barrier()
    n_warm_launch = len(events)

    t0 = time.perf_counter()
    for c, o in zip(ctrl_k, state_k):
        launch(c, o, first)
        first = False
    if state_k[-1] is not None:
        final.copy_(state_k[-1][:, -1])
    if dist:
        gather_to_rank0(final, rank, world, dist)  # end-of-run state gather (RCCL over xGMI)
    barrier()
    elapsed = time.perf_counter() - t0
    timed = [(a.elapsed_time(b), n) for a, b, n in events[n_warm_launch:]]
    allev = [(a.elapsed_time(b), n) for a, b, n in events]
    kernel_ms = sum(t for t, _ in timed)
    launch_ms_timed = float(np.mean([t for t, n in timed if n == C])) if any(n == C for _, n in timed) else kernel_ms
    launch_ms_all = float(np.mean([t for t, n in allev if n == C])) if any(n == C for _, n in allev) else launch_ms_timed

    if dist:
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])
    warn = int(batch.get("warning").sum())
    counts = batch.get("counts")

    if rank == 0:
        total_env_steps = nenv * world * K
        value = total_env_steps / elapsed
        bytes_per_env_step = 8 * (2 * nstate + nu + 2 * nv)
        # one launch = C steps of nenv envs; achieved = algorithmic bytes per launch / avg launch time
        achieved = bytes_per_env_step * nenv * C / (launch_ms_timed * 1e-3) / 1e9
        res = {
            "metric": "env-steps/sec on humanoid.xml, 4096 envs/GPU",
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
            "config": {"workload": f"model/humanoid/humanoid.xml, {nenv} envs/GPU, {args.solver.upper()} solver, {args.integrator}, fp64, "
                                   "random actions U(ctrlrange), per-step state output" +
                                   ("" if not args.no_state_output else " disabled"),
                       "envs_per_gpu": nenv, "nstep": K, "solver": args.solver.upper(), "integrator": args.integrator,
                       "parallelism": f"env-sharded x{world}",
                       "mapping": batch.lds_report().splitlines()[0] if batch.lds_report() else "no LDS plan",
                       "layout": os.environ.get("MJHIP_LAYOUT", "aos")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(C, nenv),
                         "kernel": "mjh_k_rollout", "steps_per_launch": C,
                         "launch_ms": launch_ms_timed, "launch_ms_incl_warmup": launch_ms_all,
                         "kernel_ms_total": kernel_ms,
                         "algorithmic_bytes_per_env_step": bytes_per_env_step,
                         "algorithmic_bytes_per_launch": bytes_per_env_step * nenv * C},
            "end_state": {"warnings": warn, "mean_ncon": float(counts[:, 0].mean()),
                          "mean_nefc": float(counts[:, 1].mean()), "mean_pgs_iter": float(counts[:, 5].mean())},
        }
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(os.cpu_count() or 1)
            if cb:
                res["cpu_baseline"] = cb
        print(json.dumps(res), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
