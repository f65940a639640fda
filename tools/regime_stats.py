"""Testspeed regime (settled humanoids, CtrlNoise controls): per-environment cost of a 100-step launch
against the environment's constraint count -- which environments end the launch, and on which PGS path.
usage (GPU box): python tools/regime_stats.py [settle_steps]"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mujoco_amd as ma
from mujoco_amd import _capi as K
import bench


def main():
    settle = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    lib = ma.lib()
    dev = torch.device("cuda:0")
    model = ma.MjbModel(lib, os.path.join(bench.ROOT, "tests", "golden", "humanoid.mjb"))
    model.set_option("solver", 0)
    dm = ma.DeviceModel(lib, model)
    nenv = 4096
    batch = ma.Batch(dm, nenv, device=0)
    s0 = bench.initial_states(batch.get("qpos")[0], dm.nv, nenv, seed=1234)
    nu = dm.nu
    lo, hi = -np.ones(nu), np.ones(nu)
    total = settle + 300
    seq = bench.ctrl_noise(total, nu, 0.005, lo, hi)           # one sequence shared by all envs (testspeed)
    state0 = torch.from_numpy(s0).to(dev)
    done = 0
    first = True
    while done < settle:
        n = min(100, settle - done)
        c = torch.from_numpy(np.broadcast_to(seq[done:done + n], (nenv, n, nu)).copy()).to(dev)
        batch.rollout_device(n, K.mjSTATE_CTRL, state0.data_ptr() if first else 0, 0, c.data_ptr(), 0, 0, cont=not first)
        first = False
        done += n
    torch.cuda.synchronize()
    for L in range(3):
        c = torch.from_numpy(np.broadcast_to(seq[done:done + 100], (nenv, 100, nu)).copy()).to(dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        batch.rollout_device(100, K.mjSTATE_CTRL, 0, 0, c.data_ptr(), 0, 0, cont=True)
        ev1.record()
        torch.cuda.synchronize()
        done += 100
        ms = ev0.elapsed_time(ev1)
        wall = batch.get("wall")[:, 0].astype(np.float64) * 16 / 100.0   # us
        cnt = batch.get("counts")
        nefc, niter = cnt[:, 1], cnt[:, 5]
        print(f"launch {L}: kernel {ms:.1f} ms ({nenv*100/ms/1e3:.3f} M env-steps/s); per-env wall us: mean {wall.mean():.0f} "
              f"p50 {np.quantile(wall,.5):.0f} p90 {np.quantile(wall,.9):.0f} p99 {np.quantile(wall,.99):.0f} max {wall.max():.0f}")
        edges = [0, 17, 33, 49, 65, 97, 129, 1000]
        for a, b in zip(edges[:-1], edges[1:]):
            sel = (nefc >= a) & (nefc < b)
            if sel.any():
                print(f"   nefc(last step) in [{a:3d},{b:4d}): {sel.sum():5d} envs, mean wall {wall[sel].mean():8.0f} us, max {wall[sel].max():8.0f}, mean niter(last) {niter[sel].mean():5.1f}")
        print("   corr(nefc_last, wall) = %.3f" % np.corrcoef(nefc, wall)[0, 1])


if __name__ == "__main__":
    main()
