"""Per-environment cost of one rollout launch (the `wall` field the kernel writes: wall_clock64 >> 4
ticks, 100 MHz clock): how much of a launch is the tail of its slowest environments."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mujoco_amd as ma
from mujoco_amd import _capi as K
import bench

def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    nlaunch = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    lib = ma.lib()
    dev = torch.device("cuda:0")
    model = ma.MjbModel(lib, os.path.join(bench.ROOT, "tests", "golden", "humanoid.mjb"))
    model.set_option("solver", 0)
    dm = ma.DeviceModel(lib, model)
    nenv = 4096
    batch = ma.Batch(dm, nenv, device=0)
    s0 = bench.initial_states(batch.get("qpos")[0], dm.nv, nenv, seed=1234)
    rng = np.random.default_rng(4321)
    nu = dm.nu
    state0 = torch.from_numpy(s0).to(dev)
    for L in range(nlaunch):
        ctrl = torch.from_numpy(rng.uniform(-1, 1, size=(nenv, steps, nu))).to(dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        batch.rollout_device(steps, K.mjSTATE_CTRL, state0.data_ptr() if L == 0 else 0, 0, ctrl.data_ptr(), 0, 0, cont=(L > 0))
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1)
        c = batch.get("wall")[:, 0].astype(np.float64) * 16 / 100.0   # us
        work = batch.get("cost")[:, 0].astype(np.float64)
        q = np.quantile(c, [0, .5, .9, .99, .999, 1])
        if L > 0:
            print("   corr(work estimate of previous launch, of this launch) = %.3f ; corr(work estimate, wall) = %.3f" %
                  (np.corrcoef(prev_work, work)[0, 1], np.corrcoef(work, c)[0, 1]))
        prev_work = work
        print("launch %d: kernel %.2f ms (%.2f M env-steps/s); per-env us: mean %.0f  quantiles(0,.5,.9,.99,.999,1) %s  max/mean %.2f" %
              (L, ms, nenv*steps/ms/1e3, c.mean(), np.round(q).tolist(), c.max()/c.mean()))
        cnt = batch.get("counts")
        slow = np.argsort(-c)[:6]
        print("   slowest: " + "; ".join("env %d %.0f us work %.0f (last step: ncon %d nefc %d niter %d)" %
                                         (i, c[i], work[i], cnt[i, 0], cnt[i, 1], cnt[i, 5]) for i in slow))
        print("   work estimate: mean %.0f, of the slowest six %s" % (work.mean(), np.round(work[slow]).tolist()))

if __name__ == "__main__":
    main()
