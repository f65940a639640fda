"""Per-stage time of the rollout kernel (needs a library built with -DMJH_PROFILE; see
tools/gpu_stageprof.sh).  Prints mean microseconds per env-step spent in each timeline stage."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mujoco_amd as ma
from bench import initial_states

NAMES = ["begin", "kin", "collision", "compos", "tendon", "transmission", "tavel", "comvel",
         "passive", "rne", "crb", "factor", "actuation", "accel", "make", "project", "reference", "constraint",
         "finish", "euler", "end"]
lib = ma.lib()
from bench import CONFIGS
cfg = CONFIGS[os.environ.get("MODEL", "humanoid")]        # MODEL=cube: BASELINE config 4 as shipped (Newton, implicitfast)
model = ma.MjbModel(lib, os.path.join(ROOT, "tests", "golden", cfg["mjb"]))
if cfg["solver"] or "SOLVER" in os.environ:
    model.set_option("solver", int(os.environ.get("SOLVER", "0")))     # 0 PGS, 1 CG, 2 Newton
dm = ma.DeviceModel(lib, model)
nenv = int(os.environ.get("NENV", cfg["nenv"])); K = int(os.environ.get("K", 100)); W = int(os.environ.get("W", 50))
b = ma.Batch(dm, nenv)
print("variant", b.kernel_variant(), "|", b.lds_report().splitlines()[0])
s0 = initial_states(b.get("qpos")[0], dm.nv, nenv, 1234, cfg["free_root"])
rng = np.random.Generator(np.random.PCG64(4321))
dev = torch.device("cuda", 0)
st0 = torch.from_numpy(s0).to(dev)
cw = torch.from_numpy(rng.uniform(cfg["ctrl"][0], cfg["ctrl"][1], size=(nenv, W, dm.nu))).to(dev)
ck = torch.from_numpy(rng.uniform(cfg["ctrl"][0], cfg["ctrl"][1], size=(nenv, K, dm.nu))).to(dev)

b.rollout_device(W, ma.mjSTATE_CTRL, st0.data_ptr(), 0, cw.data_ptr(), 0, 0)
b.sync()
b.set("prof", np.zeros((nenv, b.get("prof").shape[1])))
import time
t0 = time.perf_counter()
for c in range(0, K, 10):
    if c == K - 10:
        b.sync(); cost_pred = b.get("cost")[:, 0].copy()
    b.rollout_device(10, ma.mjSTATE_CTRL, 0, 0, ck[:, c:c+10].contiguous().data_ptr(), 0, 0, cont=True)
b.sync()
el = time.perf_counter() - t0
p = b.get("prof")
nst = p[:, 30].mean()
print(f"wall {el*1e3/K:.3f} ms/step -> {nenv*K/el/1e6:.3f} M env-steps/s; in-kernel us per env-step: {p[:,31].mean()/nst:.1f}")
tot = 0
for i, n in enumerate(NAMES):
    v = p[:, i].mean() / nst
    tot += v
    if v > 0: print(f"  {n:14s} {v:9.2f} us")
print(f"  {'sum':14s} {tot:9.2f} us")
if p.shape[1] >= 48 and p[:, 46:48].sum() > 0:
    print("  inside collision (convex narrowphase): lane-parallel distance phase %.1f  row-cooperative penetration phase %.1f us (row 0 of it: support queries %.1f, multi-contact %.1f)"
          % tuple(p[:, k].mean()/nst for k in (46, 47, 25, 21)))
print("  inside constraint: b/jar/warmstart %.2f  PGS %.2f  J'f %.2f us" % tuple(p[:, k].mean()/nst for k in (22, 23, 24)))
if p.shape[1] >= 40 and p[:, 32:38].sum() > 0:
    print("  inside constraint (primal solvers): set-up %.1f  Hessian+factor %.1f  factor solves %.1f  incremental updates %.1f  line search %.1f  constraint update+grad %.1f us"
          % tuple(p[:, k].mean()/nst for k in (32, 33, 34, 35, 36, 37)))
if p.shape[1] >= 46 and p[:, 41:46].sum() > 0:
    print("  per step: gradient norm %.1f  direction update + |search| %.1f  M search %.1f  J search %.1f  PrimalPrepare sums %.1f  quadratic coefficients %.1f us"
          % tuple(p[:, k].mean()/nst for k in (41, 42, 43, 44, 45, 38)))
if p.shape[1] >= 41 and p[:, 39].sum() > 0:
    print("  line search, not in the figure above: products + quadratic coefficients %.1f us; %.1f searches and %.1f evaluations per step"
          % (p[:, 38].mean()/nst, p[:, 39].mean()/nst, p[:, 40].mean()/nst))
if p.shape[1] >= 53 and p[:, 48:53].sum() > 0:
    print("  outside the stages: state checks %.1f  compressed rows + islands %.1f  state / sensor output %.1f  control input %.1f  sensors %.1f us"
          % tuple(p[:, k].mean()/nst for k in (48, 49, 50, 51, 52)))
if p.shape[1] >= 60 and p[:, 53:60].sum() > 0:
    print("  inside collision (flex jobs): planes %.1f  leaf culling %.1f  element narrowphase %.1f  filter / sort / emission %.1f us"
          % tuple(p[:, k].mean()/nst for k in (53, 54, 55, 56)))
    print("  inside compressed rows: row lengths + addresses %.1f  columns and values %.1f  transpose %.1f us (the rest of that figure: mj_island)"
          % tuple(p[:, k].mean()/nst for k in (57, 58, 59)))
c = b.get("counts")
print("mean ncon", c[:, 0].mean(), "nefc", c[:, 1].mean(), "pgs iter", c[:, 5].mean())

# residency census of the last launch: how many waves were running at the median time
st, en = p[:, 28], p[:, 29]
tm = np.median(np.concatenate([st, en]))
print("last launch: span %.1f us, mean wave duration %.1f us, waves resident at median time: %d of %d" % (
    en.max() - st.min(), (en - st).mean(), int(((st <= tm) & (en >= tm)).sum()), nenv))
hw = p[:, 27].astype(np.int64); xcc = p[:, 26].astype(np.int64) & 0xf
cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1; simd = (hw >> 4) & 3
key = xcc * 100000 + se * 10000 + sh * 1000 + cu * 10 + simd
u, cnt = np.unique(key, return_counts=True)
print("distinct (xcc,se,sh,cu,simd):", len(u), " distinct xcc:", len(np.unique(xcc)), " waves per simd: min %d max %d" % (cnt.min(), cnt.max()))
order = np.argsort(st)
print("start-time quantiles (us from first):", np.round(np.quantile(st - st.min(), [0, .1, .25, .5, .75, .9, 1]), 1))

# the heaviest environments of the measured window
tot = p[:, 31]
idx = np.argsort(-tot)[:5]
print("heaviest envs (us per step by stage):")
for e in idx:
    print("  env %d: total %.0f us/step, nefc(last) %d niter(last) %d :" % (e, tot[e]/nst, c[e,1], c[e,5]),
          " ".join("%s=%.0f" % (NAMES[i], p[e,i]/nst) for i in range(len(NAMES)) if p[e,i]/nst >= 5))
print("duration quantiles us/step:", np.round(np.quantile(tot/nst, [0, .25, .5, .75, .9, .99, 1]), 0))
print("nefc quantiles:", np.quantile(c[:,1], [0, .25, .5, .75, .9, .99, 1]), " niter quantiles:", np.quantile(c[:,5], [0,.25,.5,.75,.9,.99,1]))

t0_ = st.min()
ts = np.linspace(0, en.max() - t0_, 21)
print("running waves over time (us: count):", " ".join("%d:%d" % (t, int(((st - t0_ <= t) & (en - t0_ >= t)).sum())) for t in ts))
dur = en - st
late = np.argsort(-en)[:6]
print("last finishers: (start us, dur us, cost-rank)", [(int(st[e]-t0_), int(dur[e]), int((dur > dur[e]).sum())) for e in late])
first = np.argsort(st)[:2048]
print("mean duration of the first 2048 starters %.0f us, of the rest %.0f us" % (dur[first].mean(), np.delete(dur, first).mean()))

cost_now = b.get("cost")[:, 0]
print("cost prediction: corr(prev launch cost, this launch cost) = %.3f" % np.corrcoef(cost_pred, cost_now)[0,1])
