"""One-ulp sensitivity of the compiled reference on cube_3x3x3: every qpos component of every golden sample moved by one ulp,
the next state compared with the unperturbed step (profiles/r04/cube_discontinuity.txt; pinned by
tests/test_oracle_golden.py::test_cube_contact_discontinuity).  Runs where oracle/_ref is built (about a minute)."""
import os, sys, numpy as np
sys.path.insert(0, '/root/repo')
from oracle import refbind as rb
G='/root/repo/tests/golden'
fx = np.load(os.path.join(G, "cube_3x3x3_steps.npz"))
m = rb.MjModel.from_binary_path(os.path.join(G, "cube_3x3x3.mjb"))
d = rb.MjData(m)
spec = rb.mjSTATE_FULLPHYSICS
nq = m.nq
def step(state, warm, ctrl):
    rb.mj_resetData(m, d)
    rb.mj_setState(m, d, state, spec)
    d.qacc_warmstart[:] = warm
    d.ctrl[:] = ctrl
    rb.mj_step(m, d)
    return rb.mj_getState(m, d, spec).copy(), int(d.ncon), int(d.nefc), int(d.solver_niter[0])
best = []
N = fx["state"].shape[0]
print("samples", N, "nq", nq, flush=True)
for k in range(N):
    s0 = fx["state"][k]; w = fx["warmstart"][k]; u = fx["ctrl"][k]
    base, nc, ne, ni = step(s0, w, u)
    for j in range(nq):
        for sgn in (1, -1):
            s1 = s0.copy()
            s1[1 + j] = np.nextafter(s1[1 + j], np.inf if sgn > 0 else -np.inf)
            if s1[1+j] == s0[1+j]: continue
            nxt, nc2, ne2, ni2 = step(s1, w, u)
            err = float(np.max(np.abs(nxt - base)/np.maximum(1, np.abs(base))))
            if err > 1e-9:
                best.append((err, k, j, sgn, nc == nc2, ne == ne2, ni == ni2))
                print("hit", err, k, j, sgn, nc, nc2, ne, ne2, ni, ni2, flush=True)
best.sort(reverse=True)
print(best[:20])
