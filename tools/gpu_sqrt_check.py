"""is the device's fp64 sqrt / division / reciprocal correctly rounded?  (torch's kernels are compiled by the same hipcc lowering)"""
import numpy as np, torch
rng = np.random.default_rng(0)
for name, x in [("U(0.25,4)", rng.uniform(0.25, 4.0, 20_000_000)), ("U(0,1e-3)", rng.uniform(0, 1e-3, 5_000_000)), ("lognormal", np.exp(rng.normal(0, 20, 5_000_000)))]:
    xg = torch.from_numpy(x).cuda()
    s = torch.sqrt(xg).cpu().numpy()
    ref = np.sqrt(x)
    print("sqrt", name, "mismatches", int((s != ref).sum()), "of", x.size)
a = rng.uniform(-4, 4, 20_000_000); b = rng.uniform(0.1, 4, 20_000_000)
q = (torch.from_numpy(a).cuda()/torch.from_numpy(b).cuda()).cpu().numpy()
print("div mismatches", int((q != a/b).sum()))
r = (1.0/torch.from_numpy(b).cuda()).cpu().numpy()
print("rcp mismatches", int((r != 1.0/b).sum()))
