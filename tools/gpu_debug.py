import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import mujoco_amd as ma
from mujoco_amd import _capi as K
lib = ma.lib()
m = ma.MjbModel(lib, %r); m.set_option('solver', 0)
dm = K.DeviceModel(lib, m)
fx = np.load(%r)
variant = sys.argv[1]
n = 8
b = K.Batch(dm, n)
s0 = fx['state0'][:n]; ctrl = fx['ctrl'][:n,:2]
ws = np.zeros((n,27))
if variant == 'ws_zero':
    out = b.rollout_host(2, K.mjSTATE_CTRL, s0, ws, ctrl)
elif variant == 'ws_small':
    out = b.rollout_host(2, K.mjSTATE_CTRL, s0, ws+0.1, ctrl)
elif variant == 'ws_nostep':
    out = b.rollout_host(0, K.mjSTATE_CTRL, s0, ws+0.1, None)
elif variant == 'no_ws':
    out = b.rollout_host(2, K.mjSTATE_CTRL, s0, None, ctrl)
elif variant == 'ws_n64':
    b = K.Batch(dm, 64); rep = np.arange(64) %% n
    out = b.rollout_host(1, K.mjSTATE_CTRL, s0[rep], np.zeros((64,27)), ctrl[rep][:, :1])
print(variant, 'OK', None if out is None else float(np.abs(out).max()))
''' % (ROOT, os.path.join(ROOT,'tests/golden/humanoid.mjb'), os.path.join(ROOT,'tests/golden/humanoid_traj.npz'))
for v in sys.argv[1:]:
    r = subprocess.run([sys.executable, '-c', CODE, v], capture_output=True, text=True)
    print('==', v, 'rc', r.returncode, (r.stdout.strip().splitlines() or [''])[-1], '|', (r.stderr.strip().splitlines() or [''])[-1][:200], flush=True)
