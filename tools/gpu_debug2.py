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
mode, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
s0 = fx['s0'][lo:hi]; ws = fx['ws'][lo:hi]; ctrl = fx['ctrl'][lo:hi]
n = hi-lo
b = K.Batch(dm, n)
if mode == 'rollout':
    out = b.rollout_host(1, K.mjSTATE_CTRL, s0, ws, ctrl)[:,0]
    print('maxdiff', float(np.abs(out-fx['out'][lo:hi]).max()))
elif mode == 'rollout_nows':
    out = b.rollout_host(1, K.mjSTATE_CTRL, s0, None, ctrl)[:,0]
    print('ok')
elif mode == 'forward':
    b.set('time', s0[:, :1]); b.set('qpos', s0[:,1:29]); b.set('qvel', s0[:,29:]); b.set('qacc_warmstart', ws); b.set('ctrl', ctrl[:,0])
    b.forward()
    print('forward ok', b.get('counts')[:, :6].max(0))
elif mode.startswith('stages'):
    b.set('time', s0[:, :1]); b.set('qpos', s0[:,1:29]); b.set('qvel', s0[:,29:]); b.set('qacc_warmstart', ws); b.set('ctrl', ctrl[:,0])
    for st in mode.split('_')[1:]:
        b.forward(int(st))
    out = np.concatenate([b.get('time'), b.get('qpos'), b.get('qvel')], axis=1)
    print(mode, 'maxdiff', float(np.abs(out-fx['out'][lo:hi]).max()))
elif mode == 'step':
    b.set('time', s0[:, :1]); b.set('qpos', s0[:,1:29]); b.set('qvel', s0[:,29:]); b.set('qacc_warmstart', ws); b.set('ctrl', ctrl[:,0])
    b.step(1)
    out = np.concatenate([b.get('time'), b.get('qpos'), b.get('qvel')], axis=1)
    print('step maxdiff', float(np.abs(out-fx['out'][lo:hi]).max()))
''' % (ROOT, os.path.join(ROOT,'tests/golden/humanoid.mjb'), os.path.join(ROOT,'tools/dbg_inputs.npz'))
for spec in sys.argv[1:]:
    mode, lo, hi = spec.split(':')
    r = subprocess.run([sys.executable, '-c', CODE, mode, lo, hi], capture_output=True, text=True, env=dict(os.environ))
    err = [l for l in r.stderr.strip().splitlines() if 'fault' in l or 'Error' in l or 'error' in l]
    print('==', spec, 'rc', r.returncode, (r.stdout.strip().splitlines() or [''])[-1], '|', (err or [''])[-1][:160], flush=True)
