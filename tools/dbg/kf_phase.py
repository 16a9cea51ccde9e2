"""Debug aid: cycle stamps of the phases of a team filter step (developer build -DHILO_KF_PROF, HILO_LIB_PATH pointing at it)."""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
import hilo_mpc_amd as H                                                  # noqa: E402
from hilo_mpc_amd import _lib                                             # noqa: E402

B, K = 4096, 16
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
x = np.array([.1, 40., .5, .2]) * (1 + .1 * rng.uniform(-1, 1, (B, 4)))
for kind in ('EKF', 'UKF'):
    f = getattr(H, kind)(H.Model('chemostat4').discretize('rk4').setup(dt=1.))
    f.setup()
    f.Q, f.R = 1e-4, 1e-2
    f.set_initial_guess(torch.as_tensor(x, device=dev), P0=torch.as_tensor(np.tile(np.eye(4), (B, 1, 1)), device=dev))
    u = torch.as_tensor(rng.uniform(0, .3, (B, 2)), device=dev)
    p = torch.as_tensor(np.tile([100., 4., 1., 0.], (B, 1)), device=dev)
    y = torch.as_tensor(x[:, [0, 2]], device=dev)[None] * (1 + torch.as_tensor(.02 * rng.normal(size=(K, B, 2)), device=dev))
    for _ in range(5):
        f.estimate(y=y, u=u, p=p, steps=K)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 16)()
    _lib.lib().hilo_debug_kf_prof(out)
    s = [int(v) for v in out]
    names = ['model / sigma points', 'P- / means', 'Pxy Pyy / entries', 'gain rows', 'P update']
    print(kind, 'last step, cycles:', {n: s[i + 1] - s[i] for i, n in enumerate(names)}, 'step', s[9] - s[8], 'stores', s[10] - s[9],
          'loop iteration', s[10] - s[8])
