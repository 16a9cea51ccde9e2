"""Host time of `EKF.estimate(steps=16)` at the configured batch (B = 4096): cProfile over 3000 calls, and the wall time per call
with and without `inputs_unchanged=True`.      python tools/dbg/kf_host_profile.py [ekf|ukf]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hilo_mpc_amd import EKF, UKF, Model      # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else 'ekf'
B, K, dev = 4096, 16, torch.device('cuda:0')
rng = np.random.default_rng(1)
x = np.array([.1, 40., .5, .2]) * (1 + .1 * rng.uniform(-1, 1, (B, 4)))
f = (EKF if kind == 'ekf' else UKF)(Model('chemostat4').discretize('rk4').setup(dt=1.))
f.setup()
f.Q, f.R = 1e-4, 1e-2
f.set_initial_guess(torch.as_tensor(x, device=dev), P0=torch.as_tensor(np.tile(np.eye(4), (B, 1, 1)), device=dev))
u = torch.as_tensor(rng.uniform(0, .3, (B, 2)), device=dev)
p = torch.as_tensor(np.tile([100., 4., 1., 0.], (B, 1)), device=dev)
y = torch.as_tensor(x[:, [0, 2]], device=dev)[None] * (1 + torch.as_tensor(.02 * rng.normal(size=(K, B, 2)), device=dev))
for kw in ({}, {'inputs_unchanged': True}):
    for _ in range(50):
        f.estimate(y=y, u=u, p=p, steps=K, **kw)
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 2000
    for _ in range(n):
        f.estimate(y=y, u=u, p=p, steps=K, **kw)
    t_host = time.perf_counter() - t
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t
    print(kind, kw, f'host {t_host / n * 1e6:.1f} us/call, with the device {t_all / n * 1e6:.1f} us/call')
pr = cProfile.Profile()
pr.enable()
for _ in range(3000):
    f.estimate(y=y, u=u, p=p, steps=K)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
