"""Debug aid: host time of one estimate(steps=K) call (the launch is asynchronous) and where it goes."""
import cProfile
import pstats
import sys
import time
import numpy as np
import torch
sys.path.insert(0, '.')
import hilo_mpc_amd as H                                                  # noqa: E402

B, K = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
x = np.array([.1, 40., .5, .2]) * (1 + .1 * rng.uniform(-1, 1, (B, 4)))
f = H.EKF(H.Model('chemostat4').discretize('rk4').setup(dt=1.))
f.setup()
f.Q, f.R = 1e-4, 1e-2
f.set_initial_guess(torch.as_tensor(x, device=dev), P0=torch.as_tensor(np.tile(np.eye(4), (B, 1, 1)), device=dev))
u = torch.as_tensor(rng.uniform(0, .3, (B, 2)), device=dev)
p = torch.as_tensor(np.tile([100., 4., 1., 0.], (B, 1)), device=dev)
y = torch.as_tensor(x[:, [0, 2]], device=dev)[None] * (1 + torch.as_tensor(.02 * rng.normal(size=(K, B, 2)), device=dev))
call = (lambda: f.estimate(y=y, u=u, p=p, steps=K)) if K > 1 else (lambda: f.estimate(y=y[0], u=u, p=p))
for _ in range(20):
    call()
torch.cuda.synchronize()
n = 500
t = time.perf_counter()
for _ in range(n):
    call()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'K={K}: host {1e6 * (t1 - t) / n:.1f} us per call, with drain {1e6 * (t2 - t) / n:.1f} us per call')
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    call()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(12)
