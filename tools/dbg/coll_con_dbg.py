"""Developer aid: chemostat4 under collocation with stage constraints, product vs oracle per block of lam_g."""
import sys
import numpy as np
sys.path.insert(0, '.')
from oracle.nmpc import IpmOptions
from oracle.nmpc_coll_gen import GenCollIpm
from tests.problems import C2, c2_x0, oracle_coll_gen, product_gen

obj = sys.argv[1] if len(sys.argv) > 1 else 'discrete'
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 2
tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-9
con = int(sys.argv[4]) if len(sys.argv) > 4 else 1
kw = dict(C2, N=8, collocation=dict(degree=deg, objective=obj))
kw.pop('order', None)
if con:
    kw['constraint'] = dict(expr=['X * S', 'S - X'], lb=[-np.inf, 0.], ub=[60., np.inf])
x0 = c2_x0(4)
pb = oracle_coll_gen(kw)
ipm = GenCollIpm(pb, IpmOptions(tol=tol))
ref = ipm.solve(x0, C2['p'])
nmpc = product_gen(kw, **{'ipopt.tol': tol})
u = nmpc.optimize(x0, cp=C2['p'])
print('status', nmpc.solver_status_code, ref['status'], 'iters', nmpc.stats()['iter_count'], ref['iters'])
v, vr = nmpc._nlp_solution['x'].cpu().numpy(), ipm.to_v(ref)
print('v rel', (np.abs(v - vr) / np.maximum(1., np.abs(vr))).max(1), 'f', nmpc._nlp_solution['f'].cpu().numpy() - ref['f'])
lam, lr = nmpc._nlp_solution['lam_g'].cpu().numpy(), ipm.lam_g(ref)
N, d, R, nxa = pb.N, pb.d, pb.n_con_ref, pb.nxa
per = lr.shape[1] // N
L, Lr = lam.reshape(4, N, per), lr.reshape(4, N, per)
dl = np.abs(L - Lr) / np.maximum(1., np.abs(Lr))
print('lam rel per instance', dl.reshape(4, -1).max(1))
print('blocks: coll rows', dl[:, :, :d * R].max() if R else 0, 'coll eq', dl[:, :, d * R:d * R + d * nxa].max(), 'cont', dl[:, :, d * R + d * nxa:d * R + (d + 1) * nxa].max(),
      'node', dl[:, :, per - R:].max() if R else 0)
b = int(np.argmax(dl.reshape(4, -1).max(1)))
k = int(np.argmax(dl[b].max(1)))
np.set_printoptions(precision=8, linewidth=220)
print('worst', b, k)
print(L[b, k])
print(Lr[b, k])
print((L - Lr)[b, k])
