"""Developer aid: C5-DAE (tests/problems.py::C5DS) product vs oracle, printed differences per block."""
import sys
import numpy as np
sys.path.insert(0, '.')
from oracle.nmpc import IpmOptions
from oracle.nmpc_coll_gen import GenCollIpm
from tests.problems import C5DS, c5_x0, oracle_coll_gen, product_gen

tol = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-10
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N = int(sys.argv[3]) if len(sys.argv) > 3 else 10
spec = dict(C5DS, N=N, collocation=dict(degree=deg))
x0 = c5_x0(8)
pb = oracle_coll_gen(spec)
ipm = GenCollIpm(pb, IpmOptions(tol=tol))
ref = ipm.solve(x0, [])
nmpc = product_gen(spec, **{'ipopt.tol': tol})
u = nmpc.optimize(x0)
st = nmpc.stats()
print('status', nmpc.solver_status_code, ref['status'])
print('iters', st['iter_count'], ref['iters'])
print('kkt', st['kkt_error'], ref['kkt'])
v, vr = nmpc._nlp_solution['x'].cpu().numpy(), ipm.to_v(ref)
rel = np.abs(v - vr) / np.maximum(1., np.abs(vr))
print('f', nmpc._nlp_solution['f'].cpu().numpy() - ref['f'], ref['f'])
nxa, nua, nzg, d = pb.nxa, pb.nua, pb.nzalg, pb.d
o1 = (N + 1) * nxa
o2 = o1 + N * nua
o3 = o2 + (N + 1) * nzg
print('v: x', rel[:, :o1].max(), 'u', rel[:, o1:o2].max(), 'z nodes', rel[:, o2:o3].max(), 'blocks', rel[:, o3:-pb.ne].max(), 'e', rel[:, -pb.ne:].max())
print('per-instance v rel', rel.max(1))
print('e', v[:, -pb.ne:].ravel(), vr[:, -pb.ne:].ravel())
lam, lr = nmpc._nlp_solution['lam_g'].cpu().numpy(), ipm.lam_g(ref)
per = lr.shape[1] // N
L, Lr = lam.reshape(8, N, per), lr.reshape(8, N, per)
R = pb.n_con_ref
dl = np.abs(L - Lr) / np.maximum(1., np.abs(Lr))
print('per-instance lam rel', dl.reshape(8, -1).max(1))
print('lam: coll-point rows', dl[:, :, :d * R].max(), 'coll eq', dl[:, :, d * R:d * R + d * (nxa + nzg)].max(), 'cont',
      dl[:, :, d * R + d * (nxa + nzg):d * R + d * (nxa + nzg) + nxa].max(), 'node rows', dl[:, :, -R:].max())
b, k = np.unravel_index(np.argmax(dl.reshape(8, -1).max(1)), (8,)), 0
bb = int(np.argmax(dl.reshape(8, -1).max(1)))
kk = int(np.argmax(dl[bb].max(1)))
np.set_printoptions(precision=6, linewidth=200, suppress=False)
print('worst instance', bb, 'interval', kk)
print(L[bb, kk])
print(Lr[bb, kk])
