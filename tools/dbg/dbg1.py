import sys; sys.path.insert(0, '.')
import numpy as np
from tests.problems import product_nmpc, symbolic_model
spec = dict(model='pendulum4', dt=.1, N=15, order=4, stage_states=[([0, 2], [10., 5.], [0., 0.])],
            stage_inputs=[([0], [.1], None)], terminal_states=[([0, 2], [10., 5.], [0., 0.])],
            u_lb=[-20.], u_ub=[20.], x_guess=[0., 0., 0., 0.], u_guess=[0.], p=[])
x0 = np.array([.5, 0., .3, 0.]) * (1 + .2 * np.random.default_rng(1).uniform(-1, 1, (16, 4)))
zoo = product_nmpc(spec); sym = product_nmpc(spec, model=symbolic_model('pendulum4'))
u = np.random.default_rng(3).uniform(-5, 5, (16, 1))
a = zoo.plant_step(x0, u).cpu().numpy(); b = sym.plant_step(x0, u).cpu().numpy()
print('plant equal', np.array_equal(a, b), np.abs(a - b).max())
for mi in (1, 2, 3):
    z = product_nmpc(spec, max_iter=mi); s = product_nmpc(spec, model=symbolic_model('pendulum4'), max_iter=mi)
    z.optimize(x0); s.optimize(x0)
    va, vb = z._nlp_solution['x'].cpu().numpy(), s._nlp_solution['x'].cpu().numpy()
    print(mi, np.array_equal(va, vb), np.abs(va - vb).max(), np.array_equal(z._nlp_solution['f'].cpu().numpy(), s._nlp_solution['f'].cpu().numpy()))
print(sym._user_source)
