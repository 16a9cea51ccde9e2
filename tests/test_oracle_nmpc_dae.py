"""The DAE extension of the collocation oracle (oracle/nmpc_dae.py: the reference's simultaneous form, algebraic states at the
collocation points as variables) against the ODE oracle on problems whose algebraic state can be eliminated by hand."""
import numpy as np

from oracle import models
from oracle.nmpc_coll import CollIpm, CollNmpcProblem
from oracle.nmpc_dae import DaeCollIpm, DaeCollProblem
from tests.problems import C2, c2_x0

PEND = dict(dt=.1, N=6, stage_states=[([1, 2], [10., 5.], [0., 0.])], stage_inputs=[([0], [.1], None)],
            x_lb=[-5, -10, -10, -10], x_ub=[5, 10, 10, 10], x_guess=[2.5, 0., .1, 0.], u_guess=[0.])


def test_layout_of_the_reference_with_algebraic_states():
    pb = DaeCollProblem(models.get('pendulum4_dae'), z_guess=[1.4], **PEND)
    N, nx, nu, nz, d = 6, 4, 1, 1, 3
    assert pb.n_v == (N + 1) * nx + N * nu + (N + 1) * nz + N * d * (nx + nz)           # mpc.py:1440-1445
    assert pb.n_g == N * (d * (nx + nz) + nx)
    assert pb.z_ind[0] == [(N + 1) * nx + N * nu] and pb.ip_ind[0][0] == pb.z_ind[-1][-1] + 1
    assert pb.zp_ind[0] == list(range(pb.ip_ind[0][-1] + 1, pb.ip_ind[0][-1] + 1 + d * nz))     # mpc.py:1510-1518
    assert pb.ip_ind[1][0] == pb.zp_ind[0][-1] + 1


def test_output_like_algebraic_state_leaves_the_optimum_alone():
    """tests/test_NMPC.py:1866-1911: the algebraic state (height of the pendulum tip) feeds nothing back."""
    x0 = np.array([[2.5, 0., .1, 0.]])
    pb = DaeCollProblem(models.get('pendulum4_dae'), z_guess=[1.4], **PEND)
    ipm = DaeCollIpm(pb)
    r = ipm.solve(x0, [])
    r2 = CollIpm(CollNmpcProblem(models.get('pendulum4'), objective='continuous', **PEND)).solve(x0, [])
    assert r['status'][0] == r2['status'][0] == 1
    np.testing.assert_allclose(r['X'], r2['X'], rtol=0, atol=1e-10)
    np.testing.assert_allclose(r['Zc'][0, :, :, 0], .5 + np.cos(r['Xc'][0, :, :, 2]), rtol=0, atol=1e-12)
    lam = r['lam'].reshape(1, pb.N, -1)
    assert np.abs(lam[:, :, [4, 9, 14]]).max() < 1e-9                                  # the algebraic rows carry no force
    v = ipm.to_v(r)
    assert v.shape[1] == pb.n_v and np.all(v[0, [i for ind in pb.z_ind for i in ind]] == 1.4)


def test_algebraic_state_that_feeds_back_equals_the_eliminated_ode():
    """chemostat4 with the growth rate as algebraic state: same trajectories as chemostat4, z at the collocation points equals
    the closed-form rate, and the multipliers of the algebraic rows satisfy dt f_z^T mu + g_z^T nu = 0."""
    spec = {k: v for k, v in dict(C2, N=5).items() if k not in ('model', 'p', 'order')}
    x0 = c2_x0(2)
    pb = DaeCollProblem(models.get('chemostat4_dae'), z_guess=[.3], **spec)
    r = DaeCollIpm(pb).solve(x0, C2['p'])
    r2 = CollIpm(CollNmpcProblem(models.get('chemostat4'), objective='continuous', **spec)).solve(x0, C2['p'])
    assert np.all(r['status'] == 1) and np.all(r2['status'] == 1)
    np.testing.assert_allclose(r['X'], r2['X'], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(r['U'], r2['U'], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(r['f'], r2['f'], rtol=1e-10)
    S, I = r['Xc'][..., 1] * pb.sx[1], r['Xc'][..., 3] * pb.sx[3]
    p = np.asarray(C2['p'])
    mu = 0.407 * S / (0.108 + S + S ** 2 / 14814.0) * (p[2] + 0.22 * p[3] / (0.22 + I))
    np.testing.assert_allclose(r['Zc'][..., 0], mu, rtol=1e-9)
    lam = r['lam'].reshape(2, pb.N, pb.d, 5 + 0)[:, :, :, :] if False else r['lam'].reshape(2, pb.N, -1)
    Xq = r['Xc'][..., 0] * pb.sx[0]
    for i in range(pb.d):
        m_ode = lam[:, :, i * 5:i * 5 + 4]                                              # multipliers of the scaled ode rows
        nu = lam[:, :, i * 5 + 4]
        fz = np.stack([Xq[:, :, i], -2 * Xq[:, :, i], 0 * Xq[:, :, i], 0 * Xq[:, :, i]], -1) / pb.sx     # d f_s / d mu
        np.testing.assert_allclose(nu, -pb.dt * np.sum(fz * m_ode, -1), rtol=1e-6, atol=1e-10)          # g_z = 1
