"""The partial-condensing prototype of tools/prototypes/partial_condensing.py (next step of the solve kernel):
condensing M stages into one and running the Riccati recursion on the shorter horizon gives the states, inputs and multipliers
of the full-horizon KKT system (dense solve) - for the headline dimensions and every block size dividing the horizon."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'prototypes'))
import partial_condensing as pc     # noqa: E402


def _problem(N, nx, nu, seed):
    rng = np.random.default_rng(seed)
    A = [np.eye(nx) + .3 * rng.standard_normal((nx, nx)) for _ in range(N)]
    B = [rng.standard_normal((nx, nu)) for _ in range(N)]
    c = [.1 * rng.standard_normal(nx) for _ in range(N)]
    H = []
    for _ in range(N):
        W = rng.standard_normal((nx + nu, nx + nu))
        H.append(W @ W.T + .5 * np.eye(nx + nu))           # positive definite with a cross term S
    g = [rng.standard_normal(nx + nu) for _ in range(N)]
    W = rng.standard_normal((nx, nx))
    return A, B, c, H, g, W @ W.T + np.eye(nx), rng.standard_normal(nx), rng.standard_normal(nx)


def _dense_kkt(A, B, c, H, g, PN, pN, x0):
    N, (nx, nu) = len(A), B[0].shape
    nz = nx + nu
    nv = N * nz + nx                    # [x_0 u_0 | x_1 u_1 | ... | x_N]
    Hs, gs = np.zeros((nv, nv)), np.zeros(nv)
    for k in range(N):
        Hs[k * nz:(k + 1) * nz, k * nz:(k + 1) * nz] = H[k]
        gs[k * nz:(k + 1) * nz] = g[k]
    Hs[N * nz:, N * nz:], gs[N * nz:] = PN, pN
    m = (N + 1) * nx
    C, rhs = np.zeros((m, nv)), np.zeros(m)
    C[:nx, :nx], rhs[:nx] = np.eye(nx), x0                 # x_0 = given        (multiplier lam_0)
    for k in range(N):                                     # x_{k+1} - A x_k - B u_k = c   (multiplier lam_{k+1})
        r = (k + 1) * nx
        C[r:r + nx, k * nz:k * nz + nx] = -A[k]
        C[r:r + nx, k * nz + nx:(k + 1) * nz] = -B[k]
        C[r:r + nx, (k + 1) * nz:(k + 1) * nz + nx] = np.eye(nx)
        rhs[r:r + nx] = c[k]
    KKT = np.block([[Hs, C.T], [C, np.zeros((m, m))]])
    sol = np.linalg.solve(KKT, np.concatenate([-gs, rhs]))
    v, lam = sol[:nv], -sol[nv:]                           # stationarity H v + g - C^T lam = 0
    X = [v[k * nz:k * nz + nx] for k in range(N)] + [v[N * nz:]]
    U = [v[k * nz + nx:(k + 1) * nz] for k in range(N)]
    return X, U, [lam[k * nx:(k + 1) * nx] for k in range(N + 1)]


@pytest.mark.parametrize('N,nx,nu,M', [(20, 4, 2, 1), (20, 4, 2, 2), (20, 4, 2, 4), (20, 4, 2, 5), (12, 6, 2, 3), (8, 3, 1, 8)])
def test_partial_condensing_equals_the_dense_kkt_solution(N, nx, nu, M):
    A, B, c, H, g, PN, pN, x0 = _problem(N, nx, nu, seed=N + M)
    Xd, Ud, ld = _dense_kkt(A, B, c, H, g, PN, pN, x0)
    Xr, Ur, lr = pc.riccati(A, B, c, H, g, PN, pN, x0)
    Xc, Uc, lc = pc.solve_partially_condensed(A, B, c, H, g, PN, pN, x0, M)
    for got in ((Xr, Ur, lr), (Xc, Uc, lc)):
        np.testing.assert_allclose(np.array(got[0]), np.array(Xd), rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(np.array(got[1]), np.array(Ud), rtol=1e-8, atol=1e-9)
    # multipliers: lam_k = d cost-to-go / d x_k (sign convention of the engine: Riccati's P_k x_k + p_k)
    np.testing.assert_allclose(np.array(lr[1:]), np.array(ld[1:]), rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(np.array(lc[1:]), np.array(ld[1:]), rtol=1e-7, atol=1e-8)
