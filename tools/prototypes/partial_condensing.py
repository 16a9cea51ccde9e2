"""Prototype (numpy) for the next step of the solve kernel: PARTIAL CONDENSING of the stage-structured KKT system.

The interior-point engine (csrc/hilo_ocp.h) solves, per iteration, the equality-constrained QP

    min  sum_k 1/2 [x_k; u_k]^T H_k [x_k; u_k] + g_k^T [x_k; u_k]  +  1/2 x_N^T P_N x_N + p_N^T x_N
    s.t. x_{k+1} = A_k x_k + B_k u_k + c_k,   x_0 given

by a Riccati recursion of depth N (N = 20 for the headline configuration): 46 % of the kernel's cycles are this serial chain,
every link four LDS round trips, two dependent MFMAs and a Cholesky of an n_u x n_u block that fills a corner of a 16-wide tile.
Condensing M consecutive stages into one - eliminating the M - 1 interior states of each block - gives an equivalent problem of
horizon N / M with n_u M inputs per stage: for n_x = 4, n_u = 2, M = 4 the recursion is 5 links deep, its blocks (n_x + M n_u =
12 columns, Cholesky 8 x 8) fit the 16-wide matrix-core tile, and the condensing itself (products of the A_k inside a block,
O(M^2) small products) is independent per block - work for the lanes that idle during the recursion today.

This file states the algorithm and checks it (tests/test_partial_condensing.py) against the dense KKT solve:
    condense(...)   block problem  (A~, B~, c~, H~, g~)  from M stages
    riccati(...)    the recursion the engine runs (any horizon)
    expand(...)     interior states by the dynamics, multipliers lambda_k by the adjoint recursion inside each block
"""
import numpy as np


def riccati(A, B, c, H, g, PN, pN, x0):
    """Backward Riccati recursion + forward roll-out.  H[k] = [[Q, S^T], [S, R]] on (x, u).  Returns X [N+1], U [N], lam [N+1]
    (lam_k: multiplier of x_k's defining equation, lam_k = P_k x_k + p_k)."""
    N = len(A)
    nx = A[0].shape[0]
    P, p = [None] * (N + 1), [None] * (N + 1)
    K, kff = [None] * N, [None] * N
    P[N], p[N] = PN, pN
    for k in range(N - 1, -1, -1):
        Q, S, R = H[k][:nx, :nx], H[k][nx:, :nx], H[k][nx:, nx:]
        q, r = g[k][:nx], g[k][nx:]
        Pc = P[k + 1] @ c[k] + p[k + 1]
        Ruu = R + B[k].T @ P[k + 1] @ B[k]
        Rux = S + B[k].T @ P[k + 1] @ A[k]
        L = np.linalg.cholesky(Ruu)
        sol = lambda M_: np.linalg.solve(L.T, np.linalg.solve(L, M_))
        K[k] = -sol(Rux)
        kff[k] = -sol(r + B[k].T @ Pc)
        P[k] = Q + A[k].T @ P[k + 1] @ A[k] + Rux.T @ K[k]
        P[k] = .5 * (P[k] + P[k].T)
        p[k] = q + A[k].T @ Pc + Rux.T @ kff[k]
    X, U = [x0], []
    for k in range(N):
        U.append(K[k] @ X[k] + kff[k])
        X.append(A[k] @ X[k] + B[k] @ U[k] + c[k])
    lam = [P[k] @ X[k] + p[k] for k in range(N + 1)]
    return X, U, lam


def condense(A, B, c, H, g):
    """One block of M stages -> one stage with the stacked input U = [u_0; ..; u_{M-1}]:
        x_j   = Phi_j x_0 + Gam_j U + d_j      (j = 0..M, Gam_j lower block triangular)
        x_M   = A~ x_0 + B~ U + c~
        cost  = 1/2 [x_0; U]^T H~ [x_0; U] + g~^T [x_0; U] + const
    Returns A~, B~, c~, H~, g~ and the maps (Phi, Gam, d) for the expansion."""
    M = len(A)
    nx, nu = B[0].shape
    Phi, Gam, d = [np.eye(nx)], [np.zeros((nx, M * nu))], [np.zeros(nx)]
    for j in range(M):
        G = A[j] @ Gam[j]
        G[:, j * nu:(j + 1) * nu] += B[j]
        Phi.append(A[j] @ Phi[j])
        Gam.append(G)
        d.append(A[j] @ d[j] + c[j])
    nz = nx + M * nu
    Ht, gt = np.zeros((nz, nz)), np.zeros(nz)
    for j in range(M):
        # [x_j; u_j] = T_j [x_0; U] + t_j
        T = np.zeros((nx + nu, nz))
        T[:nx, :nx], T[:nx, nx:] = Phi[j], Gam[j]
        T[nx:, nx + j * nu:nx + (j + 1) * nu] = np.eye(nu)
        t = np.concatenate([d[j], np.zeros(nu)])
        Ht += T.T @ H[j] @ T
        gt += T.T @ (H[j] @ t + g[j])
    return Phi[M], Gam[M], d[M], Ht, gt, (Phi, Gam, d)


def solve_partially_condensed(A, B, c, H, g, PN, pN, x0, M):
    """The full-horizon solution through the condensed problem of horizon N / M."""
    N = len(A)
    assert N % M == 0
    nx, nu = B[0].shape
    blocks = [condense(A[b:b + M], B[b:b + M], c[b:b + M], H[b:b + M], g[b:b + M]) for b in range(0, N, M)]
    Xb, Ub, lamb = riccati([q[0] for q in blocks], [q[1] for q in blocks], [q[2] for q in blocks], [q[3] for q in blocks],
                           [q[4] for q in blocks], PN, pN, x0)
    X, U, lam = [x0], [], [None] * (N + 1)
    for i, (blk, b) in enumerate(zip(blocks, range(0, N, M))):
        Phi, Gam, d = blk[5]
        for j in range(M):
            U.append(Ub[i][j * nu:(j + 1) * nu])
        for j in range(1, M + 1):
            X.append(Phi[j] @ Xb[i] + Gam[j] @ Ub[i] + d[j])
        # multipliers inside the block: lam_{k} = Q x_k + S^T u_k + q_k + A_k^T lam_{k+1}, started from the block's end
        lam[b + M] = lamb[i + 1]
        for j in range(M - 1, 0, -1):
            k = b + j
            lam[k] = H[k][:nx, :nx] @ X[k] + H[k][nx:, :nx].T @ U[k] + g[k][:nx] + A[k].T @ lam[k + 1]
    lam[0] = lamb[0]
    return X, U, lam
