// The LMPC's QP solved stage by stage: the same Mehrotra predictor-corrector iteration as csrc/hilo_qp.hip (start, step rule,
// merit, tolerances, termination), with the Newton step taken by a Riccati recursion over the stages instead of the dense
// Schur complement.  Included by hilo_qp.hip.
//
// Reference: `LMPC.setup` / `LMPC.optimize`, hilo_mpc/modules/controller/mpc.py:2198-2266, :2307-2394 - the QP
//     v = [x_0 .. x_N | u_0 .. u_{N-1}]       H = blkdiag(Q .. Q, P, R .. R)  (:2252-2256)
//     rows k:  A_k x_k + B_k u_k - x_{k+1} = b_k  (:2209-2245, time-varying :2200-2206, :2236-2240; the parameter-free branch's
//     input block `kron(B, I_N)` (:2243) does NOT have this shape and stays on the dense kernels),  x_0 pinned by lbx == ubx (:2361-2362)
// A 32-variable QP (BASELINE configuration 1: nx = 2, nu = 1, N = 10) is a chain of 10 dependent 2x2 / 1x1 steps: the dense
// kernel spends 29 us per iteration on 32x32 and 20x20 factorisations a CPU core does in 13 us, the recursion below needs ~4 us.
//
// One STAGE per lane, G = 16 (N <= 15) or 64 lanes per instance; the stage's blocks, iterate and multipliers live in registers.
// The recursion runs N uniform iterations in which every lane evaluates its stage against the cost-to-go handed over from the
// neighbouring lane (data-parallel primitive row_shl:1 / row_shr:1 inside a row of 16 lanes, a lane permute for G = 64) and only
// the stage whose turn it is keeps the result.  Reductions (merit, step lengths) are row reductions.  Nothing touches LDS.
#pragma once

namespace hilo {

#ifdef HILO_QPO_PROF   // developer builds: cycle stamps of one iteration of workgroup 0 (tools/dbg/qp_stage_phase.py)
__device__ unsigned long long hilo_qpo_prof[16];
#define QPO_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) hilo_qpo_prof[k] = __builtin_readcyclecounter(); } while (0)
#else
#define QPO_STAMP(k) do { } while (0)
#endif

// 1 / x: v_rcp_f64 + two Newton steps (<= 1 ulp) instead of the 12-instruction IEEE division sequence; the slacks' reciprocals
// are formed once per iteration and shared by Sigma, the corrector's right-hand side and the step rule
__device__ __forceinline__ double ocp_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

template <int G>
__device__ __forceinline__ double ocp_from_next(double v) {   // lane k <- lane k + 1
  if constexpr (G == 16) return qp_dpp<0x101, 0xf>(v, v);     // row_shl:1
  else return __shfl_down(v, 1, 64);
}
template <int G>
__device__ __forceinline__ double ocp_from_prev(double v) {   // lane k <- lane k - 1
  if constexpr (G == 16) return qp_dpp<0x111, 0xf>(v, v);     // row_shr:1
  else return __shfl_up(v, 1, 64);
}
template <int G, class Op>
__device__ __forceinline__ double ocp_reduce(double v, double ident, Op op) {
  if constexpr (G == 16) {
    v = op(v, qp_dpp<0xB1, 0xf>(v, ident));     // quad_perm [1, 0, 3, 2]
    v = op(v, qp_dpp<0x4E, 0xf>(v, ident));     // quad_perm [2, 3, 0, 1]
    v = op(v, qp_dpp<0x141, 0xf>(v, ident));    // row_half_mirror
    v = op(v, qp_dpp<0x140, 0xf>(v, ident));    // row_mirror: every lane of the row holds the row's result
    return v;
  } else {
    return qp_wave_reduce(v, ident, op);
  }
}
template <int G> __device__ __forceinline__ double ocp_sum(double v) { return ocp_reduce<G>(v, 0.0, [](double a, double b) { return a + b; }); }
template <int G> __device__ __forceinline__ double ocp_min(double v) { return ocp_reduce<G>(v, INFINITY, [](double a, double b) { return fmin(a, b); }); }
template <int G> __device__ __forceinline__ double ocp_max(double v) { return ocp_reduce<G>(v, -INFINITY, [](double a, double b) { return fmax(a, b); }); }

// One step of a scan over the affine maps v -> Mat v + d the lanes of a row hold (Hillis-Steele): compose this lane's map with
// the one CTRL lanes away (row_shl:o - the stages behind, row_shr:o - the stages before; beyond the row: the identity).
template <int NX, int CTRL>
__device__ __forceinline__ void ocp_scan_step(double (&Mat)[NX][NX], double (&d)[NX]) {
  double Mo[NX][NX], dn[NX], Mn[NX][NX], dd[NX];
#pragma unroll
  for (int a = 0; a < NX; ++a) {
    dn[a] = qp_dpp<CTRL, 0xf>(d[a], 0.0);
#pragma unroll
    for (int c = 0; c < NX; ++c) Mo[a][c] = qp_dpp<CTRL, 0xf>(Mat[a][c], a == c ? 1.0 : 0.0);
  }
#pragma unroll
  for (int a = 0; a < NX; ++a) {
    double s = d[a];
#pragma unroll
    for (int e = 0; e < NX; ++e) s += Mat[a][e] * dn[e];
    dd[a] = s;
#pragma unroll
    for (int c = 0; c < NX; ++c) {
      double q = 0.0;
#pragma unroll
      for (int e = 0; e < NX; ++e) q += Mat[a][e] * Mo[e][c];
      Mn[a][c] = q;
    }
  }
#pragma unroll
  for (int a = 0; a < NX; ++a) {
    d[a] = dd[a];
#pragma unroll
    for (int c = 0; c < NX; ++c) Mat[a][c] = Mn[a][c];
  }
}

// Arguments of qp_solve_kernel (dense H [n][n], A [m][n] per instance or shared) + the horizon.  n = (N+1) NX + N NU, m = N NX.
template <int NX, int NU, int G>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void qp_ocp_kernel(
    QpDims qd, int N, int64_t batch, const double* __restrict__ Hg, int64_t hs, const double* __restrict__ gg, int64_t gs,
    const double* __restrict__ Ag, int64_t as_, const double* __restrict__ lbx, const double* __restrict__ ubx, int64_t bs,
    const double* __restrict__ lba, const double* __restrict__ uba, int64_t bas, double* __restrict__ x_out,
    double* __restrict__ f_out, double* __restrict__ lam_a, double* __restrict__ lam_x, int32_t* __restrict__ status,
    int32_t* __restrict__ iters, const double* __restrict__ xpin, int npin, int64_t ps) {
  constexpr int IPW = 64 / G;
  const int n = qd.n, m = qd.m;
  const int grp = threadIdx.x / G, k = threadIdx.x - grp * G;          // stage of this lane
  const int64_t b0 = (int64_t)blockIdx.x * IPW + grp;
  const bool valid = b0 < batch;
  const int64_t b = valid ? b0 : batch - 1;
  const bool stage = k <= N, inner = k < N;                            // lanes beyond the horizon idle along
  const int kc = stage ? k : N;                                        // (they repeat the last stage and store nothing)
  const int xo = kc * NX, uo = (N + 1) * NX + (inner ? k : 0) * NU, ro = (inner ? k : 0) * NX;
  const double* H = Hg + b * hs;
  const double* A = Ag + b * as_;
  const double* g = gg + b * gs;

  // ---- this stage's blocks ----
  double Q[NX][NX], R[NU][NU], Ak[NX][NX], Bk[NX][NU], gx[NX], gu[NU], bf[NX];
  double xs[NX], us[NU], lx[NX], ux[NX], lu[NU], uu[NU], zlx[NX], zux[NX], zlu[NU], zuu[NU], y[NX];
  int bad = 0;
#pragma unroll
  for (int i = 0; i < NX; ++i) {
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      Q[i][j] = H[(int64_t)(xo + i) * n + xo + j];
      Ak[i][j] = inner ? A[(int64_t)(ro + i) * n + xo + j] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < NU; ++j) Bk[i][j] = inner ? A[(int64_t)(ro + i) * n + uo + j] : 0.0;
    gx[i] = g[xo + i];
    qp_bounds(lbx, ubx, bs, xpin, npin, ps, b, xo + i, lx[i], ux[i]);
    const bool fx = lx[i] == ux[i];
    if (stage && (fx != (k == 0))) bad = 1;            // x_0 pinned, nothing else: the shape this kernel is built for
    if (!stage) { lx[i] = -INFINITY; ux[i] = INFINITY; }
    bf[i] = inner ? uba[b * bas + ro + i] : 0.0;
    if (inner && lba[b * bas + ro + i] != uba[b * bas + ro + i]) bad = 1;
    y[i] = 0.0;
  }
#pragma unroll
  for (int i = 0; i < NU; ++i) {
#pragma unroll
    for (int j = 0; j < NU; ++j) R[i][j] = inner ? H[(int64_t)(uo + i) * n + uo + j] : (i == j ? 1.0 : 0.0);
    gu[i] = inner ? g[uo + i] : 0.0;
    lu[i] = -INFINITY; uu[i] = INFINITY;
    if (inner) qp_bounds(lbx, ubx, bs, xpin, npin, ps, b, uo + i, lu[i], uu[i]);
    if (inner && lu[i] == uu[i]) bad = 1;
  }
  const bool first = k == 0;
  // the pinned x_0: substituted like the dense kernels do (its columns leave A, b_0 takes A_0 x_0; H is block diagonal)
  double x0v[NX], A0[NX][NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    x0v[i] = lx[i];
#pragma unroll
    for (int j = 0; j < NX; ++j) A0[i][j] = Ak[i][j];
  }
  if (first) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      double s = bf[i];
#pragma unroll
      for (int j = 0; j < NX; ++j) { s -= Ak[i][j] * x0v[j]; }
      bf[i] = s;
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
#pragma unroll
      for (int j = 0; j < NX; ++j) Ak[i][j] = 0.0;
      gx[i] = 0.0;
      lx[i] = -INFINITY;
      ux[i] = INFINITY;
    }
  }
  // starting point: strictly inside the box, unit multipliers (hilo_qp.hip)
  double nbp = 0.0, gmx = 0.0;
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const bool hl = stage && !first && lx[i] > -INFINITY, hu = stage && !first && ux[i] < INFINITY;
    double v = 0.0;
    if (hl && hu) v = 0.5 * (lx[i] + ux[i]);
    else if (hl) v = fmax(0.0, lx[i] + 1.0);
    else if (hu) v = fmin(0.0, ux[i] - 1.0);
    xs[i] = v;
    zlx[i] = hl ? 1.0 : 0.0;
    zux[i] = hu ? 1.0 : 0.0;
    nbp += (hl ? 1.0 : 0.0) + (hu ? 1.0 : 0.0);
    if (stage && !first) gmx = fmax(gmx, fabs(gx[i]));
  }
#pragma unroll
  for (int i = 0; i < NU; ++i) {
    const bool hl = inner && lu[i] > -INFINITY, hu = inner && uu[i] < INFINITY;
    double v = 0.0;
    if (hl && hu) v = 0.5 * (lu[i] + uu[i]);
    else if (hl) v = fmax(0.0, lu[i] + 1.0);
    else if (hu) v = fmin(0.0, uu[i] - 1.0);
    us[i] = v;
    zlu[i] = hl ? 1.0 : 0.0;
    zuu[i] = hu ? 1.0 : 0.0;
    nbp += (hl ? 1.0 : 0.0) + (hu ? 1.0 : 0.0);
    if (inner) gmx = fmax(gmx, fabs(gu[i]));
  }
  const double nb = fmax(1.0, ocp_sum<G>(nbp));
  const double gmax = ocp_max<G>(gmx);
  bad = ocp_max<G>(bad ? 1.0 : 0.0) > 0.0;

  int st = bad ? HILO_STATUS_OTHER : HILO_STATUS_MAXITER, it = 0;
  bool done = bad;
  double phi_min = INFINITY;
  double P[NX][NX], Pn[NX][NX], K[NU][NX], Qux[NU][NX], Qi[NU][NU], pv[NX], kff[NU], dx[NX], du[NU], dy[NX];
  double dzlx[NX], dzux[NX], dzlu[NU], dzuu[NU], basex[NX], baseu[NU], rp[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) { dx[i] = dzlx[i] = dzux[i] = 0.0; }
#pragma unroll
  for (int i = 0; i < NU; ++i) { du[i] = dzlu[i] = dzuu[i] = 0.0; }

  for (int iter = 0; iter < qd.max_iter; ++iter) {
    if (!__any(!done)) break;
    QPO_STAMP(0);
    // ---- residuals: base = -(H v + g + A^T y), rp = A v - b, mu ----
    double rdm = 0.0, mup = 0.0, nonf = 0.0, rpm = 0.0;
    double yprev[NX], xnext[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) { yprev[i] = ocp_from_prev<G>(y[i]); xnext[i] = ocp_from_next<G>(xs[i]); }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      double s = gx[i];
#pragma unroll
      for (int j = 0; j < NX; ++j) s += Q[i][j] * xs[j];
#pragma unroll
      for (int r = 0; r < NX; ++r) s += Ak[r][i] * y[r];
      s -= yprev[i];
      if (first || !stage) s = 0.0;
      basex[i] = -s;
      const double rd = s - zlx[i] + zux[i];
      rdm = fmax(rdm, fabs(rd));
      nonf += isfinite(rd) ? 0.0 : 1.0;
      if (lx[i] > -INFINITY) mup += (xs[i] - lx[i]) * zlx[i];
      if (ux[i] < INFINITY) mup += (ux[i] - xs[i]) * zux[i];
      double q = -bf[i];
#pragma unroll
      for (int j = 0; j < NX; ++j) q += Ak[i][j] * xs[j];
#pragma unroll
      for (int j = 0; j < NU; ++j) q += Bk[i][j] * us[j];
      q -= xnext[i];
      if (!inner) q = 0.0;
      rp[i] = q;
      rpm = fmax(rpm, fabs(q));
      nonf += isfinite(q) ? 0.0 : 1.0;
    }
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      double s = gu[i];
#pragma unroll
      for (int j = 0; j < NU; ++j) s += R[i][j] * us[j];
#pragma unroll
      for (int r = 0; r < NX; ++r) s += Bk[r][i] * y[r];
      if (!inner) s = 0.0;
      baseu[i] = -s;
      const double rd = s - zlu[i] + zuu[i];
      rdm = fmax(rdm, fabs(rd));
      nonf += isfinite(rd) ? 0.0 : 1.0;
      if (lu[i] > -INFINITY) mup += (us[i] - lu[i]) * zlu[i];
      if (uu[i] < INFINITY) mup += (uu[i] - us[i]) * zuu[i];
    }
    const double rdmax = ocp_max<G>(rdm), rpmax = ocp_max<G>(rpm), mu = ocp_sum<G>(mup) / nb;
    nonf = ocp_sum<G>(nonf);
    if (!done) {
      if (nonf > 0.0 || !isfinite(rdmax) || !isfinite(rpmax) || !isfinite(mu)) { st = HILO_STATUS_INFEASIBLE; done = true; }
      else {
        const double phi = fmax(fmax(rdmax / (1.0 + gmax), rpmax), mu);
        if (phi <= qd.tol) { st = HILO_STATUS_SOLVED; done = true; }
        else {
          phi_min = fmin(phi_min, phi);
          if (phi >= 1.0e4 * phi_min) { st = HILO_STATUS_INFEASIBLE; done = true; }   // OOQP's rule (hilo_qp.hip)
        }
      }
    }
    if (!done) it = iter + 1;
    QPO_STAMP(1);
    // ---- M = H + Sigma + reg, factorisation sweep: P_N = M_N; k = N-1 .. 0 ----
    double Mx[NX][NX], Mu[NU][NU], ilx[NX], iux[NX], ilu[NU], iuu[NU];   // reciprocal slacks (0 where there is no bound)
#pragma unroll
    for (int i = 0; i < NX; ++i) {
#pragma unroll
      for (int j = 0; j < NX; ++j) Mx[i][j] = Q[i][j];
      ilx[i] = lx[i] > -INFINITY ? ocp_rcp(xs[i] - lx[i]) : 0.0;
      iux[i] = ux[i] < INFINITY ? ocp_rcp(ux[i] - xs[i]) : 0.0;
      Mx[i][i] += qd.reg + zlx[i] * ilx[i] + zux[i] * iux[i];
    }
#pragma unroll
    for (int i = 0; i < NU; ++i) {
#pragma unroll
      for (int j = 0; j < NU; ++j) Mu[i][j] = R[i][j];
      ilu[i] = lu[i] > -INFINITY ? ocp_rcp(us[i] - lu[i]) : 0.0;
      iuu[i] = uu[i] < INFINITY ? ocp_rcp(uu[i] - us[i]) : 0.0;
      Mu[i][i] += qd.reg + zlu[i] * ilu[i] + zuu[i] * iuu[i];
    }
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
      for (int j = 0; j < NX; ++j) P[i][j] = Mx[i][j];      // stage N's is final, the others are overwritten on their turn
    double okf = 1.0;
    for (int j = 0; j < N; ++j) {
      const int kk = N - 1 - j;
      double Pq[NX][NX], PA[NX][NX], PB[NX][NU];
#pragma unroll
      for (int a = 0; a < NX; ++a)
#pragma unroll
        for (int c = 0; c < NX; ++c) Pq[a][c] = ocp_from_next<G>(P[a][c]);
#pragma unroll
      for (int a = 0; a < NX; ++a) {
#pragma unroll
        for (int c = 0; c < NX; ++c) {
          double s = 0.0;
#pragma unroll
          for (int e = 0; e < NX; ++e) s += Pq[a][e] * Ak[e][c];
          PA[a][c] = s;
        }
#pragma unroll
        for (int c = 0; c < NU; ++c) {
          double s = 0.0;
#pragma unroll
          for (int e = 0; e < NX; ++e) s += Pq[a][e] * Bk[e][c];
          PB[a][c] = s;
        }
      }
      double Quu[NU][NU], Qx[NU][NX], L[NU][NU], id[NU];
#pragma unroll
      for (int a = 0; a < NU; ++a) {
#pragma unroll
        for (int c = 0; c < NU; ++c) {
          double s = Mu[a][c];
#pragma unroll
          for (int e = 0; e < NX; ++e) s += Bk[e][a] * PB[e][c];
          Quu[a][c] = s;
        }
#pragma unroll
        for (int c = 0; c < NX; ++c) {
          double s = 0.0;
#pragma unroll
          for (int e = 0; e < NX; ++e) s += Bk[e][a] * PA[e][c];
          Qx[a][c] = s;
        }
      }
      // Quu = L L^T with the reciprocal diagonal; Quu^-1 = L^-T L^-1 column by column
      bool okk = true;
#pragma unroll
      for (int c = 0; c < NU; ++c) {
        double s = Quu[c][c];
#pragma unroll
        for (int e = 0; e < c; ++e) s -= L[c][e] * L[c][e];
        if (!(s > 0.0)) okk = false;
        id[c] = qp_rsq(s > 0.0 ? s : 1.0);
#pragma unroll
        for (int a = c + 1; a < NU; ++a) {
          double t = Quu[a][c];
#pragma unroll
          for (int e = 0; e < c; ++e) t -= L[a][e] * L[c][e];
          L[a][c] = t * id[c];
        }
      }
      double Qinv[NU][NU];
#pragma unroll
      for (int c = 0; c < NU; ++c) {
        double z[NU];
#pragma unroll
        for (int a = 0; a < NU; ++a) {
          double s = a == c ? 1.0 : 0.0;
#pragma unroll
          for (int e = 0; e < a; ++e) s -= L[a][e] * z[e];
          z[a] = s * id[a];
        }
#pragma unroll
        for (int a = NU - 1; a >= 0; --a) {
          double s = z[a];
#pragma unroll
          for (int e = a + 1; e < NU; ++e) s -= L[e][a] * Qinv[e][c];
          Qinv[a][c] = s * id[a];
        }
      }
      double Kq[NU][NX], Pk[NX][NX];
#pragma unroll
      for (int a = 0; a < NU; ++a)
#pragma unroll
        for (int c = 0; c < NX; ++c) {
          double s = 0.0;
#pragma unroll
          for (int e = 0; e < NU; ++e) s -= Qinv[a][e] * Qx[e][c];
          Kq[a][c] = s;
        }
#pragma unroll
      for (int a = 0; a < NX; ++a)
#pragma unroll
        for (int c = 0; c < NX; ++c) {
          double s = Mx[a][c];
#pragma unroll
          for (int e = 0; e < NX; ++e) s += Ak[e][a] * PA[e][c];
#pragma unroll
          for (int e = 0; e < NU; ++e) s += Qx[e][a] * Kq[e][c];
          Pk[a][c] = s;
        }
      const bool mine = k == kk;
      if (mine && !okk) okf = 0.0;
#pragma unroll
      for (int a = 0; a < NX; ++a)
#pragma unroll
        for (int c = 0; c < NX; ++c) { P[a][c] = mine ? Pk[a][c] : P[a][c]; Pn[a][c] = mine ? Pq[a][c] : Pn[a][c]; }
#pragma unroll
      for (int a = 0; a < NU; ++a) {
#pragma unroll
        for (int c = 0; c < NX; ++c) { K[a][c] = mine ? Kq[a][c] : K[a][c]; Qux[a][c] = mine ? Qx[a][c] : Qux[a][c]; }
#pragma unroll
        for (int c = 0; c < NU; ++c) Qi[a][c] = mine ? Qinv[a][c] : Qi[a][c];
      }
    }
    QPO_STAMP(2);
    okf = ocp_min<G>(okf);
    if (!done && okf == 0.0) { st = HILO_STATUS_OTHER; done = true; it = iter; }
    // closed-loop matrix A + B K of this stage: both linear sweeps below are recursions v_next = Acl^(T) v + const
    double Acl[NX][NX];
#pragma unroll
    for (int a = 0; a < NX; ++a)
#pragma unroll
      for (int c = 0; c < NX; ++c) {
        double q = Ak[a][c];
#pragma unroll
        for (int e = 0; e < NU; ++e) q += Bk[a][e] * K[e][c];
        Acl[a][c] = inner ? q : 0.0;
      }

    double sigma_mu = 0.0;
    for (int pass = 0; pass < 2; ++pass) {
      QPO_STAMP(3 + 4 * pass);
      double r1x[NX], r1u[NU];
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        double s = basex[i];
        if (pass == 1) s += (sigma_mu - dx[i] * dzlx[i]) * ilx[i] - (sigma_mu + dx[i] * dzux[i]) * iux[i];
        r1x[i] = s;
      }
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        double s = baseu[i];
        if (pass == 1) s += (sigma_mu - du[i] * dzlu[i]) * ilu[i] - (sigma_mu + du[i] * dzuu[i]) * iuu[i];
        r1u[i] = s;
      }
      double ex[NX], eu[NU];
      if constexpr (G == 16) {
        // The two sweeps as SCANS over the 16 lanes of the row (4 steps each instead of N dependent ones):
        //   p_k = Acl_k^T (P_{k+1} rp_k + p_{k+1}) - r1x_k - K_k^T r1u_k,  p_N = -r1x_N        (suffix scan, row_shl)
        //   dx_{k+1} = Acl_k dx_k + B_k kff_k + rp_k,  dx_0 = 0                                 (prefix scan, row_shr)
        double Mat[NX][NX], d[NX], w[NX];
#pragma unroll
        for (int a = 0; a < NX; ++a) {
          double q = 0.0;
#pragma unroll
          for (int e = 0; e < NX; ++e) q += Pn[a][e] * rp[e];
          w[a] = q;
        }
#pragma unroll
        for (int a = 0; a < NX; ++a) {
          double q = -r1x[a];
#pragma unroll
          for (int e = 0; e < NX; ++e) q += Acl[e][a] * w[e];
#pragma unroll
          for (int e = 0; e < NU; ++e) q -= K[e][a] * r1u[e];
          d[a] = inner ? q : (stage ? -r1x[a] : 0.0);
#pragma unroll
          for (int c = 0; c < NX; ++c) Mat[a][c] = inner ? Acl[c][a] : ((!stage && a == c) ? 1.0 : 0.0);
        }
        ocp_scan_step<NX, 0x101>(Mat, d);
        ocp_scan_step<NX, 0x102>(Mat, d);
        ocp_scan_step<NX, 0x104>(Mat, d);
        ocp_scan_step<NX, 0x108>(Mat, d);
#pragma unroll
        for (int a = 0; a < NX; ++a) pv[a] = d[a];
        double h[NX];
#pragma unroll
        for (int a = 0; a < NX; ++a) h[a] = w[a] + ocp_from_next<G>(pv[a]);
#pragma unroll
        for (int a = 0; a < NU; ++a) {
          double q = -r1u[a];
#pragma unroll
          for (int e = 0; e < NX; ++e) q += Bk[e][a] * h[e];
          eu[a] = q;                                   // (qu)
        }
#pragma unroll
        for (int a = 0; a < NU; ++a) {
          double q = 0.0;
#pragma unroll
          for (int e = 0; e < NU; ++e) q -= Qi[a][e] * eu[e];
          kff[a] = inner ? q : 0.0;
        }
#pragma unroll
        for (int a = 0; a < NX; ++a) {
          double q = rp[a];
#pragma unroll
          for (int e = 0; e < NU; ++e) q += Bk[a][e] * kff[e];
          d[a] = inner ? q : 0.0;
#pragma unroll
          for (int c = 0; c < NX; ++c) Mat[a][c] = inner ? Acl[a][c] : (a == c ? 1.0 : 0.0);
        }
        ocp_scan_step<NX, 0x111>(Mat, d);
        ocp_scan_step<NX, 0x112>(Mat, d);
        ocp_scan_step<NX, 0x114>(Mat, d);
        ocp_scan_step<NX, 0x118>(Mat, d);
#pragma unroll
        for (int a = 0; a < NX; ++a) {
          const double prev = ocp_from_prev<G>(d[a]);
          ex[a] = first ? 0.0 : prev;
        }
#pragma unroll
        for (int a = 0; a < NU; ++a) {
          double q = kff[a];
#pragma unroll
          for (int e = 0; e < NX; ++e) q += K[a][e] * ex[e];
          eu[a] = inner ? q : 0.0;
        }
      } else {
      // backward: p_N = -r1x_N;  h = P_{k+1} rp_k + p_{k+1};  kff = -Quu^-1 (B^T h - r1u);  p_k = A^T h - r1x + Qux^T kff
#pragma unroll
      for (int i = 0; i < NX; ++i) pv[i] = -r1x[i];
      for (int j = 0; j < N; ++j) {
        const int kk = N - 1 - j;
        double pq[NX], h[NX], qu[NU], kq[NU], pk[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) pq[i] = ocp_from_next<G>(pv[i]);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          double s = pq[i];
#pragma unroll
          for (int e = 0; e < NX; ++e) s += Pn[i][e] * rp[e];
          h[i] = s;
        }
#pragma unroll
        for (int a = 0; a < NU; ++a) {
          double s = -r1u[a];
#pragma unroll
          for (int e = 0; e < NX; ++e) s += Bk[e][a] * h[e];
          qu[a] = s;
        }
#pragma unroll
        for (int a = 0; a < NU; ++a) {
          double s = 0.0;
#pragma unroll
          for (int e = 0; e < NU; ++e) s -= Qi[a][e] * qu[e];
          kq[a] = s;
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          double s = -r1x[i];
#pragma unroll
          for (int e = 0; e < NX; ++e) s += Ak[e][i] * h[e];
#pragma unroll
          for (int e = 0; e < NU; ++e) s += Qux[e][i] * kq[e];
          pk[i] = s;
        }
        const bool mine = k == kk;
#pragma unroll
        for (int i = 0; i < NX; ++i) pv[i] = mine ? pk[i] : pv[i];
#pragma unroll
        for (int a = 0; a < NU; ++a) kff[a] = mine ? kq[a] : kff[a];
      }
      // forward: dx_0 = 0;  du_k = K dx_k + kff;  dx_{k+1} = A dx_k + B du_k + rp_k;  dy_k = P_{k+1} dx_{k+1} + p_{k+1}
#pragma unroll
      for (int i = 0; i < NX; ++i) ex[i] = 0.0;
#pragma unroll
      for (int a = 0; a < NU; ++a) eu[a] = 0.0;
      for (int j = 0; j < N; ++j) {
        double un[NU], xn[NX];
#pragma unroll
        for (int a = 0; a < NU; ++a) {
          double s = kff[a];
#pragma unroll
          for (int e = 0; e < NX; ++e) s += K[a][e] * ex[e];
          un[a] = s;
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          double s = rp[i];
#pragma unroll
          for (int e = 0; e < NX; ++e) s += Ak[i][e] * ex[e];
#pragma unroll
          for (int e = 0; e < NU; ++e) s += Bk[i][e] * un[e];
          xn[i] = s;
        }
        const bool mine = k == j, nxt = k == j + 1;
#pragma unroll
        for (int a = 0; a < NU; ++a) eu[a] = mine ? un[a] : eu[a];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          const double fromprev = ocp_from_prev<G>(xn[i]);
          ex[i] = nxt ? fromprev : ex[i];
        }
      }
      }
      QPO_STAMP(4 + 4 * pass);
      double wy[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        double s = pv[i];
#pragma unroll
        for (int e = 0; e < NX; ++e) s += P[i][e] * ex[e];
        wy[i] = s;
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const double nxt = ocp_from_next<G>(wy[i]);     // (taken by every lane: a lane switched off is no source for its neighbour)
        dy[i] = inner ? nxt : 0.0;
      }
      // bound-multiplier steps, step lengths (hilo_qp.hip)
      double ap = 1.0, ad = 1.0;
      const double tau = pass == 0 ? 1.0 : fmax(0.995, 1.0 - mu);
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const double d = (first || !stage) ? 0.0 : ex[i];
        double dl = 0.0, dub = 0.0;
        const double cl = pass == 1 ? dx[i] * dzlx[i] : 0.0, cu = pass == 1 ? -dx[i] * dzux[i] : 0.0;
        const double rd_ = ocp_rcp(d);
        if (lx[i] > -INFINITY) {
          dl = (sigma_mu - cl) * ilx[i] - zlx[i] - zlx[i] * ilx[i] * d;
          if (d < 0.0) ap = fmin(ap, -tau * (xs[i] - lx[i]) * rd_);
          if (dl < 0.0) ad = fmin(ad, -tau * zlx[i] * ocp_rcp(dl));
        }
        if (ux[i] < INFINITY) {
          dub = (sigma_mu - cu) * iux[i] - zux[i] + zux[i] * iux[i] * d;
          if (d > 0.0) ap = fmin(ap, tau * (ux[i] - xs[i]) * rd_);
          if (dub < 0.0) ad = fmin(ad, -tau * zux[i] * ocp_rcp(dub));
        }
        dx[i] = d; dzlx[i] = dl; dzux[i] = dub;
      }
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        const double d = inner ? eu[i] : 0.0;
        double dl = 0.0, dub = 0.0;
        const double cl = pass == 1 ? du[i] * dzlu[i] : 0.0, cu = pass == 1 ? -du[i] * dzuu[i] : 0.0;
        const double rd_ = ocp_rcp(d);
        if (lu[i] > -INFINITY) {
          dl = (sigma_mu - cl) * ilu[i] - zlu[i] - zlu[i] * ilu[i] * d;
          if (d < 0.0) ap = fmin(ap, -tau * (us[i] - lu[i]) * rd_);
          if (dl < 0.0) ad = fmin(ad, -tau * zlu[i] * ocp_rcp(dl));
        }
        if (uu[i] < INFINITY) {
          dub = (sigma_mu - cu) * iuu[i] - zuu[i] + zuu[i] * iuu[i] * d;
          if (d > 0.0) ap = fmin(ap, tau * (uu[i] - us[i]) * rd_);
          if (dub < 0.0) ad = fmin(ad, -tau * zuu[i] * ocp_rcp(dub));
        }
        du[i] = d; dzlu[i] = dl; dzuu[i] = dub;
      }
      QPO_STAMP(5 + 4 * pass);
      ap = ocp_min<G>(ap);
      ad = ocp_min<G>(ad);
      QPO_STAMP(6 + 4 * pass);
      if (pass == 0) {
        double mp = 0.0;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          if (lx[i] > -INFINITY) mp += (xs[i] - lx[i] + ap * dx[i]) * (zlx[i] + ad * dzlx[i]);
          if (ux[i] < INFINITY) mp += (ux[i] - xs[i] - ap * dx[i]) * (zux[i] + ad * dzux[i]);
        }
#pragma unroll
        for (int i = 0; i < NU; ++i) {
          if (lu[i] > -INFINITY) mp += (us[i] - lu[i] + ap * du[i]) * (zlu[i] + ad * dzlu[i]);
          if (uu[i] < INFINITY) mp += (uu[i] - us[i] - ap * du[i]) * (zuu[i] + ad * dzuu[i]);
        }
        const double mu_aff = ocp_sum<G>(mp) / nb;
        const double sg = mu > 0.0 ? (mu_aff / mu) : 0.0;
        sigma_mu = sg * sg * sg * mu;
      } else if (!done) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          double v = xs[i] + ap * dx[i];
          if (lx[i] > -INFINITY) v = fmax(v, lx[i] + 4.0e-16 * fmax(1.0, fabs(lx[i])));
          if (ux[i] < INFINITY) v = fmin(v, ux[i] - 4.0e-16 * fmax(1.0, fabs(ux[i])));
          xs[i] = v;
          zlx[i] += ad * dzlx[i];
          zux[i] += ad * dzux[i];
          y[i] += ad * dy[i];
        }
#pragma unroll
        for (int i = 0; i < NU; ++i) {
          double v = us[i] + ap * du[i];
          if (lu[i] > -INFINITY) v = fmax(v, lu[i] + 4.0e-16 * fmax(1.0, fabs(lu[i])));
          if (uu[i] < INFINITY) v = fmin(v, uu[i] - 4.0e-16 * fmax(1.0, fabs(uu[i])));
          us[i] = v;
          zlu[i] += ad * dzlu[i];
          zuu[i] += ad * dzuu[i];
        }
      }
    }
  }

  // ---- outputs (CasADi conic sign convention: H x + g + A^T lam_a + lam_x = 0) ----
  double fp = 0.0;
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const double xv = first ? x0v[i] : xs[i];
    xs[i] = xv;
  }
  double yprev[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) yprev[i] = ocp_from_prev<G>(y[i]);
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    double hx = 0.0, aty = -(first ? 0.0 : yprev[i]);
#pragma unroll
    for (int j = 0; j < NX; ++j) hx += Q[i][j] * xs[j];
#pragma unroll
    for (int r = 0; r < NX; ++r) aty += (first ? A0[r][i] : Ak[r][i]) * y[r];
    const double gi = g[xo + i];
    if (stage) fp += xs[i] * (0.5 * hx + gi);
    if (valid && stage) {
      x_out[b * n + xo + i] = xs[i];
      if (lam_x) lam_x[b * n + xo + i] = first ? -(hx + gi + aty) : zux[i] - zlx[i];
      if (lam_a && inner) lam_a[b * m + ro + i] = y[i];
    }
  }
#pragma unroll
  for (int i = 0; i < NU; ++i) {
    double hx = 0.0;
#pragma unroll
    for (int j = 0; j < NU; ++j) hx += R[i][j] * us[j];
    if (inner) fp += us[i] * (0.5 * hx + gu[i]);
    if (valid && inner) {
      x_out[b * n + uo + i] = us[i];
      if (lam_x) lam_x[b * n + uo + i] = zuu[i] - zlu[i];
    }
  }
  fp = ocp_sum<G>(fp);
  if (valid && k == 0) {
    f_out[b] = fp;
    status[b] = st;
    iters[b] = it;
  }
}

}  // namespace hilo
