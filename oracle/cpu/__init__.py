"""ctypes front of the C++/OpenMP CPU baseline (oracle/cpu/nmpc_cpu.cpp, mhe_cpu.cpp, pf_cpu.cpp, kf_cpu.cpp, qp_cpu.cpp, gp_cpu.cpp).

TEST INFRASTRUCTURE / BASELINE ONLY - used by tests/test_cpu_baseline.py and by bench.py's `cpu_baseline` leg, never by the
product package.  The library takes the product's own problem descriptor (`hilo_nmpc_desc`, include/hilo_hip.h); here the
descriptor is filled from the oracle's `NmpcProblem`, so the C++ solver and the numpy solver see the same numbers.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, 'libhilo_cpu.so')
_lib = None


def build(force=False):
    src = [os.path.join(HERE, f) for f in ('nmpc_cpu.cpp', 'mhe_cpu.cpp', 'pf_cpu.cpp', 'pfdae_cpu.cpp', 'kf_cpu.cpp', 'qp_cpu.cpp', 'gp_cpu.cpp', 'models_cpu.h', 'ipm_cpu.h', 'Makefile')] + \
        [os.path.join(HERE, '..', '..', 'include', 'hilo_hip.h')]
    if force or not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in src):
        subprocess.check_call(['make', '-s', '-C', HERE] + (['-B'] if force else []))
    return LIB


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB)
        i32, i64, vp = C.c_int32, C.c_int64, C.c_void_p
        _lib.hilo_cpu_last_error.restype = C.c_char_p
        _lib.hilo_cpu_nmpc_create.argtypes = [vp, C.POINTER(vp)]
        _lib.hilo_cpu_nmpc_destroy.argtypes = [vp]
        _lib.hilo_cpu_nmpc_destroy.restype = None
        _lib.hilo_cpu_nmpc_solve.argtypes = [vp, i64, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, C.c_int]
        _lib.hilo_cpu_plant_step.argtypes = [vp, i64, vp, vp, vp, i64, vp, C.c_int]
        _lib.hilo_cpu_mhe_last_error.restype = C.c_char_p
        _lib.hilo_cpu_mhe_create.argtypes = [vp, C.POINTER(vp)]
        _lib.hilo_cpu_mhe_destroy.argtypes = [vp]
        _lib.hilo_cpu_mhe_destroy.restype = None
        _lib.hilo_cpu_mhe_estimate.argtypes = [vp, i64, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int]
        _lib.hilo_cpu_pf_last_error.restype = C.c_char_p
        _lib.hilo_cpu_pf_create.argtypes = [vp, C.POINTER(vp)]
        _lib.hilo_cpu_pf_destroy.argtypes = [vp]
        _lib.hilo_cpu_pf_destroy.restype = None
        _lib.hilo_cpu_pf_solve.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int]
        _lib.hilo_cpu_pf_plant_step.argtypes = [vp, i64, vp, vp, vp, C.c_int]
        _lib.hilo_cpu_pfdae_last_error.restype = C.c_char_p
        _lib.hilo_cpu_pfdae_create.argtypes = [vp, C.POINTER(vp)]
        _lib.hilo_cpu_pfdae_destroy.argtypes = [vp]
        _lib.hilo_cpu_pfdae_destroy.restype = None
        _lib.hilo_cpu_pfdae_solve.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int]
        _lib.hilo_cpu_pfdae_plant_step.argtypes = [vp, i64, vp, vp, vp, C.c_int]
        _lib.hilo_cpu_set_gp.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, vp, vp, vp]
        _lib.hilo_cpu_kf_steps.argtypes = [C.c_int, C.c_int, C.c_double, i64, C.c_int, vp, vp, vp, vp, C.c_double, C.c_double, C.c_int]
        _lib.hilo_cpu_gp_predict.argtypes = [C.c_int, C.c_int, vp, vp, vp, C.c_double, vp, C.c_double, C.c_double, C.c_int, i64, vp, vp,
                                             vp, C.c_int]
        _lib.hilo_cpu_qp_solve.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, i64, vp, vp, C.c_double, C.c_int, C.c_double, vp, vp, vp,
                                           C.c_int]
    return _lib


def _check(rc):
    if rc != 0:
        raise RuntimeError(lib().hilo_cpu_last_error().decode())


class CpuNmpc:
    """Tracking NMPC of an oracle `NmpcProblem` on the C++ baseline."""

    def __init__(self, pb, **options):
        from hilo_mpc_amd._lib import NmpcDesc              # the ctypes mirror of hilo_nmpc_desc (tests/test_abi.py checks it)
        self.pb = pb
        d = NmpcDesc()
        d.model_id, d.N, d.erk_order, d.n_sub, d.dt = pb.model.model_id, pb.N, pb.order, pb.smap.n_sub, pb.dt
        d.bound_relax_factor = -1.
        for k, v in options.items():
            setattr(d, k, v)
        keep = []

        def hp(a):
            a = np.ascontiguousarray(a, dtype=np.float64)
            keep.append(a)
            return a.ctypes.data
        d.Wz, d.zref, d.WN, d.xrefN = hp(pb.Wz), hp(pb.zref), hp(pb.WN), hp(pb.xrefN)
        d.x_lb, d.x_ub, d.u_lb, d.u_ub = hp(pb.x_lb * pb.sx), hp(pb.x_ub * pb.sx), hp(pb.u_lb * pb.su), hp(pb.u_ub * pb.su)
        d.x_scaling, d.u_scaling = hp(pb.sx), hp(pb.su)
        d.x_guess, d.u_guess = hp(pb.x_guess * pb.sx), hp(pb.u_guess * pb.su)
        if np.abs(pb.Wdu).max() > 0 or pb.Nc != pb.N:
            raise NotImplementedError("the CPU baseline covers tracking NMPC with box bounds only")
        h = C.c_void_p()
        _check(lib().hilo_cpu_nmpc_create(C.byref(d), C.byref(h)))
        self._h = h
        self.n_v = (pb.N + 1) * pb.nx + pb.N * pb.nu

    def __del__(self):
        if getattr(self, '_h', None) is not None and _lib is not None:
            _lib.hilo_cpu_nmpc_destroy(self._h)
            self._h = None

    def solve(self, x0, p, v0=None, n_threads=0):
        pb = self.pb
        x0 = np.ascontiguousarray(np.atleast_2d(x0), dtype=np.float64)
        B = x0.shape[0]
        par = np.ascontiguousarray(np.broadcast_to(np.atleast_2d(np.asarray(p, dtype=np.float64)), (B, pb.np_))) if pb.np_ else None
        v0 = None if v0 is None else np.ascontiguousarray(np.broadcast_to(v0, (B, self.n_v)), dtype=np.float64)
        v, f, u0 = np.empty((B, self.n_v)), np.empty(B), np.empty((B, pb.nu))
        st, it, kkt = np.empty(B, np.int32), np.empty(B, np.int32), np.empty(B)
        _check(lib().hilo_cpu_nmpc_solve(self._h, B, x0.ctypes.data, par.ctypes.data if par is not None else None, pb.np_,
                                         v0.ctypes.data if v0 is not None else None, v.ctypes.data, f.ctypes.data, u0.ctypes.data,
                                         st.ctypes.data, it.ctypes.data, kkt.ctypes.data, int(n_threads)))
        return dict(v=v, f=f, u0=u0, status=st, iters=it, kkt=kkt)

    def plant_step(self, x, u, p, n_threads=0):
        pb = self.pb
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
        B = x.shape[0]
        u = np.ascontiguousarray(np.broadcast_to(np.atleast_2d(u), (B, pb.nu)), dtype=np.float64)
        par = np.ascontiguousarray(np.broadcast_to(np.atleast_2d(np.asarray(p, dtype=np.float64)), (B, pb.np_))) if pb.np_ else None
        xn = np.empty_like(x)
        _check(lib().hilo_cpu_plant_step(self._h, B, x.ctypes.data, u.ctypes.data, par.ctypes.data if par is not None else None,
                                         pb.np_, xn.ctypes.data, int(n_threads)))
        return xn


class CpuMhe:
    """Moving-horizon estimator of an oracle `MheProblem` (state noise, pinned parameters) on the C++ baseline."""

    def __init__(self, pb, **options):
        from hilo_mpc_amd._lib import MheDesc               # the ctypes mirror of hilo_mhe_desc (tests/test_abi.py checks it)
        self.pb = pb
        d = MheDesc()
        d.model_id, d.N, d.erk_order, d.n_sub, d.dt = pb.model.model_id, pb.N, pb.smap.order, pb.smap.n_sub, pb.dt
        d.bound_relax_factor = -1.
        for k, v in options.items():
            setattr(d, k, v)
        keep = []

        def hp(a):
            a = np.ascontiguousarray(a, dtype=np.float64)
            keep.append(a)
            return a.ctypes.data
        d.Wx, d.Wy, d.Ww = hp(pb.Wx), hp(pb.Wy), hp(pb.Ww)
        d.x_lb, d.x_ub, d.w_lb, d.w_ub = hp(pb.x_lb * pb.sx), hp(pb.x_ub * pb.sx), hp(pb.w_lb * pb.sw), hp(pb.w_ub * pb.sw)
        d.x_scaling, d.w_scaling, d.u_scaling = hp(pb.sx), hp(pb.sw), hp(pb.su)
        d.x_guess, d.w_guess = hp(pb.x_guess * pb.sx), hp(pb.w_guess * pb.sw)
        h = C.c_void_p()
        if lib().hilo_cpu_mhe_create(C.byref(d), C.byref(h)) != 0:
            raise RuntimeError(lib().hilo_cpu_mhe_last_error().decode())
        self._h = h
        self.n_v = pb.n_v

    def __del__(self):
        if getattr(self, '_h', None) is not None and _lib is not None:
            _lib.hilo_cpu_mhe_destroy(self._h)
            self._h = None

    def solve(self, x_arrival, p, u_meas, y_meas, v0=None, n_threads=0):
        """Arguments like `MheIpm.solve`; v0 = warm start in the reference layout [p | x | w] (scaled)."""
        pb = self.pb
        xa = np.ascontiguousarray(np.atleast_2d(x_arrival), dtype=np.float64)
        B = xa.shape[0]
        par = np.ascontiguousarray(np.broadcast_to(np.atleast_2d(np.asarray(p, dtype=np.float64)), (B, pb.np_)))
        um = np.ascontiguousarray(np.asarray(u_meas, dtype=np.float64).reshape(B, pb.N, pb.nu))
        ym = np.ascontiguousarray(np.asarray(y_meas, dtype=np.float64).reshape(B, pb.N, pb.ny))
        v0 = None if v0 is None else np.ascontiguousarray(np.broadcast_to(v0, (B, self.n_v)), dtype=np.float64)
        v, f, xo = np.empty((B, self.n_v)), np.empty(B), np.empty((B, pb.nx))
        st, it, kkt = np.empty(B, np.int32), np.empty(B, np.int32), np.empty(B)
        rc = lib().hilo_cpu_mhe_estimate(self._h, B, xa.ctypes.data, par.ctypes.data, pb.np_, um.ctypes.data, ym.ctypes.data,
                                         v0.ctypes.data if v0 is not None else None, v.ctypes.data, f.ctypes.data, xo.ctypes.data,
                                         st.ctypes.data, it.ctypes.data, kkt.ctypes.data, int(n_threads))
        if rc != 0:
            raise RuntimeError(lib().hilo_cpu_mhe_last_error().decode())
        return dict(v=v, f=f, x_opt=xo, status=st, iters=it, kkt=kkt)


class PfDesc(C.Structure):
    """hilo_cpu_pf_desc (oracle/cpu/pf_cpu.cpp)."""
    _fields_ = [(n, C.c_int32) for n in ('N', 'erk_order', 'n_sub', 'max_iter', 'acceptable_iter')] + \
               [(n, C.c_double) for n in ('dt', 'tol', 'acceptable_tol', 'mu_init', 'bound_relax_factor')] + \
               [(n, C.c_void_p) for n in ('Wz', 'zref', 'WN', 'xrefN', 'x_lb', 'x_ub', 'u_lb', 'u_ub', 'x_guess', 'u_guess')] + \
               [('w_path_stage', C.c_double * 2), ('w_path_term', C.c_double * 2)] + \
               [(n, C.c_double) for n in ('theta_lb', 'theta_ub', 'theta_guess', 'u_pf_lb', 'u_pf_ub', 'con_ub', 'con_weight',
                                          'max_violation')]


class CpuPathNmpc:
    """The path-following NMPC of configuration C5 (a `GenNmpcProblem` of that shape: robot6, path references (sin theta,
    sin 2 theta) for (px, py), soft constraint vx^2 + vy^2 <= ub) on the C++ baseline; anything else is refused."""

    def __init__(self, spec, pb, **options):
        path, con = spec.get('path') or {}, spec.get('constraint') or {}
        ok = spec['model'] == 'robot6' and not spec.get('terminal_constraint') and not spec.get('generic_stage') and \
            [tuple(t[0]) + tuple(t[2]) for t in path.get('stage', [])] == [(0, 2, 'sin(theta)', 'sin(2*theta)')] and \
            [tuple(t[0]) + tuple(t[2]) for t in path.get('terminal', [])] == [(0, 2, 'sin(theta)', 'sin(2*theta)')] and \
            path.get('u_pf_ref') is None and list(con.get('expr', [])) == ['vx**2 + vy**2'] and con.get('soft') and \
            not np.isfinite(np.asarray(con.get('lb', -np.inf), dtype=float)).any() and \
            np.all(pb.sx == 1.) and np.all(pb.su == 1.) and pb.Nc == pb.N and np.abs(pb.Wdu).max() == 0
        if not ok:
            raise NotImplementedError("the CPU baseline of the general NMPC holds configuration C5's problem functions only")
        self.pb = pb
        d = PfDesc()
        d.N, d.erk_order, d.n_sub, d.dt = pb.N, pb.order, pb.smap.n_sub, pb.dt
        d.bound_relax_factor = -1.
        for k, v in options.items():
            setattr(d, k, v)
        keep = []

        def hp(a):
            a = np.ascontiguousarray(a, dtype=np.float64)
            keep.append(a)
            return a.ctypes.data
        nx, nu = pb.nx, pb.nu
        d.Wz, d.zref, d.WN, d.xrefN = hp(pb.Wz), hp(pb.zref), hp(pb.WN), hp(pb.xrefN)
        d.x_lb, d.x_ub, d.u_lb, d.u_ub = hp(pb.x_lb[:nx]), hp(pb.x_ub[:nx]), hp(pb.u_lb[:nu]), hp(pb.u_ub[:nu])
        d.x_guess, d.u_guess = hp(pb.x_guess[:nx]), hp(pb.u_guess[:nu])
        d.w_path_stage[:] = [float(w) for w in path['stage'][0][1]]
        d.w_path_term[:] = [float(w) for w in path['terminal'][0][1]]
        d.theta_lb, d.theta_ub, d.theta_guess = pb.x_lb[nx], pb.x_ub[nx], pb.x_guess[nx]
        d.u_pf_lb, d.u_pf_ub = pb.u_lb[nu], pb.u_ub[nu]
        d.con_ub, d.con_weight, d.max_violation = float(pb.dub[0]), float(pb.We[0, 0]), float(pb.e_ub[0])
        h = C.c_void_p()
        if lib().hilo_cpu_pf_create(C.byref(d), C.byref(h)) != 0:
            raise RuntimeError(lib().hilo_cpu_pf_last_error().decode())
        self._h = h
        self.n_w = (pb.N + 1) * (nx + 2) + pb.N * (nu + 1)
        self.n_v = pb.n_v

    def __del__(self):
        if getattr(self, '_h', None) is not None and _lib is not None:
            _lib.hilo_cpu_pf_destroy(self._h)
            self._h = None

    def solve(self, x0, w0=None, n_threads=0):
        """w0: warm start in the solver's own layout (the `w` of a previous call).  `v` is the reference's decision vector."""
        pb = self.pb
        x0 = np.ascontiguousarray(np.atleast_2d(x0), dtype=np.float64)
        B = x0.shape[0]
        w0 = None if w0 is None else np.ascontiguousarray(np.broadcast_to(w0, (B, self.n_w)), dtype=np.float64)
        w, v, f, u0 = np.empty((B, self.n_w)), np.empty((B, self.n_v)), np.empty(B), np.empty((B, pb.nu))
        st, it, kkt = np.empty(B, np.int32), np.empty(B, np.int32), np.empty(B)
        rc = lib().hilo_cpu_pf_solve(self._h, B, x0.ctypes.data, w0.ctypes.data if w0 is not None else None, w.ctypes.data, v.ctypes.data,
                                     f.ctypes.data, u0.ctypes.data, st.ctypes.data, it.ctypes.data, kkt.ctypes.data, int(n_threads))
        if rc != 0:
            raise RuntimeError(lib().hilo_cpu_pf_last_error().decode())
        return dict(w=w, v=v, f=f, u0=u0, status=st, iters=it, kkt=kkt)

    def plant_step(self, x, u, n_threads=0):
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
        u = np.ascontiguousarray(np.broadcast_to(np.atleast_2d(u), (x.shape[0], self.pb.nu)), dtype=np.float64)
        xn = np.empty_like(x)
        lib().hilo_cpu_pf_plant_step(self._h, x.shape[0], x.ctypes.data, u.ctypes.data, xn.ctypes.data, int(n_threads))
        return xn


class PfDaeDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('N', 'degree', 'max_iter', 'acceptable_iter')] + \
               [(n, C.c_double) for n in ('dt', 'tol', 'acceptable_tol', 'mu_init', 'bound_relax_factor')] + \
               [(n, C.c_void_p) for n in ('coll_A', 'coll_D', 'coll_B', 'Wu', 'uref', 'x_lb', 'x_ub', 'u_lb', 'u_ub', 'x_guess', 'u_guess')] + \
               [('w_path_stage', C.c_double * 2), ('w_path_term', C.c_double * 2)] + \
               [(n, C.c_double) for n in ('theta_lb', 'theta_ub', 'theta_guess', 'u_pf_lb', 'u_pf_ub', 'con_ub', 'con_weight',
                                          'max_violation')]


class CpuPathDaeNmpc:
    """BASELINE configuration 5 as it is written (tests/problems.py::C5D - a `GenCollProblem` of that shape: the robot's DAE, path
    references (sin theta, sin 2 theta) for (px, py), soft limit z <= ub on the algebraic state, collocation with the continuous
    objective) on the C++ baseline (pfdae_cpu.cpp); anything else is refused."""

    def __init__(self, spec, pb, **options):
        path, con = spec.get('path') or {}, spec.get('constraint') or {}
        nx, nu = pb.nx, pb.nu
        Wx = pb.Wz.copy()
        Wx[nx:, nx:] -= np.diag(np.diag(pb.Wz[nx:, nx:]))
        ok = spec['model'] == 'robot6_dae' and not spec.get('generic_stage') and pb.objective == 'continuous' and \
            [tuple(t[0]) + tuple(t[2]) for t in path.get('stage', [])] == [(0, 2, 'sin(theta)', 'sin(2*theta)')] and \
            [tuple(t[0]) + tuple(t[2]) for t in path.get('terminal', [])] == [(0, 2, 'sin(theta)', 'sin(2*theta)')] and \
            path.get('u_pf_ref') is None and list(con.get('expr', [])) == ['z'] and con.get('soft') and \
            not np.isfinite(np.asarray(con.get('lb', -np.inf), dtype=float)).any() and \
            np.all(pb.sx == 1.) and np.all(pb.su == 1.) and pb.Nc == pb.N and np.abs(pb.Wdu).max() == 0 and \
            np.abs(Wx).max() == 0 and np.abs(pb.WN).max() == 0 and np.all(np.isinf(pb.x_lb[:nx])) and np.all(np.isinf(pb.x_ub[:nx]))
        if not ok:
            raise NotImplementedError("the CPU baseline of the general collocation NMPC holds configuration C5-DAE's problem functions only")
        self.pb = pb
        d = PfDaeDesc()
        d.N, d.degree, d.dt = pb.N, pb.d, pb.dt
        d.bound_relax_factor = -1.
        for k, v in options.items():
            setattr(d, k, v)
        keep = []

        def hp(a):
            a = np.ascontiguousarray(a, dtype=np.float64)
            keep.append(a)
            return a.ctypes.data
        d.coll_A, d.coll_D, d.coll_B = hp(np.linalg.inv(pb.C[1:, 1:].T)), hp(pb.D), hp(pb.B)
        d.Wu, d.uref = hp(np.diag(pb.Wz)[nx:]), hp(pb.zref[nx:])
        d.x_lb, d.x_ub, d.u_lb, d.u_ub = hp(pb.x_lb[:nx]), hp(pb.x_ub[:nx]), hp(pb.u_lb[:nu]), hp(pb.u_ub[:nu])
        d.x_guess, d.u_guess = hp(pb.x_guess[:nx]), hp(pb.u_guess[:nu])
        d.w_path_stage[:] = [float(w) for w in path['stage'][0][1]]
        d.w_path_term[:] = [float(w) for w in path['terminal'][0][1]]
        d.theta_lb, d.theta_ub, d.theta_guess = pb.x_lb[nx], pb.x_ub[nx], pb.x_guess[nx]
        d.u_pf_lb, d.u_pf_ub = pb.u_lb[nu], pb.u_ub[nu]
        d.con_ub, d.con_weight, d.max_violation = float(pb.rows[0][4]), float(pb.We[0, 0]), float(pb.e_ub[0])
        h = C.c_void_p()
        if lib().hilo_cpu_pfdae_create(C.byref(d), C.byref(h)) != 0:
            raise RuntimeError(lib().hilo_cpu_pfdae_last_error().decode())
        self._h = h
        self.n_w = (pb.N + 1) * (nx + 2) + pb.N * (nu + 1)
        self.n_vx = (pb.N + 1) * (nx + 1) + pb.N * (nu + 1) + 1

    def __del__(self):
        if getattr(self, '_h', None) is not None and _lib is not None:
            _lib.hilo_cpu_pfdae_destroy(self._h)
            self._h = None

    def solve(self, x0, w0=None, n_threads=0):
        """w0: warm start in the solver's own layout (the `w` of a previous call).  `vx` = [x with theta | u with u_theta | e]."""
        pb = self.pb
        x0 = np.ascontiguousarray(np.atleast_2d(x0), dtype=np.float64)
        B = x0.shape[0]
        w0 = None if w0 is None else np.ascontiguousarray(np.broadcast_to(w0, (B, self.n_w)), dtype=np.float64)
        w, v, f, u0 = np.empty((B, self.n_w)), np.empty((B, self.n_vx)), np.empty(B), np.empty((B, pb.nu))
        st, it, kkt = np.empty(B, np.int32), np.empty(B, np.int32), np.empty(B)
        rc = lib().hilo_cpu_pfdae_solve(self._h, B, x0.ctypes.data, w0.ctypes.data if w0 is not None else None, w.ctypes.data,
                                        v.ctypes.data, f.ctypes.data, u0.ctypes.data, st.ctypes.data, it.ctypes.data, kkt.ctypes.data,
                                        int(n_threads))
        if rc != 0:
            raise RuntimeError(lib().hilo_cpu_pfdae_last_error().decode())
        return dict(w=w, vx=v, f=f, u0=u0, status=st, iters=it, kkt=kkt)

    def plant_step(self, x, u, n_threads=0):
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
        u = np.ascontiguousarray(np.broadcast_to(np.atleast_2d(u), (x.shape[0], self.pb.nu)), dtype=np.float64)
        xn = np.empty_like(x)
        lib().hilo_cpu_pfdae_plant_step(self._h, x.shape[0], x.ctypes.data, u.ctypes.data, xn.ctypes.data, int(n_threads))
        return xn


def max_threads():
    return int(lib().hilo_cpu_max_threads())


_gp_keep = []


def set_gp(X_train, alpha, length_scales, signal_variance=1., bias=0.):
    """The learned term of the chemostat4_gp model (oracle/models.py::chemostat4_gp): squared-exponential posterior mean over (S, I)."""
    Xt = np.ascontiguousarray(np.atleast_2d(X_train), dtype=np.float64)
    al = np.ascontiguousarray(np.ravel(alpha), dtype=np.float64)
    ls = np.broadcast_to(np.asarray(length_scales, dtype=np.float64), (2,))
    M = np.exp(-2 * np.log(ls))
    x0, x1 = np.ascontiguousarray(Xt[0]), np.ascontiguousarray(Xt[1])
    _gp_keep[:] = [x0, x1, al]
    lib().hilo_cpu_set_gp(al.size, float(signal_variance), float(bias), float(M[0]), float(M[1]), x0.ctypes.data, x1.ctypes.data,
                          al.ctypes.data)


def kf_steps(kind, xP, y, u, p, q, r, dt=1., order=4, n_threads=0):
    """`steps` fused filter steps (predict + update) of the rk4-discretised chemostat for a batch: xP [B, 4, 5] packed [x | P],
    y [steps, B, 2]; returns the new tile."""
    xP = np.array(xP, dtype=np.float64, order='C')
    y = np.ascontiguousarray(y, dtype=np.float64)
    if y.ndim == 2:
        y = y[None]
    B = xP.shape[0]
    u = np.ascontiguousarray(np.broadcast_to(u, (B, 2)), dtype=np.float64)
    p = np.ascontiguousarray(np.broadcast_to(p, (B, 4)), dtype=np.float64)
    rc = lib().hilo_cpu_kf_steps(int(kind == 'ukf'), int(order), float(dt), B, y.shape[0], xP.ctypes.data, y.ctypes.data, u.ctypes.data,
                                 p.ctypes.data, float(q), float(r), int(n_threads))
    if rc != 0:
        raise RuntimeError('hilo_cpu_kf_steps: bad argument')
    return xP


def qp_solve(H, g, A, b, lbx, ubx, tol=1e-10, max_iter=100, reg=1e-11, n_threads=0):
    """The batch of QPs of `LMPC.optimize` (shared H, g, A, b; per-instance bounds): dict(x, status, iters)."""
    H, g, A, b = (np.ascontiguousarray(a, dtype=np.float64) for a in (H, g, A, b))
    lbx, ubx = np.ascontiguousarray(np.atleast_2d(lbx), dtype=np.float64), np.ascontiguousarray(np.atleast_2d(ubx), dtype=np.float64)
    B, n = lbx.shape
    x, st, it = np.empty((B, n)), np.empty(B, np.int32), np.empty(B, np.int32)
    rc = lib().hilo_cpu_qp_solve(n, A.shape[0], H.ctypes.data, g.ctypes.data, A.ctypes.data, b.ctypes.data, B, lbx.ctypes.data,
                                 ubx.ctypes.data, float(tol), int(max_iter), float(reg), x.ctypes.data, st.ctypes.data, it.ctypes.data,
                                 int(n_threads))
    if rc != 0:
        raise RuntimeError('hilo_cpu_qp_solve: bad argument')
    return dict(x=x, status=st, iters=it)


def gp_predict(post, Xq, noise_free=False, n_threads=0):
    """`Posterior.predict` of an oracle posterior with a squared-exponential kernel (one length scale per feature or one for all) and
    a zero / constant mean: (mean [m], var [m])."""
    ks, ms = post.kernel_spec, post.mean_spec
    if ks['type'] != 'squared_exponential' or ms['type'] not in ('zero', 'constant'):
        raise NotImplementedError("the CPU baseline of the prediction covers the squared-exponential kernel with a zero / constant mean")
    kw = ks['kwargs']
    dims = list(kw.get('active_dims') or range(post.X.shape[0]))
    X = np.ascontiguousarray(post.X[dims], dtype=np.float64)
    Xq = np.ascontiguousarray(np.atleast_2d(np.asarray(Xq, dtype=np.float64))[dims])
    nf, n = X.shape
    Minv = np.ascontiguousarray(np.broadcast_to(1. / np.asarray(kw.get('length_scales', 1.), dtype=np.float64) ** 2, (nf,)))
    al, R = np.ascontiguousarray(post.alpha, dtype=np.float64), np.ascontiguousarray(post.R, dtype=np.float64)
    bias = float((ms.get('kwargs') or {}).get('bias', 1.)) if ms['type'] == 'constant' else 0.
    m = Xq.shape[1]
    mu, var = np.empty(m), np.empty(m)
    rc = lib().hilo_cpu_gp_predict(n, nf, X.ctypes.data, al.ctypes.data, R.ctypes.data, float(kw.get('signal_variance', 1.)),
                                   Minv.ctypes.data, bias, float(post.sn2), int(noise_free), m, Xq.ctypes.data, mu.ctypes.data,
                                   var.ctypes.data, int(n_threads))
    if rc != 0:
        raise RuntimeError('hilo_cpu_gp_predict: bad argument')
    return mu, var
