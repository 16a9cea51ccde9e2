"""ctypes binding of libhilo_hip.so (include/hilo_hip.h).

There is NO CPU fallback: if the HIP library is missing or fails to load, importing any compute class raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('HILO_LIB_PATH') or os.path.join(HERE, 'libhilo_hip.so')   # HILO_LIB_PATH: a developer variant (_build.build(tag=...))

c_double_p = C.c_void_p   # device pointers travel as integers
c_i64 = C.c_int64


class HiloError(RuntimeError):
    code = 0


class NotPositiveDefinite(HiloError, ValueError):
    """HILO_ENOTPD: K + sn2 I has a non-positive pivot (the reference adds no jitter, inference.py:206)."""
    code = -5


class KfDesc(C.Structure):
    _fields_ = [('model_id', C.c_int32), ('kind', C.c_int32), ('continuous', C.c_int32), ('erk_order', C.c_int32),
                ('n_sub', C.c_int32), ('lti_nx', C.c_int32), ('lti_nu', C.c_int32), ('lti_ny', C.c_int32),
                ('dt', C.c_double), ('alpha', C.c_double), ('beta', C.c_double), ('kappa', C.c_double),
                ('user_source', C.c_char_p), ('n_user_gp', C.c_int32), ('user_gp', C.c_void_p * 4)]


class NmpcDesc(C.Structure):
    _fields_ = [('model_id', C.c_int32), ('N', C.c_int32), ('Nc', C.c_int32), ('erk_order', C.c_int32),
                ('n_sub', C.c_int32), ('max_iter', C.c_int32), ('acceptable_iter', C.c_int32), ('reserved', C.c_int32),
                ('dt', C.c_double), ('tol', C.c_double), ('acceptable_tol', C.c_double), ('mu_init', C.c_double),
                ('bound_relax_factor', C.c_double)] + \
               [(n, C.c_void_p) for n in ('Wz', 'zref', 'WN', 'xrefN', 'Wdu', 'x_lb', 'x_ub', 'u_lb', 'u_ub',
                                          'x_scaling', 'u_scaling', 'x_guess', 'u_guess', 'learned')] + \
               [('n_path_var', C.c_int32), ('has_u_pf_ref', C.c_int32)] + \
               [(n, C.c_double) for n in ('theta_guess', 'theta_lb', 'theta_ub', 'u_pf_lb', 'u_pf_ub', 'u_pf_ref',
                                          'u_pf_weight')] + \
               [('n_path_stage', C.c_int32), ('n_path_term', C.c_int32)] + \
               [(n, C.c_void_p) for n in ('path_stage_idx', 'path_stage_W', 'path_term_idx', 'path_term_W', 'path_prog')] + \
               [(n, C.c_int32) for n in ('path_prog_len', 'n_con', 'con_soft', 'con_prog_len')] + \
               [(n, C.c_void_p) for n in ('con_prog', 'con_lb', 'con_ub', 'con_weight', 'con_max_violation')] + \
               [('n_tcon', C.c_int32), ('tcon_prog_len', C.c_int32), ('tcon_prog', C.c_void_p), ('tcon_lb', C.c_void_p),
                ('tcon_ub', C.c_void_p), ('tcon_soft', C.c_int32), ('reserved4', C.c_int32), ('tcon_weight', C.c_void_p),
                ('tcon_max_violation', C.c_void_p)] + \
               [('collocation_degree', C.c_int32), ('reserved2', C.c_int32), ('coll_A', C.c_void_p), ('coll_D', C.c_void_p),
                ('time_varying', C.c_int32), ('reserved3', C.c_int32)] + \
               [('user_source', C.c_char_p)] + \
               [(n, C.c_int32) for n in ('user_nx', 'user_nu', 'user_np', 'user_ny', 'user_discrete', 'user_has_fun', 'user_policy',
                                         'objective_continuous')] + \
               [('coll_B', C.c_void_p), ('n_user_gp', C.c_int32), ('user_nz', C.c_int32), ('user_gp', C.c_void_p * 4),
                ('hess_pattern', C.c_void_p), ('max_hessian_perturbation', C.c_double), ('n_zbound', C.c_int32), ('x0_free_mask', C.c_int32),
                ('zb_lb', C.c_void_p), ('zb_ub', C.c_void_p), ('n_acc', C.c_int32), ('n_acc_expr', C.c_int32),
                ('acc_coef', C.c_void_p), ('acc_lb', C.c_void_p), ('acc_ub', C.c_void_p), ('acc_soft', C.c_int32),
                ('acc_max_violation', C.c_void_p)]


class MheDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('model_id', 'N', 'erk_order', 'n_sub', 'max_iter', 'acceptable_iter')] + \
               [(n, C.c_double) for n in ('dt', 'tol', 'acceptable_tol', 'mu_init', 'bound_relax_factor')] + \
               [(n, C.c_void_p) for n in ('Wx', 'Wy', 'Ww', 'x_lb', 'x_ub', 'w_lb', 'w_ub', 'x_scaling', 'w_scaling',
                                          'u_scaling', 'x_guess', 'w_guess')] + \
               [('estimate_parameters', C.c_int32), ('reserved', C.c_int32)] + \
               [(n, C.c_void_p) for n in ('Wp', 'p_lb', 'p_ub', 'p_scaling', 'p_guess')] + \
               [('user_source', C.c_char_p)] + \
               [(n, C.c_int32) for n in ('user_nx', 'user_nu', 'user_np', 'user_ny', 'user_discrete', 'collocation_degree')] + \
               [('coll_A', C.c_void_p), ('coll_D', C.c_void_p), ('n_con', C.c_int32), ('reserved6', C.c_int32), ('con_lb', C.c_void_p),
                ('con_ub', C.c_void_p)]


_lib = None


def _declare(lib):
    vp, i32, i64, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_double
    P = C.POINTER
    sig = {
        'hilo_abi_version': (C.c_int, []),
        'hilo_last_error': (C.c_char_p, []),
        'hilo_device_count': (C.c_int, [P(C.c_int)]),
        'hilo_model_dims': (C.c_int, [i32, P(C.c_int), P(C.c_int), P(C.c_int), P(C.c_int), P(C.c_int)]),
        'hilo_malloc': (C.c_int, [P(vp), C.c_uint64, i32]),
        'hilo_free': (C.c_int, [vp]),
        'hilo_memcpy_h2d': (C.c_int, [vp, vp, C.c_uint64, vp]),
        'hilo_memcpy_d2h': (C.c_int, [vp, vp, C.c_uint64, vp]),
        'hilo_stream_sync': (C.c_int, [vp]),
        'hilo_kf_create': (C.c_int, [P(KfDesc), i32, P(vp)]),
        'hilo_kf_destroy': (None, [vp]),
        'hilo_kf_dims': (C.c_int, [vp, P(C.c_int), P(C.c_int), P(C.c_int), P(C.c_int), P(C.c_int)]),
        'hilo_kf_predict': (C.c_int, [vp, i64, vp, vp, i64, vp, i64, vp, vp]),
        'hilo_kf_update': (C.c_int, [vp, i64, vp, vp, vp, i64, vp, i64, vp, vp, vp]),
        'hilo_kf_step': (C.c_int, [vp, i64, vp, vp, vp, i64, vp, i64, vp, i64, vp, vp, vp]),
        'hilo_pf_function': (C.c_int, [vp, i64, i32, vp, vp, vp, i64, vp, vp, vp, i64, vp, vp, vp, vp]),
        'hilo_pf_stats': (C.c_int, [vp, i64, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        'hilo_pf_resample': (C.c_int, [vp, i64, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
        'hilo_nmpc_create': (C.c_int, [P(NmpcDesc), i32, P(vp)]),
        'hilo_nmpc_destroy': (None, [vp]),
        'hilo_nmpc_dims': (C.c_int, [vp, P(C.c_int), P(C.c_int), P(C.c_int), P(C.c_int), P(C.c_int)]),
        'hilo_nmpc_reset_warm_start': (C.c_int, [vp]),
        'hilo_nmpc_set_fix_x0': (C.c_int, [vp, i32]),
        'hilo_nmpc_set_x0_box': (C.c_int, [vp, vp, vp]),
        'hilo_nmpc_set_gather': (C.c_int, [vp, vp, i32]),
        'hilo_nmpc_set_plant_out': (C.c_int, [vp, vp]),
        'hilo_nmpc_set_var_bounds': (C.c_int, [vp, vp, vp]),
        'hilo_kf_steps': (C.c_int, [vp, i64, i32, vp, vp, vp, i64, i64, vp, i64, vp, i64, vp, i32, vp, vp]),
        'hilo_kf_steps_split': (C.c_int, [vp, i64, i32, vp, vp, vp, i64, i64, vp, i64, vp, i64, vp, i64, vp, i32, vp, vp]),
        'hilo_nmpc_solve': (C.c_int, [vp, i64, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        'hilo_nmpc_solve_tv': (C.c_int, [vp, i64, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        'hilo_nmpc_profile': (C.c_int, [vp, i32, vp]),
        'hilo_nmpc_plant_step': (C.c_int, [vp, i64, vp, vp, vp, i64, vp, vp]),
        'hilo_jit_precompile': (C.c_int, [C.c_char_p] + [i32] * 11),
        'hilo_nmpc_set_aux_outputs': (C.c_int, [vp, vp, vp]),
        'hilo_gp_set_mean_program': (C.c_int, [vp, vp, i32]),
        'hilo_jit_precompile_kf': (C.c_int, [C.c_char_p]),
        'hilo_mhe_create': (C.c_int, [P(MheDesc), i32, P(vp)]),
        'hilo_mhe_destroy': (None, [vp]),
        'hilo_mhe_dims': (C.c_int, [vp] + [P(C.c_int)] * 6),
        'hilo_mhe_reset_warm_start': (C.c_int, [vp]),
        'hilo_mhe_estimate': (C.c_int, [vp, i64, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        'hilo_qp_create': (C.c_int, [i32, i32, i32, P(vp)]),
        'hilo_qp_destroy': (None, [vp]),
        'hilo_qp_set_options': (C.c_int, [vp, dbl, i32]),
        'hilo_qp_set_stages': (C.c_int, [vp, i32, i32, i32, P(i32)]),
        'hilo_qp_solve': (C.c_int, [vp, i64, vp, i64, vp, i64, vp, i64, vp, vp, i64, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp]),
        'hilo_qp_solve_pinned': (C.c_int, [vp, i64, vp, i64, vp, i64, vp, i64, vp, vp, i64, vp, i32, i64, vp, vp, i64, vp, vp, vp, vp, vp,
                                           vp, vp]),
        'hilo_gp_create': (C.c_int, [i32, i32, i32, vp, vp, vp, i32, vp, i32, dbl, P(vp)]),
        'hilo_gp_destroy': (None, [vp]),
        'hilo_gp_log_marginal_likelihood': (C.c_int, [vp, P(C.c_double)]),
        'hilo_gp_refit': (C.c_int, [vp, vp, i32, dbl]),
        'hilo_gp_lml_gradient': (C.c_int, [vp, i32, vp, vp, vp, vp]),
        'hilo_gp_predict': (C.c_int, [vp, i64, vp, i32, vp, vp, vp]),
        'hilo_gp_kernel_matrix': (C.c_int, [i32, i32, vp, i32, i64, vp, i64, vp, vp, vp]),
        'hilo_gp_mean': (C.c_int, [i32, i32, vp, i32, i64, vp, vp, vp]),
    }
    for name, (res, args) in sig.items():
        if not hasattr(lib, name):
            continue                     # symbol coverage is asserted by tests/test_abi.py against the header
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


def lib():
    """The loaded library.  Raises HiloError when it is absent - the product has no other compute path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HiloError(f"{LIB_PATH} not found: build it with `python -m hilo_mpc_amd._build` "
                            f"(hipcc --offload-arch=gfx950). hilo_mpc_amd has no CPU fallback.")
        try:
            _lib = _declare(C.CDLL(LIB_PATH))
        except OSError as e:
            raise HiloError(f"cannot load {LIB_PATH}: {e}") from e
    return _lib


COMPILED_ONLY = 1   # HILO_COMPILED_ONLY: create() stopped after compiling into the cache (HILO_JIT_COMPILE_ONLY)


def check(rc):
    if rc != 0:
        msg = lib().hilo_last_error().decode(errors='replace')
        if rc == -1:
            raise ValueError(msg)
        if rc == -5:
            raise NotPositiveDefinite(msg)
        err = HiloError(f"libhilo_hip error {rc}: {msg}")
        err.code = rc
        raise err
