"""ctypes binding of ``libmadqp_hip.so`` (C ABI declared in ``include/madqp.h``).

There is no CPU fallback: if the shared library is missing or a call fails,
an exception is raised.  ``import torch`` happens before the library is
loaded so that the process holds exactly one HIP runtime (torch's bundled
``libamdhip64.so`` has the same SONAME the library links against).
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede CDLL: single HIP runtime per process)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmadqp_hip.so")

i32, i64, f64, vp = C.c_int32, C.c_int64, C.c_double, C.c_void_p
pf64, pi64, pi32 = C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int32)

PROF_CLASSES = ("syrk", "potrf_gemm", "potrf_diag", "potrf_trsm", "trsv", "gemv", "vec")


class MadQPError(RuntimeError):
    pass


class CState(C.Structure):
    """``madqp_state`` of include/madqp.h (device-pointer view of MPCSolver)."""

    _fields_ = (
        [("n", i64), ("m", i64), ("nlb", i64), ("nub", i64), ("ind_lb", vp), ("ind_ub", vp)]
        + [(k, vp) for k in ("x", "xl", "xu", "zl", "zu", "f", "y", "c", "jacl", "d", "p",
                             "correction_lb", "correction_ub", "reg", "pr_diag", "du_diag",
                             "l_diag", "l_lower", "u_diag", "u_lower")]
    )


pstate = C.POINTER(CState)


class CMpcOptions(C.Structure):
    """``madqp_mpc_options`` of include/madqp.h."""

    _fields_ = [("tol", f64), ("max_iter", i64), ("max_ncorr", i32), ("step_rule", i32),
                ("step_param", f64), ("regularization", i32), ("check_residual", i32),
                ("delta_p", f64), ("delta_d", f64), ("delta_min", f64), ("mu_min", f64),
                ("tol_linear_solve", f64), ("refine_steps", i32), ("kkt_form", i32)]


class CBatchData(C.Structure):
    """``madqp_batch_data`` of include/madqp.h."""

    _fields_ = [(k, vp) for k in ("H", "A", "q", "rhs", "c0", "x", "xl", "xu", "zl", "zu", "y")]


BATCH_SCALARS = ("mu", "alpha_p", "alpha_d", "obj", "inf_pr", "inf_du", "inf_compl", "dnorm", "norm_b", "norm_c",
                 "del_w", "del_c", "residual_ratio", "reg_p", "reg_d", "n_factorizations")


class CMpcInfo(C.Structure):
    """``madqp_mpc_info`` of include/madqp.h."""

    _fields_ = ([("k", i64)]
                + [(k, f64) for k in ("obj", "inf_pr", "inf_du", "inf_compl", "mu", "dnorm", "del_w",
                                      "del_c", "alpha_p", "alpha_d", "residual_ratio")]
                + [("n_factorizations", i64), ("factor_info", i32)])

# name -> argtypes (every function returns int32 unless listed in _RESTYPE)
_SIGNATURES = {
    "madqp_version": [],
    "madqp_ctx_create": [i32, vp, C.POINTER(vp)],
    "madqp_ctx_destroy": [vp],
    "madqp_last_error": [vp],
    "madqp_ctx_sync": [vp],
    "madqp_debug_inject_fault": [vp],
    "madqp_malloc": [vp, C.c_size_t, C.POINTER(vp)],
    "madqp_free": [vp, vp],
    "madqp_memcpy_h2d": [vp, vp, vp, C.c_size_t],
    "madqp_memcpy_d2h": [vp, vp, vp, C.c_size_t],
    "madqp_prof_enable": [vp, i32],
    "madqp_prof_reset": [vp],
    "madqp_prof_get": [vp, i32, pf64, pi64],
    "madqp_probe_mfma_f64": [vp, i32, pf64],
    "madqp_gen_normal": [vp, C.c_uint64, C.c_uint64, i64, vp],
    "madqp_gen_wigner": [vp, C.c_uint64, i64, f64, vp, i64],
    "madqp_syrk_assemble": [vp, i64, i64, vp, i64, vp, vp, i64, vp, vp, i64],
    "madqp_chol_create": [vp, i64, C.POINTER(vp)],
    "madqp_chol_destroy": [vp],
    "madqp_chol_set_signature": [vp, i64],
    "madqp_chol_factor": [vp, vp, i64, pi32],
    "madqp_chol_solve": [vp, vp],
    "madqp_gemv": [vp, i32, i64, i64, f64, vp, i64, vp, f64, vp],
    "madqp_set_aug_diagonal_reg": [vp, pstate, f64, f64],
    "madqp_set_initial_primal_rhs": [vp, pstate],
    "madqp_set_initial_dual_rhs": [vp, pstate],
    "madqp_set_predictive_rhs": [vp, pstate],
    "madqp_set_correction_rhs": [vp, pstate, f64],
    "madqp_get_correction": [vp, pstate],
    "madqp_set_extra_correction": [vp, pstate, f64, f64, f64, f64, f64],
    "madqp_get_complementarity_measure": [vp, pstate, pf64],
    "madqp_get_affine_complementarity_measure": [vp, pstate, f64, f64, pf64],
    "madqp_get_alpha_max": [vp, pstate, f64, pf64, pi64],
    "madqp_update_iterates": [vp, pstate, f64, f64],
    "madqp_get_inf": [vp, pstate, pf64],
    "madqp_adjust_boundary": [vp, pstate, f64],
    "madqp_reduce_rhs": [vp, pstate, vp],
    "madqp_finish_aug_solve": [vp, pstate, vp],
    "madqp_kktmul": [vp, pstate, vp, vp, f64, f64],
    "madqp_norm_inf3": [vp, i64, vp, vp, vp, pf64],
    "madqp_norm_inf": [vp, i64, vp, pf64],
    "madqp_axpy": [vp, i64, f64, vp, vp],
    "madqp_copy": [vp, i64, vp, vp],
    "madqp_fill": [vp, i64, f64, vp],
    "madqp_sp_init_duals": [vp, pstate],
    "madqp_sp_mins": [vp, pstate, pf64],
    "madqp_sp_shift": [vp, pstate, f64, f64],
    "madqp_sp_sums": [vp, pstate, pf64],
    "madqp_sp_project": [vp, pstate, f64],
    "madqp_sp_check": [vp, pstate, pi32],
    "madqp_kkt_create": [vp, i64, i64, i64, pi64, vp, i64, vp, i64, C.POINTER(vp)],
    "madqp_kkt_create_normal": [vp, i64, i64, i64, pi64, vp, i64, C.POINTER(vp)],
    "madqp_kkt_create_augmented": [vp, i64, i64, i64, pi64, vp, i64, vp, i64, C.POINTER(vp)],
    "madqp_kkt_create_scaled_augmented": [vp, i64, i64, i64, pi64, vp, i64, vp, i64, C.POINTER(vp)],
    "madqp_kkt_destroy": [vp],
    "madqp_kkt_set_aug_diagonal_reg": [vp, pstate, f64, f64],
    "madqp_kkt_initialize": [vp, pstate],
    "madqp_kkt_build": [vp, pstate],
    "madqp_kkt_factorize": [vp, pi32],
    "madqp_kkt_solve": [vp, pstate, vp],
    "madqp_kkt_set_refine": [vp, C.c_int32],
    "madqp_kkt_mul": [vp, pstate, vp, vp, f64, f64],
    "madqp_kkt_mul_solved": [vp, pstate, vp, vp, f64, f64],
    "madqp_kkt_jtprod": [vp, vp, vp],
    "madqp_kkt_eval": [vp, pstate, vp, vp, f64, pf64],
    "madqp_kkt_matrix": [vp, C.POINTER(vp), pi64],
    "madqp_kkt_create_sparse": [vp, i32, i64, i64, i64, pi64, vp, i64, vp, vp, vp, vp, vp, vp, C.POINTER(vp)],
    "madqp_kkt_set_hdiag": [vp, vp],
    "madqp_coo_map_create": [vp, i64, vp, vp, i64, i64, i32, C.POINTER(vp)],
    "madqp_coo_map_create_cols_cyclic": [vp, i64, vp, vp, i64, i64, i64, i32, i32, C.POINTER(vp)],
    "madqp_coo_map_create_tiles_cyclic": [vp, i64, vp, vp, i64, i64, i32, i32, i32, i32, C.POINTER(vp)],
    "madqp_coo_map_apply": [vp, vp, vp, i64],
    "madqp_coo_map_destroy": [vp],
    "madqp_kkt_chol": [vp, C.POINTER(vp), pi64],
    "madqp_chol_factor_begin": [vp, vp, i64],
    "madqp_chol_factor_panel": [vp, i64, i64],
    "madqp_chol_panel_pack": [vp, i64, i64, vp],
    "madqp_dist_unique_id": [vp, vp],
    "madqp_dist_create": [vp, i32, i32, i32, i32, i64, i64, vp, vp, C.POINTER(vp)],
    "madqp_dist_destroy": [vp],
    "madqp_dist_layout": [vp, pi64],
    "madqp_dist_matrix": [vp, C.POINTER(vp), pi64],
    "madqp_dist_factor": [vp, pi32],
    "madqp_dist_solve": [vp, vp],
    "madqp_dist_bytes_sent": [vp, pi64],
    "madqp_dist_comm_info": [vp, pi64],
    "madqp_dist_memory": [vp, pi64],
    "madqp_dkkt_create": [vp, i64, i64, i64, pi64, vp, i64, vp, i64, vp, i64, C.POINTER(vp)],
    "madqp_dkkt_destroy": [vp],
    "madqp_dkkt_build": [vp, pstate],
    "madqp_dkkt_factorize": [vp, pi32],
    "madqp_dkkt_solve": [vp, pstate, vp],
    "madqp_dkkt_mul": [vp, pstate, vp, vp, f64, f64],
    "madqp_dkkt_jtprod": [vp, vp, vp],
    "madqp_dkkt_eval": [vp, pstate, vp, vp, f64, pf64],
    "madqp_gen_normal_cyclic": [vp, C.c_uint64, i64, i64, i64, i32, i32, i64, vp, i64],
    "madqp_gen_wigner_cyclic": [vp, C.c_uint64, i64, f64, i64, i32, i32, i32, i32, i64, i64, vp, i64],
    "madqp_batch_create": [vp, i64, i64, i64, i64, pi64, i64, vp, i64, vp, C.POINTER(CBatchData),
                           C.POINTER(CMpcOptions), C.POINTER(vp)],
    "madqp_batch_destroy": [vp],
    "madqp_batch_init": [vp, f64, f64],
    "madqp_batch_iterate": [vp, i32, i32, pi32],
    "madqp_batch_results": [vp, pi32, pi32, pf64],
    "madqp_mpc_create": [vp, pstate, vp, vp, vp, vp, f64, f64, f64, C.POINTER(CMpcOptions), C.POINTER(vp)],
    "madqp_mpc_destroy": [vp],
    "madqp_mpc_set_scalars": [vp, f64, f64, f64, f64, i64],
    "madqp_mpc_head": [vp, C.POINTER(CMpcInfo), pi32],
    "madqp_mpc_body": [vp, C.POINTER(CMpcInfo)],
    "madqp_mpc_readbacks": [vp, C.POINTER(C.c_int64)],
    "madqp_mpc_ahead_stats": [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
}
_RESTYPE = {"madqp_last_error": C.c_char_p}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_cdll = None


def load_cdll() -> C.CDLL:
    """Load the shared library and attach the prototypes (no GPU needed)."""
    global _cdll
    if _cdll is None:
        if not os.path.exists(LIB_PATH):
            raise MadQPError(
                f"{LIB_PATH} is missing: build it with `make -C madqp_jl_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback")
        lib = C.CDLL(LIB_PATH)
        for name, args in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = _RESTYPE.get(name, i32)
        _cdll = lib
    return _cdll


def ptr(t) -> int:
    """Device address of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()
