"""
ctypes binding of libparopt_amd.so (include/paropt_amd.h).

There is NO fallback: if the shared library has not been built (``python -c 'import
__graft_entry__ as g; g.build()'`` or ``make -C paropt_amd/csrc``) importing this module raises,
and every call on a machine without a gfx950 device fails with PO_ERR_NO_DEVICE.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PAROPT_AMD_LIB") or os.path.join(_HERE, "libparopt_amd.so")  # override: A/B kernel builds

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "paropt_amd: %s is missing -- build it with `make -C paropt_amd/csrc` "
        "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH
    )

lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)

po_ctx = C.c_void_p
po_vec = C.c_void_p
po_qn = C.c_void_p
po_problem = C.c_void_p
po_ip = C.c_void_p
po_tr = C.c_void_p
po_eig = C.c_void_p
po_trsub = C.c_void_p
po_mma = C.c_void_p
c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)
c_i64_p = C.POINTER(C.c_int64)
vec_p = C.POINTER(po_vec)

ALLGATHER_FN = C.CFUNCTYPE(C.c_int, c_double_p, c_double_p, C.c_int, C.c_void_p)
ITER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)
GET_VARS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, po_vec, po_vec, po_vec)
EVAL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, po_vec, c_double_p, c_double_p)
GRAD_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, po_vec, po_vec, vec_p)
QNCORR_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, po_vec, c_double_p, po_vec, po_vec)
WRITE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, po_vec)
HVEC_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, po_vec, c_double_p, po_vec, po_vec, po_vec)
HDIAG_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, po_vec, c_double_p, po_vec, po_vec)
EIG_UPDATE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, po_vec, po_eig)
TR_ITER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)
SPARSE_CON_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, po_vec, po_vec)
SPARSE_JAC_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_double, po_vec, po_vec, po_vec)
SPARSE_OBJCON_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, po_vec, c_double_p, c_double_p, po_vec)
SPARSE_GRAD_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, po_vec, po_vec, vec_p, C.c_void_p, C.c_int64)
po_csr_symbolic = C.c_void_p
c_int_pp = C.POINTER(c_int_p)


class ProblemCallbacks(C.Structure):
    _fields_ = [
        ("user", C.c_void_p),
        ("get_vars_and_bounds", GET_VARS_FN),
        ("eval_obj_con", EVAL_FN),
        ("eval_obj_con_gradient", GRAD_FN),
        ("qn_update_correction", QNCORR_FN),
        ("write_output", WRITE_FN),
    ]


class ProblemSparseCallbacks(C.Structure):
    _fields_ = [
        ("eval_sparse_con", SPARSE_CON_FN),
        ("add_sparse_jacobian", SPARSE_JAC_FN),
        ("add_sparse_jacobian_transpose", SPARSE_JAC_FN),
        ("add_sparse_inner_product", SPARSE_JAC_FN),
    ]


# Every symbol include/paropt_amd.h declares, with its signature (restype is always int unless
# stated).  tests/test_capi_symbols.py checks this table against the header and the .so.
# po_trsub_callbacks (include/paropt_amd.h): a user-written ParOptTrustRegionSubproblem
TRSUB_GETQN_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(po_qn))
TRSUB_SIZE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_double)
TRSUB_TRIAL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, po_vec, c_double_p, po_vec, c_double_p, c_double_p)
TRSUB_ACCEPT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, po_vec, c_double_p, po_vec)
TRSUB_VOID_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)
TRSUB_MODEL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(po_vec), c_double_p, C.POINTER(po_vec),
                             C.POINTER(c_double_p), C.POINTER(vec_p), C.POINTER(po_vec), C.POINTER(po_vec))
TRSUB_BOUNDS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, po_vec, po_vec, po_vec)
TRSUB_EVAL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, po_vec, c_double_p, c_double_p)
TRSUB_GRAD_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, po_vec, po_vec, vec_p)


class TrSubCallbacks(C.Structure):
    _fields_ = [("user", C.c_void_p), ("get_quasi_newton", TRSUB_GETQN_FN), ("init_model_and_bounds", TRSUB_SIZE_FN),
                ("set_trust_region_bounds", TRSUB_SIZE_FN), ("eval_trial_step_and_update", TRSUB_TRIAL_FN),
                ("accept_trial_step", TRSUB_ACCEPT_FN), ("reject_trial_step", TRSUB_VOID_FN),
                ("get_quasi_newton_update_type", TRSUB_VOID_FN), ("get_linear_model", TRSUB_MODEL_FN),
                ("get_vars_and_bounds", TRSUB_BOUNDS_FN), ("eval_obj_con", TRSUB_EVAL_FN),
                ("eval_obj_con_gradient", TRSUB_GRAD_FN), ("sparse_constraints_are_model", C.c_int)]


class KKTDump(C.Structure):
    """po_ip_kkt_dump (include/paropt_amd.h): borrowed views of the pieces of one KKT step."""
    _fields_ = [("c", C.c_int), ("k", C.c_int), ("Dinv", po_vec), ("res_x", po_vec),
                ("res_z", c_double_p), ("res_s", c_double_p), ("res_t", c_double_p), ("res_zs", c_double_p),
                ("res_zt", c_double_p), ("res_norms", C.c_double * 4),
                ("W", c_double_p), ("G", c_double_p), ("Ce", c_double_p), ("gpiv", c_int_p), ("cpiv", c_int_p),
                ("px", po_vec), ("pzl", po_vec), ("pzu", po_vec),
                ("pz", c_double_p), ("ps", c_double_p), ("pt", c_double_p), ("pzs", c_double_p), ("pzt", c_double_p),
                ("step_mins", C.c_double * 2)]


SIGNATURES = {
    "po_last_error": (C.c_char_p, []),
    "po_version": (C.c_char_p, []),
    "po_ctx_create": (C.c_int, [C.c_int, C.POINTER(po_ctx)]),
    "po_ctx_destroy": (C.c_int, [po_ctx]),
    "po_ctx_synchronize": (C.c_int, [po_ctx]),
    "po_ctx_rank": (C.c_int, [po_ctx, c_int_p, c_int_p]),
    "po_ctx_stream": (C.c_void_p, [po_ctx]),
    "po_ctx_memcpy": (C.c_int, [po_ctx, C.c_void_p, C.c_void_p, C.c_int64, C.c_int]),
    "po_ctx_counters": (C.c_int, [po_ctx, c_i64_p, c_i64_p]),
    "po_ctx_sync_counters": (C.c_int, [po_ctx, c_i64_p, c_i64_p, c_i64_p, c_i64_p]),
    "po_ctx_algorithmic_bytes": (C.c_int, [po_ctx, c_double_p, c_double_p]),
    "po_live_objects": (C.c_int, [c_i64_p, c_i64_p]),
    "po_live_host_mirrors": (C.c_int, [c_i64_p]),
    "po_device_count": (C.c_int, [c_int_p]),
    "po_options_visit_defaults": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p]),
    "po_vec_release_array": (C.c_int, [po_vec, C.c_int]),
    "po_vec_peek_array": (C.c_int, [po_vec, C.POINTER(c_double_p)]),
    "po_ctx_time_mdot": (C.c_int, [po_ctx, C.c_int]),
    "po_ctx_time_mdot_result": (C.c_int, [po_ctx, c_double_p, c_i64_p]),
    "po_ctx_time_wgram": (C.c_int, [po_ctx, C.c_int]),
    "po_ctx_time_wgram_result": (C.c_int, [po_ctx, C.c_int, c_double_p, c_i64_p, c_int_p, c_double_p]),
    "po_ctx_comm_info": (C.c_int, [po_ctx, c_int_p, c_i64_p, c_i64_p]),
    "po_ctx_set_reduction_batching": (C.c_int, [po_ctx, C.c_int]),
    "po_ctx_batched_reductions": (C.c_int, [po_ctx, c_i64_p]),
    "po_rccl_unique_id": (C.c_int, [C.c_void_p]),
    "po_rccl_version": (C.c_int, [c_int_p]),
    "po_ctx_allreduce": (C.c_int, [po_ctx, c_double_p, C.c_int, C.c_int]),
    "po_ctx_after_reduce": (C.c_int, [po_ctx, C.c_void_p, C.c_void_p]),
    "po_ctx_reduce_device": (C.c_int, [po_ctx, C.c_void_p, C.c_int, C.c_int, c_double_p]),
    "po_problem_set_deferred_reductions": (C.c_int, [po_problem, C.c_int]),
    "po_ctx_bench_collective": (C.c_int, [po_ctx, C.c_int, C.c_int, C.c_int, c_double_p]),
    "po_ctx_comm_init_rccl": (C.c_int, [po_ctx, C.c_int, C.c_int, C.c_void_p]),
    "po_ctx_comm_init_callback": (C.c_int, [po_ctx, C.c_int, C.c_int, ALLGATHER_FN, C.c_void_p]),
    "po_vec_create": (C.c_int, [po_ctx, C.c_int64, C.POINTER(po_vec)]),
    "po_vec_incref": (C.c_int, [po_vec]),
    "po_vec_decref": (C.c_int, [po_vec]),
    "po_vec_size": (C.c_int, [po_vec, c_i64_p]),
    "po_vec_set": (C.c_int, [po_vec, C.c_double]),
    "po_vec_zero": (C.c_int, [po_vec]),
    "po_vec_copy": (C.c_int, [po_vec, po_vec]),
    "po_vec_scale": (C.c_int, [po_vec, C.c_double]),
    "po_vec_axpy": (C.c_int, [po_vec, C.c_double, po_vec]),
    "po_vec_dot": (C.c_int, [po_vec, po_vec, c_double_p]),
    "po_vec_mdot": (C.c_int, [po_vec, vec_p, C.c_int, c_double_p]),
    "po_vec_norm": (C.c_int, [po_vec, c_double_p]),
    "po_vec_maxabs": (C.c_int, [po_vec, c_double_p]),
    "po_vec_l1norm": (C.c_int, [po_vec, c_double_p]),
    "po_vec_get_array": (C.c_int, [po_vec, C.POINTER(c_double_p)]),
    "po_vec_sync_to_device": (C.c_int, [po_vec]),
    "po_vec_sync_to_host": (C.c_int, [po_vec]),
    "po_vec_get_device_array": (C.c_int, [po_vec, C.POINTER(c_double_p)]),
    "po_vec_maxpy": (C.c_int, [po_vec, C.c_double, c_double_p, vec_p, C.c_int]),
    "po_vec_fill_hash": (C.c_int, [po_vec, C.c_uint64, C.c_uint64, C.c_int64, C.c_double, C.c_double]),
    "po_qn_create": (C.c_int, [po_ctx, C.c_int, C.c_int64, C.c_int, C.POINTER(po_qn)]),
    "po_qn_destroy": (C.c_int, [po_qn]),
    "po_qn_set_update_type": (C.c_int, [po_qn, C.c_int]),
    "po_qn_set_diag_type": (C.c_int, [po_qn, C.c_int]),
    "po_qn_reset": (C.c_int, [po_qn]),
    "po_qn_update": (C.c_int, [po_qn, po_vec, po_vec, c_int_p]),
    "po_qn_mult": (C.c_int, [po_qn, po_vec, po_vec]),
    "po_qn_mult_add": (C.c_int, [po_qn, C.c_double, po_vec, po_vec]),
    "po_qn_get_compact": (
        C.c_int,
        [po_qn, c_int_p, c_double_p, C.POINTER(c_double_p), C.POINTER(c_double_p), C.POINTER(vec_p)],
    ),
    "po_qn_max_size": (C.c_int, [po_qn, c_int_p]),
    "po_problem_create_callbacks": (
        C.c_int,
        [po_ctx, C.c_int64, C.c_int, C.c_int, C.POINTER(ProblemCallbacks), C.POINTER(po_problem)],
    ),
    "po_problem_create_separable": (
        C.c_int,
        [po_ctx, C.c_int, C.c_int64, C.c_int, C.c_uint64, C.c_double, C.c_double, C.POINTER(po_problem)],
    ),
    "po_problem_set_sparse_callbacks": (
        C.c_int, [po_problem, C.c_int64, C.c_int64, C.POINTER(ProblemSparseCallbacks)]),
    "po_problem_set_hessian_callbacks": (C.c_int, [po_problem, HVEC_FN, HDIAG_FN]),
    "po_problem_set_weighting": (C.c_int, [po_problem, C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_int64]),
    "po_problem_sparse_sizes": (C.c_int, [po_problem, c_i64_p, c_i64_p]),
    "po_problem_set_sparse_jacobian_data": (
        C.c_int, [po_problem, C.c_int64, C.c_int64, c_int_p, c_int_p, SPARSE_OBJCON_FN, SPARSE_GRAD_FN]),
    "po_problem_get_sparse_jacobian_data": (
        C.c_int, [po_problem, c_int_pp, c_int_pp, C.POINTER(C.c_void_p), c_i64_p]),
    "po_problem_set_chain": (C.c_int, [po_problem, C.c_int, C.c_int, C.c_int]),
    "po_problem_set_sparse_block_size": (C.c_int, [po_problem, C.c_int]),
    "po_quasidef_factor": (C.c_int, [po_problem, po_vec, po_vec, po_vec]),
    "po_quasidef_apply": (C.c_int, [po_problem, po_vec, po_vec, po_vec, po_vec, po_vec, po_vec, po_vec]),
    "po_quasidef_factor_info": (C.c_char_p, [po_problem]),
    "po_csr_symbolic_create": (C.c_int, [C.c_int64, C.c_int64, c_int_p, c_int_p, C.POINTER(po_csr_symbolic)]),
    "po_csr_symbolic_info": (C.c_int, [po_csr_symbolic, c_i64_p]),
    "po_csr_symbolic_arrays": (
        C.c_int, [po_csr_symbolic, c_int_pp, c_int_pp, c_int_pp, c_int_pp, c_int_pp, c_int_pp]),
    "po_csr_symbolic_destroy": (C.c_int, [po_csr_symbolic]),
    "po_problem_set_bounds_mode": (C.c_int, [po_problem, C.c_int]),
    "po_problem_set_linear_constraints": (C.c_int, [po_problem, C.c_int]),
    "po_problem_set_var_bound_options": (C.c_int, [po_problem, C.c_int, C.c_int]),
    "po_problem_destroy": (C.c_int, [po_problem]),
    "po_problem_sizes": (C.c_int, [po_problem, c_i64_p, c_i64_p, c_int_p]),
    "po_problem_eval_obj_con": (C.c_int, [po_problem, po_vec, c_double_p, c_double_p]),
    "po_problem_eval_obj_con_gradient": (C.c_int, [po_problem, po_vec, po_vec, vec_p]),
    "po_problem_get_vars_and_bounds": (C.c_int, [po_problem, po_vec, po_vec, po_vec]),
    "po_ip_create": (C.c_int, [po_problem, C.POINTER(po_ip)]),
    "po_ip_destroy": (C.c_int, [po_ip]),
    "po_ip_set_option_str": (C.c_int, [po_ip, C.c_char_p, C.c_char_p]),
    "po_ip_set_option_int": (C.c_int, [po_ip, C.c_char_p, C.c_int]),
    "po_ip_set_option_float": (C.c_int, [po_ip, C.c_char_p, C.c_double]),
    "po_ip_optimize": (C.c_int, [po_ip, C.c_char_p]),
    "po_ip_get_optimized_point": (
        C.c_int,
        [po_ip, C.POINTER(po_vec), C.POINTER(c_double_p), C.POINTER(po_vec), C.POINTER(po_vec)],
    ),
    "po_ip_get_optimized_slacks": (
        C.c_int,
        [po_ip, C.POINTER(c_double_p), C.POINTER(c_double_p), C.POINTER(c_double_p), C.POINTER(c_double_p)],
    ),
    "po_ip_get_optimized_sparse": (C.c_int, [po_ip] + [C.POINTER(po_vec)] * 5),
    "po_ip_get_counters": (C.c_int, [po_ip, c_int_p, c_int_p, c_int_p]),
    "po_ip_get_barrier_parameter": (C.c_int, [po_ip, c_double_p]),
    "po_ip_get_complementarity": (C.c_int, [po_ip, c_double_p]),
    "po_ip_get_objective": (C.c_int, [po_ip, c_double_p, c_double_p]),
    "po_ip_set_penalty_gamma": (C.c_int, [po_ip, C.c_double]),
    "po_ip_set_penalty_gamma_array": (C.c_int, [po_ip, c_double_p]),
    "po_ip_set_quasi_newton": (C.c_int, [po_ip, po_qn]),
    "po_ip_reset_problem_instance": (C.c_int, [po_ip, po_problem]),
    "po_ip_get_hvec_count": (C.c_int, [po_ip, c_int_p]),
    "po_ip_reset_design_and_bounds": (C.c_int, [po_ip]),
    "po_ip_check_gradients": (C.c_int, [po_ip, C.c_double, C.POINTER(C.c_char_p)]),
    "po_ip_check_merit_func_gradient": (C.c_int, [po_ip, po_vec, C.c_double, c_double_p, c_double_p]),
    "po_ip_reset_quasi_newton": (C.c_int, [po_ip]),
    "po_ip_get_quasi_newton": (C.c_int, [po_ip, C.POINTER(po_qn)]),
    "po_ip_write_solution_file": (C.c_int, [po_ip, C.c_char_p]),
    "po_ip_read_solution_file": (C.c_int, [po_ip, C.c_char_p]),
    "po_ip_set_iteration_callback": (C.c_int, [po_ip, ITER_FN, C.c_void_p]),
    "po_ip_get_history": (C.c_int, [po_ip, C.POINTER(C.c_char_p)]),
    "po_ip_get_phase_times": (C.c_int, [po_ip, C.POINTER(C.c_char_p), C.POINTER(c_double_p), c_int_p]),
    "po_ip_set_callback_timing": (C.c_int, [po_ip, C.c_int]),
    "po_ip_debug_kkt_step": (
        C.c_int,
        [po_ip, C.c_double, C.POINTER(po_vec), C.POINTER(po_vec), C.POINTER(po_vec)]
        + [C.POINTER(c_double_p)] * 5,
    ),
    "po_ip_debug_kkt_step_sparse": (C.c_int, [po_ip] + [C.POINTER(po_vec)] * 5),
    "po_ip_debug_set_state": (C.c_int, [po_ip] + [c_double_p] * 5 + [C.c_double]),
    "po_ip_debug_kkt": (C.c_int, [po_ip, C.c_double, C.c_int, C.c_double, C.POINTER(KKTDump)]),
    "po_qn_debug_load": (C.c_int, [po_qn, C.c_int, C.c_double, c_double_p, c_double_p, c_double_p, C.c_int,
                                   vec_p, vec_p]),
    "po_tr_create": (C.c_int, [po_problem, C.POINTER(po_tr)]),
    "po_tr_destroy": (C.c_int, [po_tr]),
    "po_tr_set_option_str": (C.c_int, [po_tr, C.c_char_p, C.c_char_p]),
    "po_tr_set_option_int": (C.c_int, [po_tr, C.c_char_p, C.c_int]),
    "po_tr_set_option_float": (C.c_int, [po_tr, C.c_char_p, C.c_double]),
    "po_tr_set_eigen_model": (C.c_int, [po_tr, C.c_int, C.c_int, EIG_UPDATE_FN, C.c_void_p]),
    "po_tr_set_eigen_model_synthetic": (C.c_int, [po_tr, C.c_int, C.c_int, C.c_uint64, C.c_double]),
    "po_eig_get_approximation": (
        C.c_int,
        [po_eig, C.POINTER(c_double_p), C.POINTER(po_vec), c_int_p, C.POINTER(c_double_p),
         C.POINTER(c_double_p), C.POINTER(vec_p)],
    ),
    "po_tr_optimize": (C.c_int, [po_tr]),
    "po_tr_get_optimized_point": (C.c_int, [po_tr, C.POINTER(po_vec), C.POINTER(c_double_p), C.POINTER(po_vec)]),
    "po_tr_get_state": (
        C.c_int,
        [po_tr, c_double_p, c_int_p, c_int_p, c_int_p, C.POINTER(c_double_p), c_double_p, C.POINTER(c_double_p)],
    ),
    "po_tr_get_last_row": (C.c_int, [po_tr, C.POINTER(c_double_p), C.POINTER(C.c_char_p)]),
    "po_tr_get_last_solve_lines": (C.c_int, [po_tr, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)]),
    "po_tr_get_history": (C.c_int, [po_tr, C.POINTER(C.c_char_p)]),
    "po_tr_get_quasi_newton": (C.c_int, [po_tr, C.POINTER(po_qn)]),
    "po_tr_get_model_vectors": (C.c_int, [po_tr, C.POINTER(po_vec), C.POINTER(po_vec)]),
    "po_tr_set_iteration_callback": (C.c_int, [po_tr, TR_ITER_FN, C.c_void_p]),
    # the trust-region layer piece by piece (the reference's own assembly, eigenvalue_opt.py:298-308)
    "po_eig_create": (C.c_int, [po_problem, C.c_int, C.POINTER(po_eig)]),
    "po_eig_destroy": (C.c_int, [po_eig]),
    "po_eig_mult_add": (C.c_int, [po_eig, C.c_double, po_vec, po_vec]),
    "po_eig_eval_approximation": (C.c_int, [po_eig, po_vec, po_vec, c_double_p]),
    "po_eig_eval_approximation_gradient": (C.c_int, [po_eig, po_vec, po_vec]),
    "po_eigqn_create": (C.c_int, [po_qn, po_eig, C.c_int, C.POINTER(po_qn)]),
    "po_eigqn_set_use_quasi_newton_objective": (C.c_int, [po_qn, C.c_int]),
    "po_eigqn_update_multipliers": (C.c_int, [po_qn, c_double_p]),
    "po_eigqn_get_multiplier_index": (C.c_int, [po_qn, c_int_p]),
    "po_trsub_create_quadratic": (C.c_int, [po_problem, po_qn, C.POINTER(po_trsub)]),
    "po_trsub_create_callbacks": (C.c_int, [po_problem, C.POINTER(TrSubCallbacks), C.POINTER(po_trsub)]),
    "po_trsub_create_eigen": (C.c_int, [po_problem, po_qn, C.POINTER(po_trsub)]),
    "po_trsub_destroy": (C.c_int, [po_trsub]),
    "po_trsub_set_eigen_model_update": (C.c_int, [po_trsub, EIG_UPDATE_FN, C.c_void_p]),
    "po_trsub_problem": (C.c_int, [po_trsub, C.POINTER(po_problem)]),
    "po_trsub_sync_linear_model": (C.c_int, [po_trsub]),
    "po_infeas_create": (C.c_int, [po_trsub, C.c_int, C.c_int, C.POINTER(po_problem)]),
    "po_infeas_set_objective_scaling": (C.c_int, [po_problem, C.c_double]),
    "po_trsub_get_quasi_newton": (C.c_int, [po_trsub, C.POINTER(po_qn)]),
    "po_trsub_init_model_and_bounds": (C.c_int, [po_trsub, C.c_double]),
    "po_trsub_set_trust_region_bounds": (C.c_int, [po_trsub, C.c_double]),
    "po_trsub_eval_trial_step_and_update": (
        C.c_int, [po_trsub, C.c_int, po_vec, c_double_p, po_vec, c_double_p, c_double_p]),
    "po_trsub_accept_trial_step": (C.c_int, [po_trsub, po_vec, c_double_p, po_vec]),
    "po_trsub_reject_trial_step": (C.c_int, [po_trsub]),
    "po_trsub_get_quasi_newton_update_type": (C.c_int, [po_trsub, c_int_p]),
    "po_trsub_get_linear_model": (
        C.c_int, [po_trsub, C.POINTER(po_vec), c_double_p, C.POINTER(po_vec), C.POINTER(c_double_p),
                  C.POINTER(vec_p), C.POINTER(po_vec), C.POINTER(po_vec), c_int_p]),
    "po_tr_create_subproblem": (C.c_int, [po_trsub, C.POINTER(po_tr)]),
    "po_tr_optimize_with": (C.c_int, [po_tr, po_ip]),
    "po_tr_initialize": (C.c_int, [po_tr]),
    "po_tr_set_penalty_gamma": (C.c_int, [po_tr, C.c_double]),
    "po_tr_set_penalty_gamma_array": (C.c_int, [po_tr, c_double_p]),
    "po_mma_create": (C.c_int, [po_problem, C.POINTER(po_mma)]),
    "po_mma_destroy": (C.c_int, [po_mma]),
    "po_mma_set_option_str": (C.c_int, [po_mma, C.c_char_p, C.c_char_p]),
    "po_mma_set_option_int": (C.c_int, [po_mma, C.c_char_p, C.c_int]),
    "po_mma_set_option_float": (C.c_int, [po_mma, C.c_char_p, C.c_double]),
    "po_mma_optimize": (C.c_int, [po_mma]),
    "po_mma_get_optimized_point": (
        C.c_int, [po_mma, C.POINTER(po_vec), C.POINTER(c_double_p)] + [C.POINTER(po_vec)] * 3),
    "po_mma_get_asymptotes": (C.c_int, [po_mma, C.POINTER(po_vec), C.POINTER(po_vec)]),
    "po_mma_get_state": (C.c_int, [po_mma, c_int_p, c_int_p, c_double_p, C.POINTER(c_double_p)]),
    "po_mma_get_last_row": (C.c_int, [po_mma, C.POINTER(c_double_p)]),
    "po_mma_get_history": (C.c_int, [po_mma, C.POINTER(C.c_char_p)]),
    "po_mma_set_iteration_callback": (C.c_int, [po_mma, TR_ITER_FN, C.c_void_p]),
    "po_wgram": (C.c_int, [po_vec, vec_p, C.c_int, c_double_p]),
    "po_wgram_with_rhs": (C.c_int, [po_vec, vec_p, C.c_int, c_double_p]),
    "po_group_panel": (C.c_int, [po_vec, vec_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_double, vec_p]),
    "po_wgram_with_groups": (C.c_int, [po_vec, vec_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_double,
                                       vec_p, C.c_int, c_double_p, c_int_p]),
    "po_bench_mdot": (C.c_int, [po_vec, vec_p, C.c_int, C.c_int, c_double_p, c_double_p]),
    "po_bench_wgram": (C.c_int, [po_vec, vec_p, C.c_int, C.c_int, c_double_p]),
    "po_qn_get_pivots": (C.c_int, [po_qn, c_int_pp, c_int_p]),
    "po_ip_get_debug_ints": (C.c_int, [po_ip, c_int_pp, c_int_p, c_int_p, c_i64_p]),
    "po_ip_get_bounds": (C.c_int, [po_ip, C.POINTER(po_vec), C.POINTER(po_vec)]),
    "po_bench_kernels": (C.c_int, [po_ctx, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]),
    "po_bench_stream": (C.c_int, [po_vec, po_vec, C.c_int, C.c_int, c_double_p]),
    "po_bench_vec_api": (C.c_int, [po_ctx, C.c_int64, C.c_int, C.c_char_p, C.c_int]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here == the .so does not export a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args


class ParOptAMDError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("paropt_amd error %d: %s" % (code, text))
        self.code = code


def check(rc):
    if rc != 0:
        raise ParOptAMDError(rc, lib.po_last_error().decode(errors="replace"))
    return rc
