"""ctypes binding of libcipkkt.so (the C ABI declared in include/cipkkt.h).

The HIP library is the product; there is no CPU fallback: if the shared object is
missing or no GPU is usable, everything here raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CIPKKT_LIB", os.path.join(_HERE, "libcipkkt.so"))   # override: A/B two builds in one session

CONE_R, CONE_Q, CONE_S = 0, 1, 2
ROUTE_SCHUR, ROUTE_FULL3X3 = 0, 1
OP_F, OP_FT, OP_FINV, OP_FINVT = 0, 1, 2, 3
MAT_Q, MAT_A, MAT_G = 0, 1, 2
FLAG_DEVICE_PTRS = 1
FLAG_CSR_HOST = 2
E_SINGULAR = -5
E_UNSUPPORTED = -6

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)


class CipProblem(C.Structure):
    _fields_ = [("n", C.c_int), ("m", C.c_int), ("p", C.c_int), ("ncones", C.c_int),
                ("cone_type", c_int_p), ("cone_dim", c_int_p),
                ("Q", C.c_void_p), ("ldq", C.c_int),
                ("A", C.c_void_p), ("lda", C.c_int),
                ("A_rowptr", C.c_void_p), ("A_colind", C.c_void_p), ("A_val", C.c_void_p),
                ("G", C.c_void_p), ("ldg", C.c_int),
                ("route", C.c_int), ("flags", C.c_int)]


class CipOptions(C.Structure):
    _fields_ = [("optTol", C.c_double), ("DTB", C.c_double), ("infeasTol", C.c_double),
                ("refinementThreshold", C.c_double), ("maxRefinementSteps", C.c_int), ("maxIters", C.c_int),
                ("verbose", C.c_int)]


class CipResult(C.Structure):
    _fields_ = [("status", C.c_int), ("iter", C.c_int), ("mu", C.c_double), ("prFeas", C.c_double),
                ("duFeas", C.c_double), ("muFeas", C.c_double), ("pobj", C.c_double), ("dobj", C.c_double),
                ("n_factor", C.c_int), ("n_solve", C.c_int), ("trace_rows", C.c_int), ("wall_s", C.c_double)]


STATUS_NAMES = {0: "None", 1: "Optimal", 2: "Infeasible", 3: "Unbounded", 4: "Abandoned", 5: "Error"}
TRACE_COLS = 9

# name -> (restype, argtypes); every symbol include/cipkkt.h declares
SIGNATURES = {
    "cip_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, c_int_p, c_int_p, C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "cip_create_ex": (C.c_int, [C.POINTER(CipProblem), C.POINTER(C.c_void_p)]),
    "cip_update_problem": (C.c_int, [C.c_void_p, C.POINTER(CipProblem)]),
    "cip_destroy": (C.c_int, [C.c_void_p]),
    "cip_last_error": (C.c_char_p, []),
    "cip_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cip_scaling_packed_len": (C.c_size_t, [C.c_void_p]),
    "cip_set_scaling_packed": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cip_set_scaling_identity": (C.c_int, [C.c_void_p]),
    "cip_set_scaling_from_iterate_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cip_get_scaling_packed": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cip_factor": (C.c_int, [C.c_void_p]),
    "cip_check_factor": (C.c_int, [C.c_void_p]),
    "cip_set_regularization": (C.c_int, [C.c_void_p, C.c_double, C.c_int]),
    "cip_get_regularization": (C.c_int, [C.c_void_p, c_double_p, c_int_p]),
    "cip_solve3x3": (C.c_int, [C.c_void_p] + [C.c_void_p] * 6),
    "cip_solve3x3_dev": (C.c_int, [C.c_void_p] + [C.c_void_p] * 6),
    "cip_solve2x2": (C.c_int, [C.c_void_p] + [C.c_void_p] * 4),
    "cip_solve2x2_dev": (C.c_int, [C.c_void_p] + [C.c_void_p] * 4),
    "cip_solve4x4_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cip_apply_F_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "cip_cone_prod_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cip_cone_div_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cip_maxstep_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, c_double_p]),
    "cip_maxstep_pair_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, c_double_p]),
    "cip_cone_identity_dev": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cip_gemv_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_double, C.c_void_p]),
    "cip_dots_dev": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), c_int_p,
                               c_double_p]),
    "cip_axpby_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_double, C.c_void_p]),
    "cip_conicip": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(CipOptions), C.c_void_p,
                              C.c_void_p, C.c_void_p, C.POINTER(CipResult), C.c_void_p, C.c_int]),
    "cip_batch_create": (C.c_int, [C.c_int, C.POINTER(CipProblem), C.POINTER(C.c_void_p)]),
    "cip_batch_destroy": (C.c_int, [C.c_void_p]),
    "cip_batch_size": (C.c_int, [C.c_void_p]),
    "cip_batch_handle": (C.c_void_p, [C.c_void_p, C.c_int]),
    "cip_batch_conicip": (C.c_int, [C.c_void_p] + [C.POINTER(C.c_void_p)] * 3 + [C.POINTER(CipOptions)] +
                          [C.POINTER(C.c_void_p)] * 3 + [C.POINTER(CipResult), C.c_int]),
    "cip_conicip_problems": (C.c_int, [C.c_int, C.POINTER(CipProblem)] + [C.POINTER(C.c_void_p)] * 3 +
                             [C.POINTER(CipOptions)] + [C.POINTER(C.c_void_p)] * 3 + [C.POINTER(CipResult), C.c_int]),
    "cip_conicip_lockstep": (C.c_int, [C.c_int, C.POINTER(CipProblem)] + [C.POINTER(C.c_void_p)] * 3 +
                             [C.POINTER(CipOptions)] + [C.POINTER(C.c_void_p)] * 3 + [C.POINTER(CipResult)]),
    "cip_conicip_mixed": (C.c_int, [C.c_int, C.POINTER(CipProblem)] + [C.POINTER(C.c_void_p)] * 3 +
                          [C.POINTER(CipOptions)] + [C.POINTER(C.c_void_p)] * 3 + [C.POINTER(CipResult), C.c_int]),
    "cip_release_cached_memory": (C.c_int, []),
    "cip_lockstep_stats": (C.c_int, [c_int_p]),
    "cip_conicip_many": (C.c_int, [C.POINTER(C.c_void_p), C.c_int] + [C.POINTER(C.c_void_p)] * 3 +
                         [C.POINTER(CipOptions)] + [C.POINTER(C.c_void_p)] * 3 + [C.POINTER(CipResult), C.c_int]),
    "cip_ldlt_workspace_bytes": (C.c_int, [C.c_int, C.POINTER(C.c_size_t)]),
    "cip_ldlt_factor_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, c_int_p]),
    "cip_ldlt_solve_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "cip_gemm_nt_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_int,
                                  C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]),
    "cip_kkt_order": (C.c_int, [C.c_void_p, c_int_p, c_int_p]),
    "cip_get_kkt_matrix": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cip_assemble_only": (C.c_int, [C.c_void_p]),
    "cip_stats": (C.c_int, [C.c_void_p, c_double_p]),
    "cip_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "cip_set_ldlt_outer_block": (C.c_int, [C.c_int]),
    "cip_set_solve_block_max": (C.c_int, [C.c_int]),
    "cip_set_solve_fused": (C.c_int, [C.c_int]),
    "cip_lockstep_solve_block_for": (C.c_int, [C.c_int]),
    "cip_profile_trailing_thread": (C.c_int, [C.c_int]),
    "cip_profile_thread_get": (C.c_int, [c_double_p]),
    "cip_profile_kernel_thread": (C.c_int, [C.c_int, C.c_int]),
    "cip_profile_kernel_thread_get": (C.c_int, [C.c_int, c_double_p]),
    "cip_set_lazy_copy": (C.c_int, [C.c_int]),
    "cip_set_sdp_lanczos": (C.c_int, [C.c_int]),
    "cip_set_lockstep_split": (C.c_int, [C.c_int]),
    "cip_debug_chain_giveup": (C.c_int, [C.c_int]),
    "cip_get_chain_fallbacks": (C.c_int, [C.c_void_p]),
    "cip_sdp_lanczos_fallbacks": (C.c_int, [C.c_void_p, c_int_p]),
    "cip_set_ldlt_fused_chain": (C.c_int, [C.c_int]),
    "cip_set_ldlt_side_prep": (C.c_int, [C.c_int]),
    "cip_profile_trailing": (C.c_int, [C.c_void_p, C.c_int]),
    "cip_profile_get": (C.c_int, [C.c_void_p, c_double_p]),
}

_lib = None


class CipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libcipkkt error %d: %s" % (code, msg))
        self.code = code


def load():
    """Load libcipkkt.so and type every entry point.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libcipkkt.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "-- there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().cip_last_error()
        raise CipError(rc, msg.decode() if msg else "?")
