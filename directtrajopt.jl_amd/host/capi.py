"""ctypes declarations of include/dto_engine.h (kept field-for-field identical)."""
from __future__ import annotations

import ctypes as C
import os

c_double_p = C.POINTER(C.c_double)
c_int64_p = C.POINTER(C.c_int64)
c_int32_p = C.POINTER(C.c_int32)

DTO_ABI_VERSION = 7
FLAG_GENERAL_PATH_ONLY = 1
INTEGRATOR_BILINEAR, INTEGRATOR_DERIVATIVE, INTEGRATOR_EXTERNAL, INTEGRATOR_TIME_DEPENDENT_BILINEAR = 1, 2, 3, 4
OBJECTIVE_QUADRATIC, OBJECTIVE_LINEAR, OBJECTIVE_MINTIME, OBJECTIVE_KNOT_SQDIST, OBJECTIVE_EXTERNAL_KNOT, OBJECTIVE_KNOT_LOWRANK, OBJECTIVE_EXTERNAL_GLOBAL = 1, 2, 3, 4, 5, 6, 7
CONSTRAINT_NORM, CONSTRAINT_SQNORM, CONSTRAINT_EXTERNAL, CONSTRAINT_EXTERNAL_GLOBAL = 1, 2, 3, 4


class IntegratorDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("x_off", C.c_int32), ("x_dim", C.c_int32), ("u_off", C.c_int32),
                ("u_dim", C.c_int32), ("G", c_double_p),
                ("t_off", C.c_int32), ("spline_order", C.c_int32), ("substeps", C.c_int32), ("n_mod", C.c_int32),
                ("mod_kind", c_int32_p), ("mod_omega", c_double_p), ("H", c_double_p)]


class ObjectiveDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("comp_off", C.c_int32), ("comp_dim", C.c_int32), ("reserved", C.c_int32),
                ("weight", C.c_double), ("D", C.c_double), ("R", c_double_p), ("baseline", c_double_p),
                ("times", c_int64_p), ("n_times", C.c_int64), ("comps", c_int32_p), ("n_comps", C.c_int32),
                ("reserved2", C.c_int32), ("params", c_double_p), ("Qs", c_double_p),
                ("gcomps", c_int32_p), ("n_gcomps", C.c_int32), ("reserved3", C.c_int32)]


class ConstraintDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("equality", C.c_int32), ("n_comps", C.c_int32), ("g_dim", C.c_int32),
                ("comps", c_int32_p), ("c", C.c_double), ("times", c_int64_p), ("n_times", C.c_int64),
                ("jac0", c_double_p), ("hess0", c_double_p)]


class ExternalValues(C.Structure):
    _fields_ = [("values", c_double_p), ("first", c_double_p), ("second", c_double_p)]


class ProblemDesc(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device", C.c_int32), ("N", C.c_int64), ("z", C.c_int32),
                ("gd", C.c_int32), ("dt_idx", C.c_int32), ("eval_hessian", C.c_int32),
                ("n_integrators", C.c_int32), ("n_objectives", C.c_int32), ("n_constraints", C.c_int32),
                ("flags", C.c_int32), ("integrators", C.POINTER(IntegratorDesc)),
                ("objectives", C.POINTER(ObjectiveDesc)), ("constraints", C.POINTER(ConstraintDesc)),
                ("Z0", c_double_p), ("k_lo", C.c_int64), ("k_hi", C.c_int64)]


class ShardInfo(C.Structure):
    _fields_ = [("k_lo", C.c_int64), ("k_hi", C.c_int64), ("n_vars", C.c_int64), ("n_cons", C.c_int64),
                ("jac_nnz", C.c_int64), ("hess_nnz", C.c_int64), ("grad_lo", C.c_int64), ("grad_len", C.c_int64),
                ("jac_lo", C.c_int64), ("jac_len", C.c_int64), ("hess_lo", C.c_int64), ("hess_len", C.c_int64),
                ("cons_len", C.c_int64), ("n_row_segments", C.c_int32), ("reserved", C.c_int32)]


class GatherLayout(C.Structure):
    _fields_ = [("total", C.c_int64), ("padded_len", C.c_int64), ("front_pad", C.c_int64), ("own_lo", C.c_int64),
                ("own_len", C.c_int64), ("in_place_all_gather", C.c_int32), ("world", C.c_int32)]


COMM_ID_BYTES = 128
VECTOR_JACOBIAN, VECTOR_HESSIAN, VECTOR_GRADIENT, VECTOR_CONSTRAINT = 1, 2, 3, 4

H = C.c_void_p

# every symbol include/dto_engine.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "dto_create": (C.c_int, [C.POINTER(ProblemDesc), C.POINTER(H)]),
    "dto_destroy": (None, [H]),
    "dto_last_error": (C.c_char_p, [H]),
    "dto_num_vars": (C.c_int, [H, c_int64_p]),
    "dto_num_cons": (C.c_int, [H, c_int64_p]),
    "dto_num_dynamics_cons": (C.c_int, [H, c_int64_p]),
    "dto_jac_nnz": (C.c_int, [H, c_int64_p]),
    "dto_hess_nnz": (C.c_int, [H, c_int64_p]),
    "dto_features_available": (C.c_int, [H, c_int32_p, c_int32_p, c_int32_p]),
    "dto_get_shard_info": (C.c_int, [H, C.POINTER(ShardInfo)]),
    "dto_shard_rows": (C.c_int, [H, c_int64_p, c_int64_p]),
    "dto_interval_costs": (C.c_int, [H, c_double_p, C.c_int64, C.c_int64, c_double_p]),
    "dto_jacobian_structure": (C.c_int, [H, C.c_int64, C.c_int64, c_int64_p, c_int64_p]),
    "dto_hessian_structure": (C.c_int, [H, C.c_int64, C.c_int64, c_int64_p, c_int64_p]),
    "dto_constraint_bounds": (C.c_int, [H, c_double_p, c_double_p]),
    "dto_num_external": (C.c_int, [H, c_int32_p, c_int32_p, c_int32_p]),
    "dto_set_external": (C.c_int, [H, C.c_int32, C.POINTER(ExternalValues)]),
    "dto_eval_objective": (C.c_int, [H, c_double_p, c_double_p]),
    "dto_eval_gradient": (C.c_int, [H, c_double_p, c_double_p]),
    "dto_eval_constraint": (C.c_int, [H, c_double_p, c_double_p]),
    "dto_eval_jacobian": (C.c_int, [H, c_double_p, c_double_p]),
    "dto_eval_hessian": (C.c_int, [H, c_double_p, C.c_double, c_double_p, c_double_p]),
    "dto_eval_jacobian_product": (C.c_int, [H, c_double_p, c_double_p, c_double_p]),
    "dto_eval_jacobian_transpose_product": (C.c_int, [H, c_double_p, c_double_p, c_double_p]),
    "dto_eval_objective_dev": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dto_eval_gradient_dev": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dto_eval_constraint_dev": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dto_eval_jacobian_dev": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dto_eval_hessian_dev": (C.c_int, [H, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dto_comm_unique_id": (C.c_int, [C.c_void_p]),
    "dto_comm_create": (C.c_int, [H, C.c_void_p, C.c_int32, C.c_int32]),
    "dto_comm_set_ranges": (C.c_int, [H, C.c_int32, C.c_int32, c_int64_p, c_int64_p]),
    "dto_comm_destroy": (C.c_int, [H]),
    "dto_get_gather_layout": (C.c_int, [H, C.c_int32, C.POINTER(GatherLayout)]),
    "dto_gather_slabs": (C.c_int, [H, C.c_int32, c_int64_p, c_int64_p]),
    "dto_gather_jacobian_dev": (C.c_int, [H, C.c_void_p, C.c_void_p]),
    "dto_gather_hessian_dev": (C.c_int, [H, C.c_void_p, C.c_void_p]),
    "dto_gather_gradient_dev": (C.c_int, [H, C.c_void_p, C.c_void_p]),
    "dto_gather_constraint_dev": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dto_allreduce_objective_dev": (C.c_int, [H, C.c_void_p, C.c_void_p]),
    "dto_bind_output_dev": (C.c_int, [H, C.c_int32, C.c_void_p]),
    "dto_set_option": (C.c_int, [H, C.c_char_p, C.c_int64]),
    "dto_profile_enable": (C.c_int, [H, C.c_int32]),
    "dto_profile_reset": (C.c_int, [H]),
    "dto_profile_get": (C.c_int, [H, C.c_char_p, c_double_p, c_int64_p, c_double_p]),
    "dto_last_stats": (C.c_int, [H, c_int32_p, c_int32_p]),
}


def library_path():
    """In-tree engine library; DTO_ENGINE_LIB names another build of it (A/B runs of kernel variants)."""
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return os.path.join(here, os.environ.get("DTO_ENGINE_LIB", "libdto_engine.so"))


_lib = None


def load_library(path=None):
    """dlopen the engine and type every entry point.  Raises OSError if the library is missing --
    the product path has no fallback."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # torch bundles its own libamdhip64.so.7; importing it first makes the engine bind to the same
    # HIP runtime instance, so device pointers of torch tensors are valid in the engine.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(path or library_path(), mode=C.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib
