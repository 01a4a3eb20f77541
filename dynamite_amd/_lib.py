"""
ctypes binding of libdynamite_amd.so -- the C ABI declared in
include/dynamite_amd.h.  This is the stub dynamite's Cython layer
(bpetsc.pyx / bsubspace.pyx) is replaced by.  There is no fallback: if the
library is missing it is built with hipcc, and if that fails we raise.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

_HERE = os.path.dirname(os.path.abspath(__file__))

i64p = C.POINTER(C.c_int64)
f64p = C.POINTER(C.c_double)
vp = C.c_void_p


class Subspace(C.Structure):
    """dnm_subspace"""
    _fields_ = [("type", C.c_int32), ("L", C.c_int64), ("space", C.c_int64), ("k", C.c_int64),
                ("ld_nchoosek", C.c_int64), ("nchoosek", i64p), ("dim", C.c_int64),
                ("state_map", i64p), ("rmap_indices", i64p), ("rmap_states", i64p), ("vec_swizzle", C.c_int32),
                ("site_perm", C.POINTER(C.c_int8))]


class Partition(C.Structure):
    _fields_ = [("rank", C.c_int32), ("nranks", C.c_int32)]


MULT_FN = C.CFUNCTYPE(C.c_int, vp, vp, vp)
REDUCE_FN = C.CFUNCTYPE(C.c_int, vp, f64p, C.c_int)


class Hooks(C.Structure):
    _fields_ = [("ctx", vp), ("mult", MULT_FN), ("allreduce_sum", REDUCE_FN),
                ("allreduce_max", REDUCE_FN)]


class SolverStats(C.Structure):
    _fields_ = [("reason", C.c_int32), ("its", C.c_int32), ("matvecs", C.c_int32),
                ("nconv", C.c_int32), ("err_est", C.c_double)]


# plan introspection structs (dynamite_amd/csrc/plan.h)
MAXSEG, MAXBSEG, MAXR = 4, 8, 16


LP_COUNT = 9
LP_NAMES = ["tile_real_k0", "tile_real", "tile_cplx", "tile_kvar_real", "tile_kvar_cplx", "gather_real",
            "gather_kvar_real", "gather_cplx", "gather_kvar_cplx"]
LP_KVAR, LP_CPLX, LP_GATHER = (3, 4, 6, 8), (2, 4, 7, 8), (5, 6, 7, 8)


class DevQuad(C.Structure):
    _fields_ = [("mask_tile", C.c_uint32), ("mask_loc", C.c_uint32), ("src", C.c_uint32),
                ("nslots", C.c_uint32), ("sign_tile", C.c_uint32 * 4), ("sign_ext", C.c_uint64 * 4),
                ("coeff", C.c_double * 4)]


class DevPass(C.Structure):
    _fields_ = [("nseg", C.c_int32), ("seg_off", C.c_int32 * MAXSEG), ("seg_len", C.c_int32 * MAXSEG),
                ("seg_pos", C.c_int32 * MAXSEG), ("nbseg", C.c_int32), ("bseg_off", C.c_int32 * MAXBSEG),
                ("bseg_len", C.c_int32 * MAXBSEG), ("bseg_pos", C.c_int32 * MAXBSEG),
                ("sign_base", C.c_uint64), ("accumulate", C.c_int32), ("need_tile", C.c_int32),
                ("has_diag", C.c_int32), ("cache_policy", C.c_int32), ("dext_begin", C.c_uint32), ("dext_end", C.c_uint32),
                ("dbucket", C.c_uint32 * (MAXR + 1)), ("loop", C.c_uint32 * (LP_COUNT + 1)),
                ("nquads", C.c_int32), ("n_eff", C.c_int32), ("quads", vp), ("dot_out", vp), ("zinit", vp), ("zscale", C.c_double), ("zinit2", vp), ("z2re", C.c_double), ("z2im", C.c_double),
                ("tile_bits", C.c_int32), ("log_rows", C.c_int32),
                ("swz_shift", C.c_int32), ("swz_xor_y", C.c_uint32), ("swz_xor_src", C.c_uint32), ("block_offset", C.c_uint32),
                ("pos_tmask", C.c_uint32), ("dtile", vp), ("tabs", vp), ("tabvals", vp), ("tab_loop", C.c_uint32 * 3),
                ("pad_tab", C.c_uint32), ("gbucket", C.c_uint32 * (MAXR + 1)), ("pad_g", C.c_uint32)]


class DevTab(C.Structure):
    """plan.h: DevTab"""
    _fields_ = [("mask_tile", C.c_uint32), ("mask_loc", C.c_uint32), ("src", C.c_uint32), ("first", C.c_uint32),
                ("z_tile", C.c_uint32), ("flags", C.c_uint32), ("tpos", C.c_uint32), ("twid", C.c_uint32),
                ("epos", C.c_uint32), ("ewid", C.c_uint32), ("z_ext", C.c_uint64), ("ik", C.c_uint64),
                ("ksign", C.c_uint32), ("nbits", C.c_uint32)]


class Xfer(C.Structure):
    _fields_ = [("partner", C.c_int32), ("pass_id", C.c_int32), ("offset", C.c_int64), ("count", C.c_int64)]


MAT_DEFAULT, MAT_FORCE_GATHER, MAT_USE_GLDS, MAT_HOST_ONLY = 0, 1, 2, 4
MAT_REAL_PACKED = 16      # real arithmetic for a real-symmetric operator: vectors of dim / 2 elements, two amplitudes each
EXCHANGE_AUTO, EXCHANGE_PARTNER, EXCHANGE_TRANSPOSE = 0, 1, 2      # dnm_mat_set_exchange
PHASE_ALL, PHASE_EXCHANGE, PHASE_COMPUTE = 0, 1, 2                  # dnm_comm_set_phase
MAT_AMIN_SHIFT = 8        # flags bits 8..15: log2 of the contiguous run of a window tile
WHICH = {"lowest": 0, "highest": 1, "exterior": 2}
CONVERGED_TOL, CONVERGED_ITS, DIVERGED_ITS, DIVERGED_BREAKDOWN, DIVERGED_SYMMETRY_LOST = 1, 2, -1, -2, -3

# name -> (restype, argtypes); every symbol include/dynamite_amd.h declares
SIGNATURES = {
    "dnm_last_error": (C.c_char_p, []),
    "dnm_version": (C.c_int, []),
    "dnm_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "dnm_set_device": (C.c_int, [C.c_int]),
    "dnm_malloc": (C.c_int, [C.POINTER(vp), C.c_size_t]),
    "dnm_free": (C.c_int, [vp]),
    "dnm_memcpy_h2d": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "dnm_memcpy_d2h": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "dnm_stream_synchronize": (C.c_int, [vp]),
    "dnm_subspace_dim": (C.c_int, [C.POINTER(Subspace), i64p]),
    "dnm_idx_to_state": (C.c_int, [C.POINTER(Subspace), C.c_int64, i64p, i64p]),
    "dnm_state_to_idx": (C.c_int, [C.POINTER(Subspace), C.c_int64, i64p, i64p]),
    "dnm_mat_create": (C.c_int, [C.c_int64, i64p, i64p, i64p, f64p, C.POINTER(Subspace),
                                 C.POINTER(Subspace), C.c_int, C.c_int, C.POINTER(Partition),
                                 C.POINTER(vp)]),
    "dnm_check_conserves": (C.c_int, [C.c_int64, i64p, i64p, i64p, f64p, C.POINTER(Subspace),
                                      C.POINTER(Subspace), C.c_int, C.POINTER(C.c_int), vp]),
    "dnm_reduced_density_matrix": (C.c_int, [vp, C.POINTER(Subspace), C.c_int, i64p, vp, vp]),
    "dnm_vec_set_random_swz": (C.c_int, [vp, C.c_int64, C.c_uint64, C.c_int64, C.c_int, vp]),
    "dnm_vec_swizzle_copy": (C.c_int, [vp, vp, C.c_int64, C.c_int, vp]),
    "dnm_vec_unpack_real": (C.c_int, [vp, vp, C.c_int64, C.c_int, C.c_int, vp]),
    "dnm_vec_layout_unpack_real": (C.c_int, [C.POINTER(Subspace), C.POINTER(Partition), vp, vp, vp]),
    "dnm_vec_layout_size": (C.c_int, [C.POINTER(Subspace), C.POINTER(C.c_int64)]),
    "dnm_vec_layout_partition": (C.c_int, [C.POINTER(Subspace), C.c_int, C.c_int, i64p, i64p, i64p, i64p]),
    "dnm_vec_layout_blocks": (C.c_int, [C.POINTER(Subspace), C.c_int64, i64p, i64p, i64p]),
    "dnm_vec_layout_copy": (C.c_int, [C.POINTER(Subspace), C.POINTER(Partition), vp, vp, C.c_int, vp]),
    "dnm_vec_layout_copy_f64": (C.c_int, [C.POINTER(Subspace), C.POINTER(Partition), vp, vp, C.c_int, vp]),
    "dnm_vec_layout_zero_padding": (C.c_int, [C.POINTER(Subspace), C.POINTER(Partition), vp, vp]),
    "dnm_vec_layout_positions": (C.c_int, [C.POINTER(Subspace), C.POINTER(Partition), C.c_int64, vp, vp, vp]),
    "dnm_vec_layout_positions_host": (C.c_int, [C.POINTER(Subspace), C.POINTER(Partition), C.c_int64, i64p, i64p]),
    "dnm_vec_layout_set_random": (C.c_int, [C.POINTER(Subspace), C.POINTER(Partition), vp, C.c_uint64, vp]),
    "dnm_mat_layouts": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dnm_mat_window_split": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "dnm_mat_mult_local_part": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp]),
    "dnm_mat_local_part_bits": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dnm_mat_mult_window_local": (C.c_int, [vp, vp, vp, vp]),
    "dnm_mat_mult_window_remote": (C.c_int, [vp, vp, C.c_int64, C.c_int64, vp, vp]),
    "dnm_mat_window_local_rows": (C.c_int, [vp, C.c_int64, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int), vp]),
    "dnm_mat_mult_window_rows": (C.c_int, [vp, vp, C.c_int64, C.c_int64, vp, C.c_int64, C.c_int64, vp]),
    "dnm_mat_destroy": (C.c_int, [vp]),
    "dnm_mat_sizes": (C.c_int, [vp, i64p, i64p, i64p, i64p]),
    "dnm_mat_precompute_diagonal": (C.c_int, [vp, vp]),
    "dnm_mat_get_diagonal": (C.c_int, [vp, f64p, vp]),
    "dnm_mat_mult": (C.c_int, [vp, vp, vp, vp]),
    "dnm_mat_norm_inf": (C.c_int, [vp, f64p, vp]),
    "dnm_mat_set_norm": (C.c_int, [vp, C.c_double]),
    "dnm_mat_plan_describe": (C.c_int, [vp, C.c_char_p, C.c_size_t]),
    "dnm_mat_plan_launches": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "dnm_mat_plan_counts": (C.c_int, [vp] + [C.POINTER(C.c_int)] * 6),
    "dnm_mat_export_dtile": (C.c_int, [vp, C.c_int, C.c_int, f64p, C.c_int64]),
    "dnm_mat_export_tabs": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_size_t, C.c_int, C.POINTER(C.c_int), f64p, C.c_int64, i64p]),
    "dnm_mat_export_pass": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_size_t, vp, C.c_size_t, C.c_int,
                                      C.POINTER(C.c_int)]),
    "dnm_mat_ownership": (C.c_int, [vp, i64p, i64p]),
    "dnm_mat_column_window": (C.c_int, [vp, i64p, i64p, vp]),
    "dnm_mat_column_chunks": (C.c_int, [vp, C.c_int, C.POINTER(C.c_uint8), C.c_int64, vp]),
    "dnm_mat_mult_window": (C.c_int, [vp, vp, C.c_int64, C.c_int64, vp, vp]),
    "dnm_mat_exchange_plan": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(Xfer), C.POINTER(C.c_int),
                                        C.POINTER(Xfer)]),
    "dnm_mat_mult_dot": (C.c_int, [vp, vp, vp, f64p, vp]),
    "dnm_mat_mult_lanczos": (C.c_int, [vp, vp, vp, vp, C.c_double, f64p, vp]),
    "dnm_mat_mult_local": (C.c_int, [vp, vp, vp, vp]),
    "dnm_mat_mult_remote": (C.c_int, [vp, C.c_int32, vp, vp, vp]),
    "dnm_vec_set": (C.c_int, [vp, C.c_int64, C.c_double, C.c_double, vp]),
    "dnm_vec_copy": (C.c_int, [vp, vp, C.c_int64, vp]),
    "dnm_vec_scale": (C.c_int, [vp, C.c_int64, C.c_double, C.c_double, vp]),
    "dnm_vec_axpby": (C.c_int, [vp, vp, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double, vp]),
    "dnm_vec_dot": (C.c_int, [vp, vp, C.c_int64, f64p, vp]),
    "dnm_vec_norm2": (C.c_int, [vp, C.c_int64, f64p, vp]),
    "dnm_vec_set_random": (C.c_int, [vp, C.c_int64, C.c_uint64, C.c_int64, vp]),
    "dnm_vec_mdot": (C.c_int, [vp, C.c_int64, C.c_int, vp, C.c_int64, f64p, vp]),
    "dnm_vec_maxpy": (C.c_int, [vp, vp, C.c_int64, C.c_int, C.c_int64, f64p, vp]),
    "dnm_vec_basis_update": (C.c_int, [vp, C.c_int64, C.c_int, C.c_int, C.c_int64, f64p, vp]),
    "dnm_workspace_bytes": (C.c_int, [C.POINTER(C.c_size_t)]),
    "dnm_release_workspace": (C.c_int, []),
    "dnm_mat_is_real_packed": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "dnm_comm_unique_id": (C.c_int, [C.c_void_p]),
    "dnm_comm_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "dnm_comm_destroy": (C.c_int, [C.c_void_p]),
    "dnm_comm_forget": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dnm_comm_loopback": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "dnm_comm_allreduce": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int]),
    "dnm_mat_mult_partitioned": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dnm_mat_column_ranges": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "dnm_comm_prepare": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "dnm_comm_set_phase": (C.c_int, [C.c_void_p, C.c_int]),
    "dnm_mat_set_exchange": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "dnm_mat_operator": (C.c_int, [C.c_void_p, i64p, i64p, i64p, i64p, i64p, f64p]),
    "dnm_mat_exchange_parts": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "dnm_comm_hooks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Hooks)]),
    "dnm_sc_choose_site_perm": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int64, i64p, C.c_int, C.POINTER(C.c_int8),
                                          C.POINTER(C.c_int32)]),
    "dnm_workspace_reserve": (C.c_int, [C.c_size_t, C.c_void_p]),
    "dnm_expm_chebyshev": (C.c_int, [vp, vp, vp, C.c_int64, C.c_double, C.c_double, C.POINTER(Hooks),
                                     C.POINTER(SolverStats), vp]),
    "dnm_mat_mult_sub": (C.c_int, [vp, vp, vp, vp, C.c_double, vp]),
    "dnm_mat_fuses_init": (C.c_int, [vp]),
    "dnm_mat_mult_sub2": (C.c_int, [vp, vp, vp, vp, C.c_double, vp, C.c_double, C.c_double, vp]),
    "dnm_expm_multiply": (C.c_int, [vp, vp, vp, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_int,
                                    C.c_int, C.c_size_t, C.POINTER(Hooks), C.POINTER(SolverStats), vp]),
    "dnm_eigsolve": (C.c_int, [vp, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
                               C.c_uint64, C.POINTER(Hooks), C.c_int, f64p, vp,
                               C.POINTER(SolverStats), vp]),
}


class BackendError(RuntimeError):
    """Raised when a C-ABI call returns non-zero (the reference raises petsc4py.Error)."""


_lib = None


def lib():
    global _lib
    if _lib is None:
        # torch bundles its own HIP runtime with the same SONAME (libamdhip64.so.7)
        # as /opt/rocm's: load torch first so this process has exactly one runtime
        # and torch's streams / allocations are valid handles for our launches.
        import torch  # noqa: F401
        path = _build.build()       # no-op when the in-tree .so is current
        import os
        if os.environ.get("DNM_EXPERIMENTAL") == "1":       # A/B experiments: another build of the same ABI
            path = os.environ.get("DNM_LIB", path)
        L = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)   # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise BackendError(lib().dnm_last_error().decode())


def p64(a):
    return a.ctypes.data_as(i64p)


def pf64(a):
    return a.ctypes.data_as(f64p)


def device_count():
    n = C.c_int(0)
    lib().dnm_device_count(C.byref(n))
    return n.value
