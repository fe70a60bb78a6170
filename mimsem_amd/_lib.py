"""ctypes binding of libmimsem_hip.so (include/mimsem_hip.h).  The product path: there is NO CPU
fallback -- if the HIP library is missing or a call fails this module raises."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MIMSEM_LIB") or os.path.join(_HERE, "libmimsem_hip.so")      # MIMSEM_LIB: another build of the same library (A/B runs)

c_dp = C.c_void_p          # device / host pointers travel as plain addresses
c_ll = C.c_longlong


class MeshDesc(C.Structure):
    """mimsem_mesh_desc"""
    _fields_ = [("elOrd", C.c_int), ("quadOrd", C.c_int), ("nEl", C.c_int), ("nk", C.c_int),
                ("n0", C.c_int), ("n1", C.c_int), ("n2", C.c_int),
                ("inds0", C.c_void_p), ("inds1x", C.c_void_p), ("inds1y", C.c_void_p), ("inds2", C.c_void_p),
                ("det", C.c_void_p), ("J", C.c_void_p), ("thick", C.c_void_p), ("thickInv", C.c_void_p),
                ("indsq", C.c_void_p), ("nq", C.c_int)]


class MimsemError(RuntimeError):
    pass


_SIGS = {
    "mimsem_abi_version": (C.c_int, []),
    "mimsem_build_has_experiments": (C.c_int, []),
    "mimsem_strerror": (C.c_char_p, [C.c_int]),
    "mimsem_last_hip_error": (C.c_char_p, []),
    "mimsem_device_count": (C.c_int, []),
    "mimsem_ctx_create": (C.c_int, [C.POINTER(MeshDesc), C.c_int, C.POINTER(C.c_void_p)]),
    "mimsem_ctx_destroy": (None, [C.c_void_p]),
    "mimsem_ctx_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mimsem_ctx_use_own_stream": (C.c_int, [C.c_void_p]),
    "mimsem_ctx_sync": (C.c_int, [C.c_void_p]),
    "mimsem_ctx_set_levels": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mimsem_ctx_workspace_bytes": (c_ll, [C.c_void_p]),
    "mimsem_op_level_chunk": (C.c_int, [C.c_void_p, C.c_int]),
    "mimsem_op_wave_stats": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "mimsem_ctx_set_halo_slots": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "mimsem_op_apply_part": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_uint, c_dp, c_ll, c_dp, c_ll, c_dp, c_ll, C.c_double, C.c_int]),
    "mimsem_ctx_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "mimsem_ctx_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(c_ll)]),
    "mimsem_malloc": (C.c_int, [C.POINTER(C.c_void_p), c_ll]),
    "mimsem_free": (C.c_int, [C.c_void_p]),
    "mimsem_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, c_ll]),
    "mimsem_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, c_ll]),
    "mimsem_memset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, c_ll]),
    "mimsem_op_apply": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_uint,
                                  c_dp, c_ll, c_dp, c_ll, c_dp, c_ll, C.c_double]),
    "mimsem_op_apply_up": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_uint,
                                     c_dp, c_ll, c_dp, c_ll, c_dp, c_ll, c_dp, c_ll, C.c_double]),
    "mimsem_elem_blocks_apply": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_uint, c_dp, c_ll, c_dp, c_ll, c_dp, c_ll, c_dp, c_ll, C.c_double]),
    "mimsem_op_elmat_size": (C.c_int, [C.c_void_p, C.c_int]),
    "mimsem_op_element_matrices": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_uint, c_dp, c_dp]),
    "mimsem_op_element_matrices_ex": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_uint, c_dp, c_dp, c_dp]),
    "mimsem_pvec": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, c_dp, c_ll, c_dp, c_ll]),
    "mimsem_incidence_apply": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_dp, c_ll, c_dp, c_ll]),
    "mimsem_interp_quad": (C.c_int, [C.c_void_p, C.c_int, C.c_uint, C.c_int, c_dp, c_ll, c_dp, c_ll]),
    "mimsem_sw_operator_apply": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, c_dp, c_ll, c_dp, c_ll, c_dp, c_ll]),
    "mimsem_sw_blocks_apply": (C.c_int, [C.c_void_p, C.c_int, c_dp, c_dp, c_ll, c_dp, c_ll]),
    "mimsem_op_richardson_sweep": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_uint,
                                             c_dp, c_ll, c_dp, c_ll, c_dp, c_ll, c_dp, c_ll, c_dp, c_ll, c_dp, c_ll]),
    "mimsem_op_chebyshev_sweep": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_uint,
                                            c_dp, c_ll, c_dp, c_ll, c_dp, c_ll, c_dp, c_ll, C.c_double, C.c_double, c_dp, c_ll, c_dp, c_ll, c_dp, c_ll]),
    "mimsem_block_richardson_sweep": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_uint,
                                                c_dp, c_ll, c_dp, c_dp, c_ll, c_dp, c_ll, c_dp, c_ll]),
    "mimsem_block_chebyshev_sweep": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_uint, c_dp, c_ll, c_dp, c_dp, c_ll,
                                               c_dp, c_ll, C.c_double, C.c_double, c_dp, c_ll, c_dp, c_ll, c_dp, c_ll]),
    "mimsem_block_chebyshev_solve": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_uint, c_dp, c_ll, c_dp, c_dp, c_ll,
                                               c_dp, c_ll, C.c_int, C.POINTER(C.c_double), c_dp, c_ll, c_dp, c_ll, c_dp, c_ll]),
    "mimsem_sw_operator_precond_chebyshev": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, c_dp, c_ll, c_dp, C.c_double, C.c_double,
                                             c_dp, c_ll, c_dp, c_ll, c_dp, c_ll]),
    "mimsem_sw_chebyshev_step2": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, c_dp, c_ll, c_dp, C.c_int, C.c_double, C.c_double, C.c_double,
                                           C.c_double, c_dp, c_ll, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_ll]),
    "mimsem_sw_chebyshev_flush": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, c_dp, c_ll, c_dp, c_dp, c_ll]),
    "mimsem_sw_operator_precond_apply": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, c_dp, c_ll, c_dp, c_dp, c_ll, c_dp, c_ll]),
    "mimsem_sw_operator_precond_orthogonalize": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double, c_dp, c_dp, c_dp, c_dp, C.c_int, c_dp, c_ll, C.c_double, c_dp]),
    "mimsem_krylov_reorthonormalize": (C.c_int, [C.c_void_p, C.c_int, c_ll, c_dp, c_ll, c_dp, c_dp, c_dp, c_dp, c_dp, C.c_int]),
    "mimsem_krylov_reorthonormalize_ex": (C.c_int, [C.c_void_p, C.c_int, c_ll, c_dp, c_ll, c_dp, c_dp, c_dp, c_dp, c_dp, C.c_int, C.c_int, C.c_void_p]),
    "mimsem_krylov_cgs2": (C.c_int, [C.c_void_p, C.c_int, c_ll, c_dp, c_ll, c_dp, c_dp, c_dp, c_dp, c_dp, C.c_int, C.c_void_p]),
    "mimsem_ksp_create": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "mimsem_ksp_destroy": (None, [C.c_void_p]),
    "mimsem_ksp_set_operator": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_uint, c_dp, c_ll]),
    "mimsem_ksp_set_operator_sw": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, c_dp, c_ll]),
    "mimsem_ksp_set_operator_shell": (C.c_int, [C.c_void_p, C.c_int, c_ll, C.c_void_p, C.c_void_p]),
    "mimsem_ksp_set_pc_none": (C.c_int, [C.c_void_p]),
    "mimsem_ksp_set_pc_jacobi": (C.c_int, [C.c_void_p, c_dp, c_ll]),
    "mimsem_ksp_set_pc_bjacobi": (C.c_int, [C.c_void_p]),
    "mimsem_ksp_set_pc_elem_blocks": (C.c_int, [C.c_void_p, C.c_int, c_dp, c_dp, c_ll]),
    "mimsem_ksp_set_pc_sw_blocks": (C.c_int, [C.c_void_p, c_dp]),
    "mimsem_ksp_set_pc_sw_bjacobi": (C.c_int, [C.c_void_p]),
    "mimsem_ksp_set_pc_shell": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mimsem_ksp_set_tolerances": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int]),
    "mimsem_ksp_set_initial_guess_nonzero": (C.c_int, [C.c_void_p, C.c_int]),
    "mimsem_ksp_solve": (C.c_int, [C.c_void_p, c_dp, c_ll, c_dp, c_ll]),
    "mimsem_ksp_get_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "mimsem_colop_apply_blocks": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_dp, c_dp, c_dp]),
    "mimsem_l2_transpose": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_dp, c_ll, c_dp]),
    "mimsem_colop_nblocks": (C.c_int, [C.c_void_p, C.c_int]),
    "mimsem_colop_blocks": (C.c_int, [C.c_void_p, C.c_int, C.c_uint, c_dp, c_dp, c_dp]),
    "mimsem_colop_apply": (C.c_int, [C.c_void_p, C.c_int, C.c_uint, C.c_int, c_dp, c_dp, c_dp, c_dp]),
    "mimsem_column_eos": (C.c_int, [C.c_void_p, C.c_int, c_dp, c_dp, C.c_double, C.c_double, c_dp]),
    "mimsem_column_diag_theta": (C.c_int, [C.c_void_p, C.c_int, c_dp, c_dp, c_dp]),
    "mimsem_column_newton_residual": (C.c_int, [C.c_void_p, C.c_double, C.c_double] + [c_dp]*22),
    "mimsem_column_newton_update": (C.c_int, [C.c_void_p] + [c_dp]*17),
    "mimsem_column_max_norms": (C.c_int, [C.c_void_p, c_dp, c_dp, c_dp]),
    "mimsem_column_diag_theta_blend": (C.c_int, [C.c_void_p, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, C.c_double, C.c_double]),
    "mimsem_column_solve_schur_eta": (C.c_int, [C.c_void_p, C.c_double] + [c_dp]*12),
    "mimsem_column_helmholtz_blocks": (C.c_int, [C.c_void_p, C.c_double] + [c_dp]*5),
    "mimsem_colop_blocks_ex": (C.c_int, [C.c_void_p, C.c_int, C.c_uint, C.c_double, c_dp, c_dp, c_dp, c_ll, c_dp]),
    "mimsem_colop_apply_ex": (C.c_int, [C.c_void_p, C.c_int, C.c_uint, C.c_int, C.c_double, c_dp, c_dp, c_dp, c_ll, c_dp, c_dp]),
    "mimsem_column_incidence": (C.c_int, [C.c_void_p, C.c_int, c_dp, c_dp]),
    "mimsem_column_diag_theta_up": (C.c_int, [C.c_void_p, C.c_double, c_dp, c_dp, c_dp, c_ll, c_dp]),
    "mimsem_column_temp_forcing_hs": (C.c_int, [C.c_void_p, c_dp, c_dp, c_dp, c_dp, c_dp]),
    "mimsem_column_solve_schur_3": (C.c_int, [C.c_void_p, C.c_double, C.c_uint] + [c_dp]*14),
    "mimsem_krylov_orthogonalize": (C.c_int, [C.c_void_p, C.c_int, c_ll, c_dp, c_ll, C.c_double, c_dp, c_dp]),
    "mimsem_krylov_normalize": (C.c_int, [C.c_void_p, c_ll, c_dp, c_dp, C.c_int, c_dp, c_dp, c_dp, C.c_int]),
    "mimsem_krylov_mdot": (C.c_int, [C.c_void_p, C.c_int, c_ll, c_dp, c_ll, c_dp, c_dp]),
    "mimsem_krylov_maxpy": (C.c_int, [C.c_void_p, C.c_int, c_ll, c_dp, c_ll, c_dp, C.c_double, c_dp]),
    "mimsem_krylov_rowdot": (C.c_int, [C.c_void_p, C.c_int, c_ll, c_dp, c_ll, c_dp, c_ll, c_dp]),
    "mimsem_krylov_cg_update": (C.c_int, [C.c_void_p, C.c_int, c_ll, c_dp, c_dp, c_dp, c_ll, c_dp, c_ll, c_dp, c_ll, c_dp, c_ll]),
    "mimsem_krylov_cg_direction": (C.c_int, [C.c_void_p, C.c_int, c_ll, c_dp, c_dp, c_dp, c_ll, c_dp, c_ll]),
    "mimsem_block_inverse": (C.c_int, [C.c_void_p, c_ll, C.c_int, c_dp]),
    "mimsem_block_inverse_status": (C.c_int, [C.c_void_p, c_ll, C.c_int, c_dp, C.POINTER(C.c_int)]),
    "mimsem_vec_combine": (C.c_int, [C.c_void_p, C.c_int, c_ll, C.c_double, c_dp, c_ll, C.c_int, c_dp, c_ll, C.c_double, c_dp, c_ll, c_dp, c_ll]),
    "mimsem_interface_average": (C.c_int, [C.c_void_p, C.c_int, c_ll, c_dp, c_ll, c_dp, c_ll]),
    "mimsem_halo_segments": (C.c_int, [C.c_void_p, c_dp, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, c_dp, c_dp, c_ll]),
    "mimsem_halo_pack": (C.c_int, [C.c_void_p, c_dp, C.c_int, C.c_int, c_dp, c_ll, c_dp]),
    "mimsem_halo_unpack": (C.c_int, [C.c_void_p, c_dp, C.c_int, C.c_int, C.c_int, c_dp, c_dp, c_ll]),
    "mimsem_halo_create": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                     C.POINTER(C.c_void_p)]),
    "mimsem_halo_destroy": (None, [C.c_void_p]),
    "mimsem_halo_set_rccl": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mimsem_halo_use_rccl_library": (C.c_int, [C.c_void_p]),
    "mimsem_op_apply_part_reset": (C.c_int, [C.c_void_p]),
    "mimsem_krylov_gs_control": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "mimsem_column_solve_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.c_void_p]),
    "mimsem_column_set_pivot_fallback": (C.c_int, [C.c_void_p, C.c_int]),
    "mimsem_krylov_chebyshev_start": (C.c_int, [C.c_void_p, C.c_int, c_ll, C.c_double, C.c_double, c_dp, c_ll, c_dp, c_ll, c_dp, c_ll, c_dp, c_ll]),
    "mimsem_krylov_axpy_dots": (C.c_int, [C.c_void_p, c_ll, c_dp, c_dp, c_dp]),
    "mimsem_krylov_chebyshev_px": (C.c_int, [C.c_void_p, C.c_int, c_ll, C.c_double, C.c_double, c_dp, c_ll, c_dp, c_ll, c_dp, c_ll, c_dp, c_ll, c_dp, c_ll, c_dp, c_ll]),
    "mimsem_sw_dual_chebyshev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mimsem_halo_peer_export": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "mimsem_halo_set_peer": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "mimsem_halo_peer_status": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mimsem_hessenberg_eigenvalues": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mimsem_column_flag_for_test": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "mimsem_krylov_chebyshev_update": (C.c_int, [C.c_void_p, C.c_int, c_ll, C.c_double, C.c_double, c_dp, c_ll, c_dp, c_ll, c_dp, c_ll, c_dp, c_ll]),
    "mimsem_selftest_rows_half": (C.c_int, [C.c_void_p, C.c_int, c_dp, c_dp, c_dp, c_dp, c_dp]),
    "mimsem_ksp_get_pc_blocks": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "mimsem_ksp_ritz": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "mimsem_graph_begin": (C.c_int, [C.c_void_p]),
    "mimsem_graph_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "mimsem_graph_launch": (C.c_int, [C.c_void_p]),
    "mimsem_graph_num_nodes": (C.c_int, [C.c_void_p]),
    "mimsem_graph_destroy": (None, [C.c_void_p]),
    "mimsem_halo_set_transport": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mimsem_halo_set_loopback": (C.c_int, [C.c_void_p]),
    "mimsem_halo_begin": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_dp, c_ll]),
    "mimsem_halo_end": (C.c_int, [C.c_void_p]),
}

# signature of the host transport of mimsem_halo_set_transport (include/mimsem_hip.h: mimsem_halo_transport_fn)
HALO_PEER_BLOB = 1024        # MIMSEM_HALO_PEER_BLOB
HALO_TRANSPORT = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_longlong), C.c_void_p, C.POINTER(C.c_longlong), C.c_int,
                             C.POINTER(C.c_int), C.c_void_p)

OPS = dict(UMAT=0, WMAT=1, UHMAT=2, PMAT=3, PHMAT=4, WTQUMAT=5, ROTMAT=6, WHMAT=7, UTMAT=8,
           UTMAT_H=9, UTQWMAT=10, WTQDUDZ=11, WMATINV=12, WHMATINV=13, PHMAT_UP=14, ROTMAT_UP=15, WTQ=16, PTQ=17, UTQ=18, UMAT_UP=19, UHMAT_UP=20, UVEC_HU_UP=21, UMAT_RAY=22)
COLOPS = dict(CONST=0, CONST_INV=1, CONST_RHO=2, CONST_RHO_INV=3, CONST_THETA=4, EOS_BLOCK=5,
              LINEAR=6, LINEAR_INV=7, LINEAR_RT=8, LINEAR_THETA=9, LINEAR_RHO2=10, RAYLEIGH=11,
              LINCON=12, LINCON2=13, CONLIN=14, CONLIN_W=15, CONLIN_RHODPI=16,
              LINEAR_RAYLEIGH_INV=17, EOS_BLOCK_INV=18, LINEAR_RHO2_UP=19, LINCON2_UP=20)
FLAG_VERT = 1
FLAG_ACCUM = 2
FLAG_TRANSPOSE = 4

_lib = None


def build():
    """Compile the HIP library for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc")])


def exported_symbols():
    return sorted(_SIGS)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MimsemError(f"{LIB_PATH} is missing: build it with mimsem_amd._lib.build() "
                              "(there is no CPU fallback for the operator engine)")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)      # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        if L.mimsem_abi_version() != 1:
            raise MimsemError("libmimsem_hip.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        L = lib()
        msg = L.mimsem_strerror(rc).decode()
        if rc == -3:
            msg += " [" + L.mimsem_last_hip_error().decode() + "]"
        raise MimsemError(f"{what} failed: {msg} (code {rc})")
