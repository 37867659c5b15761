"""ctypes binding of libwcsph_hip.so (include/wcsph.h).  Fails loudly when the HIP
library is missing -- there is deliberately no CPU fallback in the product path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WCSPH_LIB") or os.path.join(_HERE, "libwcsph_hip.so")   # WCSPH_LIB: development builds (tools/)

MODE_FAITHFUL, MODE_CLEAN = 0, 1
COMM_ID_BYTES = 128   # WCSPH_COMM_ID_BYTES (sizeof(ncclUniqueId))

ERRORS = {-1: "bad argument", -2: "HIP runtime error", -3: "hash bucket overflow (> maxInGrid)",
          -4: "neighbour list overflow (> maxNeighbour)", -5: "non-finite reduction"}


class Params(C.Structure):
    """struct wcsph_params (include/wcsph.h)"""
    _fields_ = [
        ("n_liquid", C.c_int32), ("n_solid", C.c_int32), ("neighbor_mode", C.c_int32), ("max_in_grid", C.c_int32),
        ("max_neighbour", C.c_int32), ("hash_table_size", C.c_int32), ("tension_fixed", C.c_int32),
        ("list_cap", C.c_int32),
        ("particle_radius", C.c_double), ("cfl_radius", C.c_double), ("user_max_t", C.c_double),
        ("user_min_t", C.c_double),
        ("rho_L0", C.c_float), ("rho_S0", C.c_float), ("VL0", C.c_float), ("VS0", C.c_float),
        ("liqiudMass", C.c_float), ("gravity", C.c_float * 3),
        ("dim_coff", C.c_float), ("viscosity", C.c_float), ("viscosity_b", C.c_float), ("viscosity_err", C.c_float),
        ("tension_coff", C.c_float), ("tension_coff_b", C.c_float),
        ("viscosity_omega", C.c_float), ("vorticity_coff", C.c_float), ("vorticity_init", C.c_float),
        ("eps", C.c_float),
        ("ss_searchR", C.c_float), ("ss_mk", C.c_float), ("ss_ml", C.c_float), ("ss_VL0", C.c_float),
        ("ss_VS0", C.c_float), ("ss_mass", C.c_float), ("ss_rho0", C.c_float), ("ss_stiff", C.c_float),
        ("ss_dim", C.c_float), ("ss_visc", C.c_float), ("ss_visc_b", C.c_float), ("list_radius", C.c_float),
        ("pci_coff", C.c_float), ("pc_visc", C.c_float), ("pc_visc_b", C.c_float), ("ii_visc", C.c_float),
        ("ii_visc_b", C.c_float), ("ii_omega", C.c_float),
    ]


# enum wcsph_field
FIELDS = ["pos", "vel", "vel_guess", "d_vel", "omega", "d_omega", "normal", "cg_r", "cg_dir", "cg_Ad", "cg_s",
          "rho", "adv_rho", "pressure", "vel_max", "alpha_coff", "kappa", "kappa_v", "cg_Minv", "neighborCount", "kbuf",
          "d_vel_pre", "pos_star", "vel_star", "d_ii", "dij_pj", "a_ii", "pressure_pre",
          "color", "color_grad", "pos_avr", "G"]
FIELD_ID = {n: i for i, n in enumerate(FIELDS)}
# enum wcsph_scalar
SCALARS = ["deltaT", "avg_density_err", "cg_delta", "cg_delta_old", "cg_delta_zero", "viscosity", "viscosity_b",
           "tension_coff", "tension_coff_b", "vel_max0", "rho_err", "table_updates_in_place", "dt_rule_mismatches"]
SCALAR_ID = {n: i for i, n in enumerate(SCALARS)}

# every entry point include/wcsph.h declares that takes only the context
CTX_ONLY = [
    "wcsph_destroy", "wcsph_sync", "wcsph_grid_update",
    "wcsph_dfsph_reset_param", "wcsph_dfsph_compute_density", "wcsph_dfsph_compute_coff",
    "wcsph_dfsph_warmstart_divergence", "wcsph_dfsph_begin_divergence", "wcsph_dfsph_divergence_iter",
    "wcsph_dfsph_end_divergence", "wcsph_dfsph_warmstart_pressure", "wcsph_dfsph_begin_pressure",
    "wcsph_dfsph_pressure_iter", "wcsph_dfsph_end_pressure", "wcsph_dfsph_clear_nonpressure",
    "wcsph_dfsph_tension", "wcsph_dfsph_visc_init", "wcsph_dfsph_visc_iter", "wcsph_dfsph_visc_end",
    "wcsph_dfsph_vorticity", "wcsph_dfsph_cfl_reduce", "wcsph_dfsph_update_vel", "wcsph_dfsph_update_pos",
    "wcsph_dfsph_compute_nonpressure_force", "wcsph_dfsph_optimize_time_step", "wcsph_dfsph_solve_divergence",
    "wcsph_dfsph_solve_pressure", "wcsph_dfsph_step",
    "wcsph_sesph_reset_param", "wcsph_sesph_density", "wcsph_sesph_pressure", "wcsph_sesph_force",
    "wcsph_sesph_integrate", "wcsph_sesph_step",
    "wcsph_pcisph_reset_param", "wcsph_pcisph_compute_nonpressure_force", "wcsph_pcisph_init_iter_info",
    "wcsph_pcisph_update_iter_info", "wcsph_pcisph_predict_density", "wcsph_pcisph_update_pos",
    "wcsph_pcisph_solve_pressure", "wcsph_pcisph_step",
    "wcsph_iisph_reset_param", "wcsph_iisph_compute_density", "wcsph_iisph_visc_init", "wcsph_iisph_visc_iter",
    "wcsph_iisph_combine_nonpressure", "wcsph_iisph_compute_nonpressure_force", "wcsph_iisph_compute_advection",
    "wcsph_iisph_update_iter_info", "wcsph_iisph_update_pressure_force", "wcsph_iisph_solve_pressure",
    "wcsph_iisph_update_pos", "wcsph_iisph_step", "wcsph_surface_color_map",
]
OTHER = ["wcsph_create", "wcsph_last_error", "wcsph_set_stream", "wcsph_device_count", "wcsph_setup",
         "wcsph_get_block_size", "wcsph_field_to_host", "wcsph_field_from_host", "wcsph_get_scalar",
         "wcsph_set_scalar", "wcsph_get_iters", "wcsph_set_iters", "wcsph_neighbors_to_host", "wcsph_compact_lists_to_host", "wcsph_list_stats", "wcsph_time_kernel", "wcsph_state_classify", "wcsph_slab_set_tail_classify", "wcsph_slab_tail_classify_result", "wcsph_slab_migrate_begin", "wcsph_slab_migrate_end", "wcsph_slab_halo_lists", "wcsph_slab_tail_neighbour_counts", "wcsph_slab_migrate", "wcsph_slab_migrate_set_seam", "wcsph_state_gather_rows", "wcsph_state_apply", "wcsph_comm_unique_id", "wcsph_comm_last_error", "wcsph_comm_init", "wcsph_comm_destroy", "wcsph_comm_set_neighbours", "wcsph_comm_set_mode", "wcsph_comm_set_halo_split", "wcsph_halo_exchange", "wcsph_comm_allreduce", "wcsph_comm_stats", "wcsph_comm_detach", "wcsph_comm_attach", "wcsph_profile_begin", "wcsph_profile_end", "wcsph_profile_laps_begin", "wcsph_profile_laps_end",
         "wcsph_field_gather_rows", "wcsph_field_scatter_rows", "wcsph_set_owned", "wcsph_set_cell_window",
         "wcsph_load_state", "wcsph_load_state_split", "wcsph_store_state", "wcsph_get_acc", "wcsph_set_acc", "wcsph_phase", "wcsph_set_part", "wcsph_set_acc_buffer",
         "wcsph_surface_anisotropic_kernel", "wcsph_dfsph_step_fixed", "wcsph_sample_surface", "wcsph_mc_tables",
         "wcsph_mc_setup", "wcsph_mc_surface_value", "wcsph_mc_value_to_host", "wcsph_mc_marching_cube", "wcsph_canvas_draw",
         "wcsph_set_owned_columns", "wcsph_set_global_liquid", "wcsph_grid_update_begin", "wcsph_grid_update_mid", "wcsph_grid_update_end", "wcsph_bucket_hist", "wcsph_bucket_hist_pack", "wcsph_bucket_hist_unpack",
         "wcsph_slab_classify", "wcsph_halo_pack", "wcsph_halo_unpack", "wcsph_slab_set_halo", "wcsph_slab_step", "wcsph_hashgrid_count_to_host", "wcsph_hashgrid_rows_to_host", "wcsph_field_rows_to_host", "wcsph_field_rows_from_host"]
ALL_SYMBOLS = CTX_ONLY + OTHER

# include/wcsph.h: wcsph_slab_callbacks (the caller's communicator behind wcsph_slab_step)
CB_BUFFERS = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p))
CB_POST = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32)
CB_WAIT = C.CFUNCTYPE(C.c_int, C.c_void_p)
CB_ACC = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32)
CB_MAX = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double))
CB_WORDS = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64)
CB_MAXBITS = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)


class SlabCallbacks(C.Structure):
    _fields_ = [("user", C.c_void_p), ("world", C.c_int32), ("halo_buffers", CB_BUFFERS), ("halo_post", CB_POST),
                ("halo_wait", CB_WAIT), ("allreduce_acc", CB_ACC), ("allreduce_max", CB_MAX), ("allreduce_words", CB_WORDS),
                ("allreduce_max_bits", CB_MAXBITS), ("allreduce_words_post", CB_WORDS), ("allreduce_words_wait", CB_WAIT)]


PHASES = ["DENSITY_COFF", "DRHO_DIV_WARM", "VEL_WARM", "DRHO_DIV_BEGIN", "MAKE_K_DIV", "VEL_ITER", "DRHO_DIV_ITER", "END_DIV",
          "WARM_PRESS_SCALE", "WARM_PRESS_VEL", "DRHO_PRESS_BEGIN", "MAKE_K_PRESS", "DRHO_PRESS_ITER", "END_PRESS", "CLEAR",
          "TENSION_NORMAL", "TENSION_ACCEL", "VISC_GUESS", "VISC_MINV", "VISC_RESIDUAL", "VISC_AD", "VISC_UPDATE", "VISC_DIR",
          "VISC_END", "VORTICITY", "VORTICITY_INT", "UPDATE_VEL", "UPDATE_POS", "DENSITY_COFF_WARM", "VISC_MINV_RESIDUAL", "VISC_MINV_RESIDUAL_VORT",
          "END_DIV_BEGIN_NONPRESSURE", "UPDATE_VEL_WARM_SCALE", "END_PRESS_UPDATE_POS"]
PHASE_ID = {n: i for i, n in enumerate(PHASES)}
ACC_STRIPES = 64   # csrc/device.h
ACC_ERR, ACC_DAD, ACC_DNEW, ACC_DZERO = 0, 1, 2, 3
ACC_STRIPES = 64   # csrc/device.h: a reduction slot is 64 striped partial sums

_lib = None


def load():
    """Load libwcsph_hip.so.  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "wcsph_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    lib.wcsph_create.argtypes = [C.POINTER(Params), C.POINTER(vp)]
    lib.wcsph_last_error.argtypes = [vp]
    lib.wcsph_last_error.restype = C.c_char_p
    lib.wcsph_set_stream.argtypes = [vp, vp]
    lib.wcsph_device_count.argtypes = []
    lib.wcsph_setup.argtypes = [vp, vp, vp, vp]
    lib.wcsph_get_block_size.argtypes = [vp, C.POINTER(C.c_int32 * 3)]
    lib.wcsph_field_to_host.argtypes = [vp, C.c_int, vp, C.c_size_t]
    lib.wcsph_field_from_host.argtypes = [vp, C.c_int, vp, C.c_size_t]
    lib.wcsph_get_scalar.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
    lib.wcsph_set_scalar.argtypes = [vp, C.c_int, C.c_float]
    lib.wcsph_get_iters.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.wcsph_set_iters.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32]
    lib.wcsph_neighbors_to_host.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, vp, vp]
    lib.wcsph_compact_lists_to_host.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, vp, vp]
    lib.wcsph_list_stats.argtypes = [vp, vp]
    lib.wcsph_slab_set_halo.argtypes = [vp, vp, C.c_int32, vp, C.c_int32]
    lib.wcsph_slab_step.argtypes = [vp, C.POINTER(SlabCallbacks), C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32 * 2)]
    lib.wcsph_time_kernel.argtypes = [vp, C.c_char_p, C.c_int, C.POINTER(C.c_float)]
    lib.wcsph_state_classify.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, C.c_int32, vp]
    lib.wcsph_slab_set_tail_classify.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, C.c_int32, vp]
    lib.wcsph_slab_tail_classify_result.argtypes = [vp, C.POINTER(C.c_int32 * 5)]
    lib.wcsph_state_gather_rows.argtypes = [vp, vp, C.c_int32, vp]
    i5 = C.POINTER(C.c_int32 * 5)
    lib.wcsph_slab_migrate_begin.argtypes = [vp, vp, C.c_int32, i5, C.POINTER(vp), C.POINTER(vp)]
    lib.wcsph_slab_migrate_end.argtypes = [vp, vp, C.c_int32, C.c_int32, vp, C.c_int32, C.c_int32, i5]
    lib.wcsph_slab_halo_lists.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int32), C.POINTER(vp), C.POINTER(C.c_int32)]
    lib.wcsph_slab_tail_neighbour_counts.argtypes = [vp, i5, i5]
    lib.wcsph_slab_migrate.argtypes = [vp, vp, C.c_int32, i5, i5, i5, i5]
    lib.wcsph_slab_migrate_set_seam.argtypes = [vp, C.c_float, C.c_float, C.c_int32]
    lib.wcsph_state_apply.argtypes = [vp, C.c_int32, vp, vp, C.c_int32, vp, vp, C.c_int32, vp, C.c_int32]
    lib.wcsph_comm_unique_id.argtypes = [vp]
    lib.wcsph_comm_last_error.argtypes = []
    lib.wcsph_comm_last_error.restype = C.c_char_p
    lib.wcsph_comm_init.argtypes = [vp, vp, C.c_int32, C.c_int32]
    lib.wcsph_comm_destroy.argtypes = [vp]
    lib.wcsph_comm_set_neighbours.argtypes = [vp, C.c_int32, C.c_int32]
    lib.wcsph_comm_set_mode.argtypes = [vp, C.c_int32]
    lib.wcsph_comm_set_halo_split.argtypes = [vp, C.c_int32, C.c_int32]
    lib.wcsph_halo_exchange.argtypes = [vp, C.c_int32, vp, C.c_int32, vp, C.c_int32]
    lib.wcsph_comm_allreduce.argtypes = [vp, vp, C.c_int64, C.c_int32]
    lib.wcsph_comm_stats.argtypes = [vp, C.POINTER(C.c_int64 * 2)]
    lib.wcsph_comm_detach.argtypes = [vp, C.POINTER(vp)]
    lib.wcsph_comm_attach.argtypes = [vp, vp]
    lib.wcsph_field_gather_rows.argtypes = [vp, C.c_int, vp, C.c_int32, vp]
    lib.wcsph_field_scatter_rows.argtypes = [vp, C.c_int, vp, C.c_int32, vp]
    lib.wcsph_set_owned.argtypes = [vp, C.c_int32]
    lib.wcsph_set_cell_window.argtypes = [vp, C.c_int32, C.c_int32]
    lib.wcsph_load_state.argtypes = [vp, C.c_int32, C.c_int32, vp]
    lib.wcsph_store_state.argtypes = [vp, vp]
    lib.wcsph_get_acc.argtypes = [vp, C.c_int, C.POINTER(C.c_double)]
    lib.wcsph_set_acc.argtypes = [vp, C.c_int, C.c_double]
    lib.wcsph_phase.argtypes = [vp, C.c_int]
    lib.wcsph_surface_anisotropic_kernel.argtypes = [vp, C.c_float]
    lib.wcsph_dfsph_step_fixed.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32]
    lib.wcsph_mc_tables.argtypes = [vp, vp]
    lib.wcsph_set_acc_buffer.argtypes = [vp, vp]
    lib.wcsph_profile_begin.argtypes = [vp, C.c_int32]
    lib.wcsph_profile_end.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    lib.wcsph_profile_laps_begin.argtypes = [vp, C.c_int32]
    lib.wcsph_profile_laps_end.argtypes = [vp, C.c_char_p, C.c_size_t]
    lib.wcsph_load_state_split.argtypes = [vp, C.c_int32, vp, C.c_int32, vp]
    lib.wcsph_set_part.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32]
    lib.wcsph_canvas_draw.argtypes = [vp, vp, C.c_int32, C.c_int32, vp, vp, vp]
    lib.wcsph_mc_setup.argtypes = [vp, C.c_double, C.c_double, vp, vp, vp]
    lib.wcsph_mc_value_to_host.argtypes = [vp, vp, C.c_int64]
    lib.wcsph_mc_marching_cube.argtypes = [vp, C.c_float, vp, C.c_int32, C.POINTER(C.c_int32)]
    lib.wcsph_set_owned_columns.argtypes = [vp, C.c_int32, C.c_int32]
    lib.wcsph_set_global_liquid.argtypes = [vp, C.c_int64]
    lib.wcsph_grid_update_begin.argtypes = [vp]
    lib.wcsph_grid_update_mid.argtypes = [vp]
    lib.wcsph_grid_update_end.argtypes = [vp]
    lib.wcsph_bucket_hist.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_int64)]
    lib.wcsph_bucket_hist_pack.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int64)]
    lib.wcsph_bucket_hist_unpack.argtypes = [vp]
    lib.wcsph_slab_classify.argtypes = [vp, vp, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, C.c_int32, vp]
    lib.wcsph_halo_pack.argtypes = [vp, C.c_int32, vp, vp, C.c_int32, vp]
    lib.wcsph_halo_unpack.argtypes = [vp, C.c_int32, vp, vp, C.c_int32, vp]
    lib.wcsph_hashgrid_count_to_host.argtypes = [vp, vp, C.c_int64]
    lib.wcsph_hashgrid_rows_to_host.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, vp]
    lib.wcsph_field_rows_to_host.argtypes = [vp, C.c_int, C.c_int32, C.c_int32, vp]
    lib.wcsph_field_rows_from_host.argtypes = [vp, C.c_int, C.c_int32, C.c_int32, vp]
    lib.wcsph_sample_surface.argtypes = [vp, C.c_int32, C.c_double, C.c_uint32, C.c_int32, vp, C.c_int32, C.POINTER(C.c_int32)]
    for name in CTX_ONLY:
        getattr(lib, name).argtypes = [vp]
    for name in ALL_SYMBOLS:
        if name != "wcsph_last_error":
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


class WcsphError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("wcsph error %d (%s): %s" % (code, ERRORS.get(code, "?"), msg))
        self.code = code


def check(ctx, rc):
    if rc != 0:
        lib = load()
        msg = lib.wcsph_last_error(ctx)
        raise WcsphError(rc, msg.decode() if msg else "")
    return rc
