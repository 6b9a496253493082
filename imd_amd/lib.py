"""ctypes binding of libimdgpu.so (C ABI: include/imdgpu.h).

There is deliberately no CPU fallback: if the HIP library is missing or a call
fails, an exception is raised (the product path never routes through oracle/).
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
# IMDGPU_LIB: experiments only (tools/variant_bench.sh builds kernel variants outside the tree)
LIB_PATH = os.environ.get("IMDGPU_LIB") or os.path.join(_CSRC, "libimdgpu.so")

SYMBOLS = [
    "imdgpu_create", "imdgpu_destroy", "imdgpu_last_error", "imdgpu_set_stream", "imdgpu_synchronize", "imdgpu_set_option", "imdgpu_get_info",
    "imdgpu_set_geometry", "imdgpu_set_params", "imdgpu_upload_tables", "imdgpu_set_restrictions",
    "imdgpu_upload_cells", "imdgpu_upload_atoms", "imdgpu_update_positions", "imdgpu_num_atoms", "imdgpu_download_atoms",
    "imdgpu_fix_cells", "imdgpu_fill_ghosts", "imdgpu_make_nblist", "imdgpu_check_nblist",
    "imdgpu_calc_forces", "imdgpu_move_atoms_nve", "imdgpu_do_forces", "imdgpu_do_forces_eam2", "imdgpu_do_embedding_energy", "imdgpu_md_step", "imdgpu_md_step_begin", "imdgpu_md_step_end", "imdgpu_get_sums",
    "imdgpu_export_nblist", "imdgpu_update_momenta", "imdgpu_nblist_stats",
    "imdgpu_profile_enable", "imdgpu_profile_reset", "imdgpu_profile_get",
    "imdgpu_ttm_init", "imdgpu_ttm_dims", "imdgpu_ttm_download", "imdgpu_ttm_upload", "imdgpu_ttm_update_fd", "imdgpu_ttm_adv", "imdgpu_ttm_commflux", "imdgpu_ttm_laser_rescale", "imdgpu_ttm_set_laser", "imdgpu_ttm_get_eabs",
    "imdgpu_ttm_cell_activation", "imdgpu_ttm_fillmesh", "imdgpu_ttm_fill_ghost_layers", "imdgpu_ttm_diff",
    "imdgpu_ttm_cfl_maxdt", "imdgpu_calc_ttm", "imdgpu_move_atoms_ttm", "imdgpu_ttm_eos_te_from_r_ee", "imdgpu_ttm_eos_lookup", "imdgpu_ttm_set_eos_triangulation", "imdgpu_eos_triangulation",
    "imdgpu_get_cell_tables", "imdgpu_upload_ghosts", "imdgpu_update_ghosts", "imdgpu_device_count", "imdgpu_autotile", "imdgpu_set_ghost_layout", "imdgpu_halo_gather", "imdgpu_halo_gather2", "imdgpu_halo_scatter",
    "imdgpu_migrate_scan", "imdgpu_migrate_gather", "imdgpu_migrate_apply", "imdgpu_forces_phase", "imdgpu_integrate", "imdgpu_device_identity",
    "imdgpu_get_invalid", "imdgpu_invalidate", "imdgpu_ttm_local_block", "imdgpu_ttm_apply_blocks",
    "imdgpu_ttm_lattice_step", "imdgpu_ttm_mailbox_bytes", "imdgpu_ttm_mailbox_attach", "imdgpu_ttm_local_step",
    "imdgpu_host_alloc", "imdgpu_host_free",
    "imdgpu_ttm_fdtd_init", "imdgpu_ttm_fdtd_set_dl", "imdgpu_ttm_fdtd_substeps", "imdgpu_ttm_fdtd_fields",
    "imdgpu_ttm_fdtd_get", "imdgpu_ttm_fdtd_set_time", "imdgpu_ttm_fdtd_set_dl_tables", "imdgpu_ttm_fdtd_dl_lookup",
    "imdgpu_ttm_fdtd_fitdl",
    "imdgpu_dist_unique_id", "imdgpu_dist_init_rccl", "imdgpu_dist_init_callback", "imdgpu_dist_set_plan",
    "imdgpu_dist_halo", "imdgpu_dist_set_ttm_blocks", "imdgpu_dist_md_step", "imdgpu_dist_allreduce_sum",
    "imdgpu_dist_rebuild", "imdgpu_dist_rebuild_count", "imdgpu_dist_step", "imdgpu_dist_calc_forces", "imdgpu_dist_plan_geometry",
    "imdgpu_ttm1d_init", "imdgpu_ttm1d_calc", "imdgpu_ttm1d_stage", "imdgpu_ttm1d_move_atoms", "imdgpu_ttm1d_download", "imdgpu_ttm1d_upload", "imdgpu_ttm1d_tmm_init", "imdgpu_ttm1d_tmm_get", "imdgpu_ttm1d_tmm_set_time", "imdgpu_ttm1d_set_epress", "imdgpu_ttm1d_epress_get",
    "imdgpu_ttm_remote_init", "imdgpu_ttm_remote_ipc_handle", "imdgpu_ttm_remote_open", "imdgpu_ttm_remote_attach", "imdgpu_dist_ttm_remote_setup", "imdgpu_ttm_remote_detach", "imdgpu_ttm_remote_ping",
    "imdgpu_local_group_create", "imdgpu_local_group_destroy", "imdgpu_local_group_barrier", "imdgpu_local_group_shared", "imdgpu_dist_init_local",
]


class ImdGpuError(RuntimeError):
    pass


class PotTableC(C.Structure):
    """imdgpu_pot_table == reference pot_table_t (types.h:537-549)."""
    _fields_ = [("begin", C.POINTER(C.c_double)), ("end", C.POINTER(C.c_double)),
                ("step", C.POINTER(C.c_double)), ("invstep", C.POINTER(C.c_double)),
                ("len", C.POINTER(C.c_int)), ("ncols", C.c_int), ("maxsteps", C.c_int),
                ("table", C.POINTER(C.c_double))]


class TtmParamsC(C.Structure):
    """imdgpu_ttm_params"""
    _fields_ = [("fd_ext", C.c_int * 3), ("fd_min_atoms", C.c_int), ("fd_n_timesteps", C.c_int),
                ("atomic_weight", C.c_double), ("atomic_charge", C.c_double), ("omega_laser", C.c_double),
                ("cve_nx", C.c_int), ("cve_ny", C.c_int), ("cve_xmin", C.c_double), ("cve_xmax", C.c_double),
                ("cve_ymin", C.c_double), ("cve_ymax", C.c_double), ("cve_z", C.POINTER(C.c_double)),
                ("ee_nx", C.c_int), ("ee_ny", C.c_int), ("ee_xmin", C.c_double), ("ee_xmax", C.c_double),
                ("ee_ymin", C.c_double), ("ee_ymax", C.c_double), ("ee_z", C.POINTER(C.c_double))]


class Ttm1dParamsC(C.Structure):
    """imdgpu_ttm1d_params"""
    _fields_ = [("nx", C.c_int), ("fd_min_atoms", C.c_int), ("fd_n_timesteps", C.c_int), ("vlatdim", C.c_int), ("vlatbuffer", C.c_int),
                ("ref_ranks", C.c_int), ("adv", C.c_int), ("vlatdens", C.c_double), ("atomic_weight", C.c_double), ("atomic_charge", C.c_double),
                ("cve_nx", C.c_int), ("cve_ny", C.c_int), ("cve_xmin", C.c_double), ("cve_xmax", C.c_double),
                ("cve_ymin", C.c_double), ("cve_ymax", C.c_double), ("cve_z", C.POINTER(C.c_double)),
                ("ee_nx", C.c_int), ("ee_ny", C.c_int), ("ee_xmin", C.c_double), ("ee_xmax", C.c_double),
                ("ee_ymin", C.c_double), ("ee_ymax", C.c_double), ("ee_z", C.POINTER(C.c_double))]


TTM1D_INT = ("natoms", "natoms_old")
TTM1D_DBL = ("temp", "temp2", "U", "U2", "xi", "md_temp", "source", "dens", "fd_k", "fd_g", "Z", "Ce", "ne")
TTM1D_GINT = ("natomsglobal", "fluxfromleft", "fluxfromright")
TTM1D_REST = ("vcomglobal", "xiarr_global", "vlat_temp", "vlat_md_temp", "vlat_fd_k", "vlat_fd_g", "vlat_Ce", "scalars")


class Ttm1dFieldsC(C.Structure):
    """imdgpu_ttm1d_fields"""
    _fields_ = [(k, C.POINTER(C.c_int)) for k in TTM1D_INT] + [(k, C.POINTER(C.c_double)) for k in TTM1D_DBL] + \
               [(k, C.POINTER(C.c_int)) for k in TTM1D_GINT] + [(k, C.POINTER(C.c_double)) for k in TTM1D_REST]


TTM_FIELDS = ("natoms", "natoms_old", "temp", "temp2", "xi", "md_temp", "source", "vcom", "dens", "fd_k", "fd_g",
              "Z", "Ce", "ne", "U", "U2", "flux")
TTM_INT_FIELDS = ("natoms", "natoms_old", "flux")


class LaserParamsC(C.Structure):
    _fields_ = [("dir", C.c_int * 3), ("offset", C.c_double), ("mu", C.c_double), ("p_peak", C.c_double),
                ("t_0", C.c_double), ("sigma_t_squared", C.c_double)]


class FdtdParamsC(C.Structure):
    """imdgpu_fdtd_params"""
    _fields_ = [("I0", C.c_double), ("lam", C.c_double), ("sigma_t", C.c_double), ("t_0", C.c_double), ("srcx", C.c_double),
                ("bw", C.c_int), ("Sc", C.c_double), ("DL", C.c_double * 5)]


class _TmmPhaseC(C.Structure):
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("xmin", C.c_double), ("xmax", C.c_double), ("ymin", C.c_double), ("ymax", C.c_double),
                ("z", C.POINTER(C.c_double))]


class TmmParamsC(C.Structure):
    """imdgpu_tmm_params"""
    _fields_ = [(k, C.c_double) for k in ("I0", "lam", "laser_sigma_t", "laser_t_0", "laser_sigma_t1", "laser_t_1", "tmm_threshold",
                                          "tmm_laser_threshold")] + \
               [("eps_bb_rows", C.c_int), ("eps_bb", C.POINTER(C.c_double)), ("K12_rows", C.c_int), ("K12", C.POINTER(C.c_double)),
                ("phase", _TmmPhaseC)]


class TricubTableC(C.Structure):
    """imdgpu_tricub_table"""
    _fields_ = [("is_", C.c_int), ("js", C.c_int), ("ks", C.c_int), ("range", C.c_double * 6)] + [(n, C.POINTER(C.c_double)) for n in "xyzf"]


class TtmFieldsC(C.Structure):
    """imdgpu_ttm_fields"""
    _fields_ = [(n, C.POINTER(C.c_int) if n in TTM_INT_FIELDS else C.POINTER(C.c_double)) for n in TTM_FIELDS]


def build(force=False):
    """Compile libimdgpu.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "imdgpu.h"))
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _CSRC, "libimdgpu.so"])
    return LIB_PATH


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImdGpuError("libimdgpu.so not built (%s): run __graft_entry__.build() / make -C imd_amd/csrc" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
    L.imdgpu_create.argtypes = [C.POINTER(vp), C.c_int]
    L.imdgpu_destroy.argtypes = [vp]
    L.imdgpu_destroy.restype = None
    L.imdgpu_last_error.argtypes = [vp]
    L.imdgpu_last_error.restype = C.c_char_p
    L.imdgpu_set_stream.argtypes = [vp, vp]
    L.imdgpu_synchronize.argtypes = [vp]
    L.imdgpu_set_option.argtypes = [vp, C.c_char_p, C.c_int]
    L.imdgpu_get_info.argtypes = [vp, C.c_char_p, C.POINTER(C.c_longlong)]
    L.imdgpu_set_geometry.argtypes = [vp, dp, ip, ip, ip, ip, ip]
    L.imdgpu_set_params.argtypes = [vp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int]
    L.imdgpu_upload_tables.argtypes = [vp, C.POINTER(PotTableC), C.POINTER(PotTableC), C.POINTER(PotTableC)]
    L.imdgpu_set_restrictions.argtypes = [vp, dp]
    L.imdgpu_upload_cells.argtypes = [vp, ip, C.c_int, dp, dp, dp, ip, ip]
    L.imdgpu_upload_atoms.argtypes = [vp, C.c_int, dp, dp, dp, ip, ip]
    L.imdgpu_num_atoms.argtypes = [vp, ip, ip]
    L.imdgpu_update_positions.argtypes = [vp, C.c_int, dp]
    L.imdgpu_update_momenta.argtypes = [vp, C.c_int, dp]
    L.imdgpu_upload_ghosts.argtypes = [vp, ip, C.c_int, dp, ip]
    L.imdgpu_update_ghosts.argtypes = [vp, C.c_int, dp, dp]
    L.imdgpu_device_count.argtypes = [ip]
    L.imdgpu_download_atoms.argtypes = [vp, ip, dp, dp, dp, dp, dp, dp, ip, ip, ip, dp]
    for f in ("imdgpu_fix_cells", "imdgpu_fill_ghosts", "imdgpu_make_nblist"):
        getattr(L, f).argtypes = [vp]
    L.imdgpu_check_nblist.argtypes = [vp, ip]
    L.imdgpu_calc_forces.argtypes = [vp, C.c_int, dp, ip]
    L.imdgpu_move_atoms_nve.argtypes = [vp, dp]
    L.imdgpu_md_step.argtypes = [vp, C.c_int]
    L.imdgpu_get_sums.argtypes = [vp, dp, dp, ip]
    L.imdgpu_export_nblist.argtypes = [vp, ip, ip, ip, ip]
    L.imdgpu_nblist_stats.argtypes = [vp, C.POINTER(C.c_longlong), ip]
    L.imdgpu_profile_enable.argtypes = [vp, C.c_int]
    L.imdgpu_profile_reset.argtypes = [vp]
    L.imdgpu_profile_get.argtypes = [vp, C.c_char_p, C.POINTER(C.c_longlong), dp]
    L.imdgpu_ttm_init.argtypes = [vp, C.POINTER(TtmParamsC)]
    L.imdgpu_ttm_dims.argtypes = [vp, ip, dp]
    L.imdgpu_ttm_download.argtypes = [vp, C.POINTER(TtmFieldsC)]
    L.imdgpu_ttm_upload.argtypes = [vp, C.POINTER(TtmFieldsC)]
    L.imdgpu_ttm_update_fd.argtypes = [vp, C.c_int]
    L.imdgpu_ttm_adv.argtypes = [vp, C.c_double, C.c_int]
    L.imdgpu_ttm_commflux.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(C.c_int)]
    L.imdgpu_ttm_laser_rescale.argtypes = [vp, C.POINTER(LaserParamsC), C.c_int]
    L.imdgpu_ttm_set_laser.argtypes = [vp, C.POINTER(LaserParamsC)]
    L.imdgpu_ttm_get_eabs.argtypes = [vp, C.POINTER(C.c_double)]
    L.imdgpu_ttm_cell_activation.argtypes = [vp]
    L.imdgpu_ttm_fillmesh.argtypes = [vp, C.c_int]
    L.imdgpu_ttm_fill_ghost_layers.argtypes = [vp]
    L.imdgpu_ttm_diff.argtypes = [vp, C.c_double]
    L.imdgpu_ttm_cfl_maxdt.argtypes = [vp, C.c_int, dp]
    L.imdgpu_calc_ttm.argtypes = [vp, C.c_int]
    L.imdgpu_ttm_eos_te_from_r_ee.argtypes = [vp, C.c_int, dp, dp, dp]
    L.imdgpu_ttm_eos_lookup.argtypes = [vp, C.c_int, C.c_int, dp, dp, dp]
    L.imdgpu_ttm_set_eos_triangulation.argtypes = [vp, C.c_int, C.c_char_p]
    L.imdgpu_eos_triangulation.argtypes = [C.c_int, C.c_int, C.c_void_p]
    L.imdgpu_move_atoms_ttm.argtypes = [vp, dp]
    L.imdgpu_ttm_fdtd_init.argtypes = [vp, C.POINTER(FdtdParamsC)]
    L.imdgpu_ttm_fdtd_set_dl.argtypes = [vp, dp]
    L.imdgpu_ttm_fdtd_substeps.argtypes = [vp, C.c_double, C.c_int, C.c_double, C.c_double]
    L.imdgpu_ttm_fdtd_fields.argtypes = [vp, C.c_int, dp]
    L.imdgpu_ttm_fdtd_get.argtypes = [vp, dp, dp, ip, dp]
    L.imdgpu_ttm_fdtd_set_time.argtypes = [vp, C.c_double]
    L.imdgpu_ttm_fdtd_set_dl_tables.argtypes = [vp, C.POINTER(TricubTableC)]
    L.imdgpu_ttm_fdtd_dl_lookup.argtypes = [vp, C.c_int, C.c_int, dp, dp]
    L.imdgpu_ttm_fdtd_fitdl.argtypes = [vp, C.c_int, dp, dp, dp, dp, dp]
    L.imdgpu_dist_unique_id.argtypes = [C.c_char_p, C.c_int]
    L.imdgpu_dist_init_rccl.argtypes = [vp, C.c_int, C.c_int, C.c_char_p, C.c_int]
    L.imdgpu_dist_init_callback.argtypes = [vp, C.c_int, C.c_int, vp, vp]
    L.imdgpu_dist_set_plan.argtypes = [vp, C.c_int, ip, ip, dp, vp, C.c_int, ip, ip]
    L.imdgpu_dist_halo.argtypes = [vp, C.c_int]
    L.imdgpu_dist_set_ttm_blocks.argtypes = [vp, ip]
    L.imdgpu_dist_md_step.argtypes = [vp, C.c_int, ip]
    L.imdgpu_dist_allreduce_sum.argtypes = [vp, dp, C.c_int]
    L.imdgpu_ttm1d_init.argtypes = [vp, C.POINTER(Ttm1dParamsC)]
    L.imdgpu_ttm1d_calc.argtypes = [vp, C.c_int]
    L.imdgpu_ttm1d_stage.argtypes = [vp, C.c_int, C.c_int]
    L.imdgpu_ttm1d_move_atoms.argtypes = [vp, dp]
    L.imdgpu_ttm1d_download.argtypes = [vp, C.POINTER(Ttm1dFieldsC)]
    L.imdgpu_ttm1d_upload.argtypes = [vp, C.POINTER(Ttm1dFieldsC)]
    L.imdgpu_ttm1d_tmm_init.argtypes = [vp, C.POINTER(TmmParamsC)]
    L.imdgpu_ttm1d_tmm_get.argtypes = [vp, dp, dp, dp, dp, ip, ip]
    L.imdgpu_ttm1d_tmm_set_time.argtypes = [vp, C.c_double]
    L.imdgpu_ttm1d_set_epress.argtypes = [vp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, dp]
    L.imdgpu_ttm1d_epress_get.argtypes = [vp, dp, dp]
    L.imdgpu_ttm_remote_init.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.imdgpu_ttm_remote_ipc_handle.argtypes = [vp, vp, ip]
    L.imdgpu_ttm_remote_open.argtypes = [vp, vp, C.POINTER(vp)]
    L.imdgpu_ttm_remote_attach.argtypes = [vp, C.POINTER(vp)]
    L.imdgpu_dist_ttm_remote_setup.argtypes = [vp]
    L.imdgpu_ttm_remote_detach.argtypes = [vp]
    L.imdgpu_ttm_remote_ping.argtypes = [vp, C.c_int, C.c_uint, ip]
    L.imdgpu_do_forces.argtypes = [vp, C.c_int, dp, ip, C.c_int, dp, ip, dp, C.c_int, dp, dp, dp, dp, dp, dp, ip, ip, dp, dp, ip]
    L.imdgpu_do_forces_eam2.argtypes = [vp, C.c_int, dp, ip, dp, C.c_int, dp, ip, dp, dp, C.c_int, dp, dp, dp, ip]
    L.imdgpu_do_embedding_energy.argtypes = [vp, C.c_int, dp, ip, dp, dp, dp]
    L.imdgpu_md_step_begin.argtypes = [vp, C.c_int]
    L.imdgpu_md_step_end.argtypes = [vp, C.c_int]
    L.imdgpu_dist_rebuild.argtypes = [vp]
    L.imdgpu_dist_rebuild_count.argtypes = [vp, ip]
    L.imdgpu_dist_step.argtypes = [vp, C.c_int]
    L.imdgpu_dist_calc_forces.argtypes = [vp]
    L.imdgpu_dist_plan_geometry.argtypes = [ip, ip, ip, ip, dp, ip, dp, ip, ip, ip, ip]
    L.imdgpu_local_group_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.imdgpu_local_group_destroy.argtypes = [vp]
    L.imdgpu_local_group_destroy.restype = None
    L.imdgpu_local_group_barrier.argtypes = [vp]
    L.imdgpu_local_group_shared.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(vp)]
    L.imdgpu_dist_init_local.argtypes = [vp, vp, C.c_int]
    L.imdgpu_get_cell_tables.argtypes = [vp, ip, ip]
    L.imdgpu_autotile.argtypes = [vp, ip]
    L.imdgpu_set_ghost_layout.argtypes = [vp, ip, ip, C.c_int]
    L.imdgpu_halo_gather.argtypes = [vp, C.c_int, vp, C.c_int, dp, vp]
    L.imdgpu_halo_gather2.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, dp, dp, vp]
    L.imdgpu_halo_scatter.argtypes = [vp, C.c_int, vp, C.c_int, vp]
    L.imdgpu_migrate_scan.argtypes = [vp, C.c_int, ip, ip]
    L.imdgpu_migrate_gather.argtypes = [vp, C.c_int, vp]
    L.imdgpu_migrate_apply.argtypes = [vp, vp, C.c_int, vp, C.c_int]
    L.imdgpu_forces_phase.argtypes = [vp, C.c_int]
    L.imdgpu_integrate.argtypes = [vp]
    L.imdgpu_device_identity.argtypes = [vp, C.c_char_p, C.c_int]
    L.imdgpu_get_invalid.argtypes = [vp, ip]
    L.imdgpu_invalidate.argtypes = [vp]
    L.imdgpu_ttm_local_block.argtypes = [vp, C.c_int, vp, ip]
    L.imdgpu_ttm_apply_blocks.argtypes = [vp, C.c_int, vp, C.c_int, vp]
    L.imdgpu_ttm_lattice_step.argtypes = [vp, C.c_int]
    L.imdgpu_ttm_mailbox_bytes.argtypes = [vp, C.POINTER(C.c_size_t)]
    L.imdgpu_ttm_mailbox_attach.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, ip]
    L.imdgpu_ttm_local_step.argtypes = [vp, C.c_int]
    for s in SYMBOLS:
        if s not in ("imdgpu_destroy", "imdgpu_last_error", "imdgpu_local_group_destroy"):
            getattr(L, s).restype = C.c_int
    _lib = L
    return L
