"""Host-side mirror of the reference's entry points for the force path.

``GpuSystem`` exposes the reference's function names (calc_forces, make_nblist,
check_nblist, fix_cells, move_atoms_nve; imd_forces_nbl.c, imd_integrate.c)
with the reference's argument meaning, on top of the C ABI of libimdgpu.so.
Errors raise ``ImdGpuError`` where the reference would call error().
"""
import ctypes as C

import numpy as np

from . import lib as _lib
from .lib import (TTM1D_DBL, TTM1D_GINT, TTM1D_INT, TTM_FIELDS, TTM_INT_FIELDS, FdtdParamsC, ImdGpuError, LaserParamsC, PotTableC, TmmParamsC, TricubTableC,
                  Ttm1dFieldsC, Ttm1dParamsC, TtmFieldsC, TtmParamsC)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def global_cell_dim(box, cellsz, cd=(1, 1, 1)):
    """Cells per direction: floor(1/sqrt(cellsz/height)), rounded down to a
    multiple of cd = cpu_dim[*fd_ext] (init_cells, imd_geom_3d.c:141-188)."""
    b = np.asarray(box, dtype=np.float64).reshape(3, 3)
    t = np.linalg.inv(b).T              # rows tbox_x, tbox_y, tbox_z
    out = []
    for d in range(3):
        height = 1.0 / float(t[d] @ t[d])
        g = int(1.0 / np.sqrt(cellsz / height))
        if g % cd[d]:
            g = (g // cd[d]) * cd[d]
        out.append(g)
    return tuple(out)


class GpuSystem:
    def __init__(self, box, pbc_dirs, cell_dim, ntypes, cellsz, nbl_margin, timestep, device=0,
                 global_cell_dim=None, cpu_dim=(1, 1, 1), my_coord=(0, 0, 0), count_ttm_neighbours=False,
                 external_halo=False, ttm_decomposed=False):
        self.L = _lib.load()
        h = C.c_void_p()
        if self.L.imdgpu_create(C.byref(h), int(device)):
            msg = self.L.imdgpu_last_error(h).decode() if h else "imdgpu_create failed"
            raise ImdGpuError(msg)
        self.h = h
        self.box = np.ascontiguousarray(box, dtype=np.float64).reshape(9)
        self.cell_dim = np.asarray(cell_dim, dtype=np.int32)
        self.pbc = np.asarray(pbc_dirs, dtype=np.int32)
        gd = np.asarray(global_cell_dim if global_cell_dim is not None
                        else (self.cell_dim - 2) * np.asarray(cpu_dim), dtype=np.int32)
        self.nallcells = int(np.prod(self.cell_dim))
        self.ntypes = int(ntypes)
        self._keep = []
        self.ttm_decomposed = bool(ttm_decomposed)
        if ttm_decomposed:           # several ranks: one TTM lattice block per rank (faces through the mailbox)
            self._ck(self.L.imdgpu_set_option(self.h, b"ttm_decomposed", 1))
        if external_halo:            # a host layer (imd_amd.parallel) fills the ghost cells even on one rank
            self._ck(self.L.imdgpu_set_option(self.h, b"external_halo", 1))
        self._ck(self.L.imdgpu_set_geometry(self.h, _dp(self.box), _ip(self.pbc), _ip(self.cell_dim), _ip(gd),
                                            _ip(np.asarray(cpu_dim, dtype=np.int32)),
                                            _ip(np.asarray(my_coord, dtype=np.int32))))
        self._ck(self.L.imdgpu_set_params(self.h, int(ntypes), float(cellsz), float(nbl_margin), float(timestep),
                                          1 if count_ttm_neighbours else 0))

    # -- plumbing --------------------------------------------------------------
    def _ck(self, rc):
        if rc:
            raise ImdGpuError(self.L.imdgpu_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.imdgpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _tab(self, t):
        if t is None:
            return None
        d = t.as_dict() if hasattr(t, "as_dict") else t
        arrs = {k: np.ascontiguousarray(d[k], dtype=np.float64) for k in ("begin", "end", "step", "invstep", "table")}
        ln = np.ascontiguousarray(d["len"], dtype=np.int32)
        self._keep.append((arrs, ln))
        return PotTableC(_dp(arrs["begin"]), _dp(arrs["end"]), _dp(arrs["step"]), _dp(arrs["invstep"]), _ip(ln),
                         int(d["ncols"]), int(d["maxsteps"]), _dp(arrs["table"]))

    def upload_tables(self, pair_pot=None, rho_h_tab=None, embed_pot=None):
        ts = [self._tab(t) for t in (pair_pot, rho_h_tab, embed_pot)]
        ptr = [C.byref(t) if t is not None else None for t in ts]
        self._ck(self.L.imdgpu_upload_tables(self.h, *ptr))

    def set_stream(self, hip_stream):
        self._ck(self.L.imdgpu_set_stream(self.h, C.c_void_p(int(hip_stream))))

    def set_restrictions(self, r):
        """`restrictions` of the parameter file (globals.h; KRAFT *= restrictions in move_atoms_*): [ntypes][3] factors"""
        a = np.ascontiguousarray(r, dtype=np.float64).reshape(-1)
        self._ck(self.L.imdgpu_set_restrictions(self.h, _dp(a)))

    def set_option(self, name, value):
        self._ck(self.L.imdgpu_set_option(self.h, name.encode(), int(value)))

    def get_info(self, name):
        v = C.c_longlong()
        self._ck(self.L.imdgpu_get_info(self.h, name.encode(), C.byref(v)))
        return v.value

    def device_identity(self):
        buf = C.create_string_buffer(256)
        self._ck(self.L.imdgpu_device_identity(self.h, buf, 256))
        return buf.value.decode()

    def synchronize(self):
        self._ck(self.L.imdgpu_synchronize(self.h))

    # -- atoms -------------------------------------------------------------------
    @staticmethod
    def _atoms(ort, impuls, masse, sorte, nummer):
        ort = np.ascontiguousarray(ort, dtype=np.float64).reshape(-1, 3)
        n = len(ort)
        impuls = np.zeros((n, 3)) if impuls is None else np.ascontiguousarray(impuls, dtype=np.float64).reshape(n, 3)
        masse = np.ascontiguousarray(np.broadcast_to(np.asarray(masse, dtype=np.float64), (n,)))
        sorte = np.zeros(n, np.int32) if sorte is None else np.ascontiguousarray(sorte, dtype=np.int32)
        nummer = np.arange(n, dtype=np.int32) if nummer is None else np.ascontiguousarray(nummer, dtype=np.int32)
        return n, ort, impuls, masse, sorte, nummer

    def upload_cells(self, cell_n, ort, impuls=None, masse=1.0, sorte=None, nummer=None):
        """Take the host's cell arrangement as is (cell ordered arrays)."""
        n, ort, impuls, masse, sorte, nummer = self._atoms(ort, impuls, masse, sorte, nummer)
        cn = np.ascontiguousarray(cell_n, dtype=np.int32)
        if len(cn) != self.nallcells:
            raise ImdGpuError("cell_n must have nallcells=%d entries" % self.nallcells)
        self._ck(self.L.imdgpu_upload_cells(self.h, _ip(cn), n, _dp(ort), _dp(impuls), _dp(masse), _ip(sorte),
                                            _ip(nummer)))

    def upload_atoms(self, ort, impuls=None, masse=1.0, sorte=None, nummer=None):
        n, ort, impuls, masse, sorte, nummer = self._atoms(ort, impuls, masse, sorte, nummer)
        self._ck(self.L.imdgpu_upload_atoms(self.h, n, _dp(ort), _dp(impuls), _dp(masse), _ip(sorte), _ip(nummer)))

    def update_positions(self, ort):
        """new positions, cell ordered like download() returns them; the list stays valid"""
        a = np.ascontiguousarray(ort, dtype=np.float64)
        self._ck(self.L.imdgpu_update_positions(self.h, len(a), _dp(a)))

    def update_momenta(self, impuls):
        a = np.ascontiguousarray(impuls, dtype=np.float64)
        self._ck(self.L.imdgpu_update_momenta(self.h, len(a), _dp(a)))

    def num_atoms(self):
        a, b = C.c_int(), C.c_int()
        self._ck(self.L.imdgpu_num_atoms(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def download(self, fields=("ort", "impuls", "kraft", "pot_eng", "eam_rho", "eam_dF", "numneighs", "nummer",
                               "sorte", "nbl_pos", "cell_n")):
        n, _ = self.num_atoms()
        out = {}
        spec = [("cell_n", np.int32, (self.nallcells,)), ("ort", np.float64, (n, 3)), ("impuls", np.float64, (n, 3)),
                ("kraft", np.float64, (n, 3)), ("pot_eng", np.float64, (n,)), ("eam_rho", np.float64, (n,)),
                ("eam_dF", np.float64, (n,)), ("numneighs", np.int32, (n,)), ("nummer", np.int32, (n,)),
                ("sorte", np.int32, (n,)), ("nbl_pos", np.float64, (n, 3))]
        args = []
        for name, dt, shp in spec:
            if name in fields:
                out[name] = np.zeros(shp, dtype=dt)
                args.append(_ip(out[name]) if dt == np.int32 else _dp(out[name]))
            else:
                args.append(None)
        self._ck(self.L.imdgpu_download_atoms(self.h, *args))
        return out

    # -- the reference's entry points ------------------------------------------
    def fix_cells(self):
        self._ck(self.L.imdgpu_fix_cells(self.h))

    def send_cells(self):
        self._ck(self.L.imdgpu_fill_ghosts(self.h))

    def make_nblist(self):
        self._ck(self.L.imdgpu_make_nblist(self.h))

    def check_nblist(self):
        v = C.c_int()
        self._ck(self.L.imdgpu_check_nblist(self.h, C.byref(v)))
        return bool(v.value)

    def calc_forces(self, steps=0):
        """Returns dict(tot_pot_energy, virial, is_short)."""
        s = np.zeros(8)
        sh = C.c_int()
        self._ck(self.L.imdgpu_calc_forces(self.h, int(steps), _dp(s), C.byref(sh)))
        return dict(tot_pot_energy=float(s[0]), virial=float(s[1]), is_short=int(sh.value))

    def move_atoms_nve(self):
        e = C.c_double()
        self._ck(self.L.imdgpu_move_atoms_nve(self.h, C.byref(e)))
        return e.value

    def md_step(self, steps=0):
        self._ck(self.L.imdgpu_md_step(self.h, int(steps)))

    def get_sums(self):
        s = np.zeros(8)
        e = C.c_double()
        c = C.c_int()
        self._ck(self.L.imdgpu_get_sums(self.h, _dp(s), C.byref(e), C.byref(c)))
        return dict(tot_pot_energy=float(s[0]), virial=float(s[1]), tot_kin_energy=e.value, nbl_count=c.value)

    def export_nblist(self):
        """(tl, tb, cl_off) in the reference's half-list format and numbering."""
        n, _ = self.num_atoms()
        ntb = C.c_int()
        self._ck(self.L.imdgpu_export_nblist(self.h, None, None, None, C.byref(ntb)))
        tl = np.zeros(n + 1, np.int32)
        tb = np.zeros(max(ntb.value, 1), np.int32)
        cl_off = np.zeros(self.nallcells, np.int32)
        self._ck(self.L.imdgpu_export_nblist(self.h, _ip(tl), _ip(tb), _ip(cl_off), C.byref(ntb)))
        return tl, tb[: ntb.value], cl_off

    def nblist_stats(self):
        e = C.c_longlong()
        m = C.c_int()
        self._ck(self.L.imdgpu_nblist_stats(self.h, C.byref(e), C.byref(m)))
        return e.value, m.value

    # -- two-temperature model: the reference's entry points (imd_ttm_3D.c) -------
    def init_ttm(self, fd_ext, fd_min_atoms, fd_n_timesteps, atomic_weight, atomic_charge, eos_cve, eos_ee,
                 omega_laser=0.0):
        """eos_cve / eos_ee: dict(x=rho grid, y=T or sqrt(T) grid, z=(nx,ny) values) on a regular grid."""
        P = TtmParamsC()
        P.fd_ext[:] = [int(v) for v in fd_ext]
        P.fd_min_atoms, P.fd_n_timesteps = int(fd_min_atoms), int(fd_n_timesteps)
        P.atomic_weight, P.atomic_charge, P.omega_laser = float(atomic_weight), float(atomic_charge), float(omega_laser)
        zc = np.ascontiguousarray(eos_cve["z"], dtype=np.float64)
        ze = np.ascontiguousarray(eos_ee["z"], dtype=np.float64)
        self._keep.append((zc, ze))
        P.cve_nx, P.cve_ny = zc.shape
        P.cve_xmin, P.cve_xmax = float(eos_cve["x"][0]), float(eos_cve["x"][-1])
        P.cve_ymin, P.cve_ymax = float(eos_cve["y"][0]), float(eos_cve["y"][-1])
        P.cve_z = _dp(zc)
        P.ee_nx, P.ee_ny = ze.shape
        P.ee_xmin, P.ee_xmax = float(eos_ee["x"][0]), float(eos_ee["x"][-1])
        P.ee_ymin, P.ee_ymax = float(eos_ee["y"][0]), float(eos_ee["y"][-1])
        P.ee_z = _dp(ze)
        self._ck(self.L.imdgpu_ttm_init(self.h, C.byref(P)))
        d = np.zeros(3, np.int32)
        h = np.zeros(3)
        self._ck(self.L.imdgpu_ttm_dims(self.h, _ip(d), _dp(h)))
        self.fd_dim, self.fd_h = tuple(int(v) for v in d), h
        self.fd_ext = tuple(int(v) for v in fd_ext)
        self.diff_substeps = int(fd_n_timesteps)

    def ttm_download(self, fields=TTM_FIELDS):
        F = TtmFieldsC()
        out = {}
        for n in fields:
            shp = self.fd_dim + ((3,) if n == "vcom" else (8,) if n == "flux" else ())
            out[n] = np.zeros(shp, dtype=np.int32 if n in TTM_INT_FIELDS else np.float64)
            setattr(F, n, _ip(out[n]) if n in TTM_INT_FIELDS else _dp(out[n]))
        self._ck(self.L.imdgpu_ttm_download(self.h, C.byref(F)))
        return out

    def ttm_upload(self, **fields):
        F = TtmFieldsC()
        keep = []
        for n, v in fields.items():
            a = np.ascontiguousarray(v, dtype=np.int32 if n in TTM_INT_FIELDS else np.float64)
            keep.append(a)
            setattr(F, n, _ip(a) if n in TTM_INT_FIELDS else _dp(a))
        self._ck(self.L.imdgpu_ttm_upload(self.h, C.byref(F)))

    # -- the reference's own 1D TTM (imd_ttm.c, -DTTM1D) -------------------------------------
    def init_ttm1d(self, ttmdimx, fd_min_atoms, fd_n_timesteps, vlatdim, vlatbuffer, vlatdens, atomic_weight, atomic_charge, eos_cve, eos_ee,
                   ref_ranks=1, adv=True):
        """init_ttm of imd_ttm.c (:955-1206): ttmdimx slabs along x, a virtual lattice of vlatdim cells behind the sample"""
        P = Ttm1dParamsC()
        P.nx, P.fd_min_atoms, P.fd_n_timesteps, P.vlatdim, P.vlatbuffer = int(ttmdimx), int(fd_min_atoms), int(fd_n_timesteps), int(vlatdim), int(vlatbuffer)
        P.ref_ranks, P.adv, P.vlatdens, P.atomic_weight, P.atomic_charge = int(ref_ranks), 1 if adv else 0, float(vlatdens), float(atomic_weight), float(atomic_charge)
        zc = np.ascontiguousarray(eos_cve["z"], dtype=np.float64); ze = np.ascontiguousarray(eos_ee["z"], dtype=np.float64)
        self._keep.append((zc, ze))
        P.cve_nx, P.cve_ny = zc.shape
        P.cve_xmin, P.cve_xmax, P.cve_ymin, P.cve_ymax = (float(eos_cve["x"][0]), float(eos_cve["x"][-1]), float(eos_cve["y"][0]), float(eos_cve["y"][-1]))
        P.cve_z = _dp(zc)
        P.ee_nx, P.ee_ny = ze.shape
        P.ee_xmin, P.ee_xmax, P.ee_ymin, P.ee_ymax = (float(eos_ee["x"][0]), float(eos_ee["x"][-1]), float(eos_ee["y"][0]), float(eos_ee["y"][-1]))
        P.ee_z = _dp(ze)
        self._ck(self.L.imdgpu_ttm1d_init(self.h, C.byref(P)))
        self.ttm1d_nx, self.ttm1d_nv = int(ttmdimx), int(vlatdim)

    def init_ttm1d_tmm(self, I0, lam, laser_sigma_t, laser_t_0, eps_bb, K12, phase, laser_sigma_t1=0.0, laser_t_1=0.0, tmm_threshold=20.0,
                       tmm_laser_threshold=1e-5):
        """tmm_init (imd_tmm.c:98-191): the transfer-matrix laser source of the 1D model; eps_bb [rows, 3] / K12 [1500, 2] as
        imd_amd.tmm.read_arr gives them (or paths of alu_eps_bb.dat / K12.dat), phase = dict(x, y, z) like the EOS tables"""
        from . import tmm as tmm_mod
        eb = np.ascontiguousarray(tmm_mod.read_arr(eps_bb, 3) if isinstance(eps_bb, str) else eps_bb, dtype=np.float64)
        k12 = np.ascontiguousarray(tmm_mod.read_arr(K12, 2) if isinstance(K12, str) else K12, dtype=np.float64)
        pz = np.ascontiguousarray(phase["z"], dtype=np.float64)
        P = TmmParamsC()
        P.I0, P.lam, P.laser_sigma_t, P.laser_t_0, P.laser_sigma_t1, P.laser_t_1 = (float(I0), float(lam), float(laser_sigma_t), float(laser_t_0),
                                                                                   float(laser_sigma_t1), float(laser_t_1))
        P.tmm_threshold, P.tmm_laser_threshold = float(tmm_threshold), float(tmm_laser_threshold)
        P.eps_bb_rows, P.eps_bb, P.K12_rows, P.K12 = len(eb), _dp(eb), len(k12), _dp(k12)
        P.phase.nx, P.phase.ny = pz.shape
        P.phase.xmin, P.phase.xmax, P.phase.ymin, P.phase.ymax = (float(phase["x"][0]), float(phase["x"][-1]), float(phase["y"][0]), float(phase["y"][-1]))
        P.phase.z = _dp(pz)
        self._ck(self.L.imdgpu_ttm1d_tmm_init(self.h, C.byref(P)))

    def ttm1d_tmm_state(self):
        """do_tmm's state after the last call: scalars, permittivity, active cells, tmm_Qabs"""
        nx = self.ttm1d_nx
        sc, er, ei, q = np.zeros(8), np.zeros(nx), np.zeros(nx), np.zeros(nx)
        act = np.zeros(nx, np.int32)
        short = C.c_int()
        self._ck(self.L.imdgpu_ttm1d_tmm_get(self.h, _dp(sc), _dp(er), _dp(ei), _dp(q), _ip(act), C.byref(short)))
        return dict(tmm_time=sc[0], laser_active=int(sc[1]), eabs=sc[2], I_t=sc[3], refl=sc[4], trans=sc[5], absorbed=sc[6], ecut=int(sc[7]),
                    eps_re=er, eps_im=ei, qabs=q, active=act, too_short=short.value)

    def ttm1d_set_epress(self, pe):
        """-DELECPRESS: the table of EOS_pe_from_r_te [Pa] as dict(x, y, z) like the EOS tables; None: off"""
        if pe is None:
            self._ck(self.L.imdgpu_ttm1d_set_epress(self.h, 0, 0, 0.0, 0.0, 0.0, 0.0, None)); return
        z = np.ascontiguousarray(pe["z"], dtype=np.float64)
        self._ck(self.L.imdgpu_ttm1d_set_epress(self.h, z.shape[0], z.shape[1], float(pe["x"][0]), float(pe["x"][-1]), float(pe["y"][0]),
                                                float(pe["y"][-1]), _dp(z)))

    def ttm1d_epress(self):
        nx = self.ttm1d_nx
        a, b = np.zeros(nx), np.zeros(nx)
        self._ck(self.L.imdgpu_ttm1d_epress_get(self.h, _dp(a), _dp(b)))
        return a, b

    def ttm1d_tmm_set_time(self, tmm_time):
        self._ck(self.L.imdgpu_ttm1d_tmm_set_time(self.h, float(tmm_time)))

    def ttm1d_calc(self, steps):
        self._ck(self.L.imdgpu_ttm1d_calc(self.h, int(steps)))

    def ttm1d_stage(self, steps, stages):
        self._ck(self.L.imdgpu_ttm1d_stage(self.h, int(steps), int(stages)))

    def ttm1d_move_atoms(self):
        e = C.c_double()
        self._ck(self.L.imdgpu_ttm1d_move_atoms(self.h, C.byref(e)))
        return e.value

    def _ttm1d_shapes(self):
        nx, nv = self.ttm1d_nx, self.ttm1d_nv
        sh = {k: (nx + 2,) for k in TTM1D_INT + TTM1D_DBL}
        sh.update({k: (nx,) for k in TTM1D_GINT + ("xiarr_global",)})
        sh.update(vcomglobal=(3, nx), scalars=(4,))
        sh.update({k: (nv,) for k in ("vlat_temp", "vlat_md_temp", "vlat_fd_k", "vlat_fd_g", "vlat_Ce")})
        return sh

    def ttm1d_download(self):
        F, out = Ttm1dFieldsC(), {}
        for k, shp in self._ttm1d_shapes().items():
            isint = k in TTM1D_INT + TTM1D_GINT
            out[k] = np.zeros(shp, np.int32 if isint else np.float64)
            setattr(F, k, _ip(out[k]) if isint else _dp(out[k]))
        self._ck(self.L.imdgpu_ttm1d_download(self.h, C.byref(F)))
        return out

    def ttm1d_upload(self, **fields):
        F, keep, sh = Ttm1dFieldsC(), [], self._ttm1d_shapes()
        for k, v in fields.items():
            isint = k in TTM1D_INT + TTM1D_GINT
            a = np.ascontiguousarray(v, np.int32 if isint else np.float64).reshape(sh[k])
            keep.append(a)
            setattr(F, k, _ip(a) if isint else _dp(a))
        self._ck(self.L.imdgpu_ttm1d_upload(self.h, C.byref(F)))

    def EOS_te_from_r_ee(self, r, e):
        """imd_ttm_3D.c:2786: vectorised over host arrays, K."""
        r = np.ascontiguousarray(r, dtype=np.float64); e = np.ascontiguousarray(e, dtype=np.float64)
        out = np.zeros_like(r)
        self._ck(self.L.imdgpu_ttm_eos_te_from_r_ee(self.h, int(r.size), _dp(r), _dp(e), _dp(out)))
        return out

    def set_eos_triangulation(self, which, diag):
        """which diagonal the reference's triangulation puts into every grid cell of EOS table `which` (0 cve, 1 ee):
        array (nx-1, ny-1) of 0 (unknown: bilinear) / 1 ((i+1,j)-(i,j+1)) / 2 ((i,j)-(i+1,j+1)); None: bilinear everywhere"""
        if diag is None:
            self._ck(self.L.imdgpu_ttm_set_eos_triangulation(self.h, int(which), None))
            return
        a = np.ascontiguousarray(diag, dtype=np.uint8)
        self._ck(self.L.imdgpu_ttm_set_eos_triangulation(self.h, int(which), a.tobytes()))

    def eos_lookup(self, which, x, y):
        """the raw table look-up behind EOS_cve_from_r_te (which=0) / EOS_ee_from_r_te (which=1): the reference's
        lpi_interpolate_point (imd_ttm_3D.c:2722,2759), vectorised over host arrays, table coordinates"""
        x = np.ascontiguousarray(x, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
        out = np.zeros_like(x)
        self._ck(self.L.imdgpu_ttm_eos_lookup(self.h, int(which), int(x.size), _dp(x), _dp(y), _dp(out)))
        return out

    def update_fd(self, steps=0):
        self._ck(self.L.imdgpu_ttm_update_fd(self.h, int(steps)))

    def do_cell_activation(self):
        self._ck(self.L.imdgpu_ttm_cell_activation(self.h))

    def do_FILLMESH(self, steps=0):
        self._ck(self.L.imdgpu_ttm_fillmesh(self.h, int(steps)))

    def do_ADV(self, tau=1.0, steps=1):
        """do_ADV(tau) (imd_ttm_3D.c:1321-1458)"""
        self._ck(self.L.imdgpu_ttm_adv(self.h, float(tau), int(steps)))

    @staticmethod
    def _laser(direction, offset, mu, p_peak, t_0, sigma_t_squared):
        p = LaserParamsC()
        p.dir[:] = [int(v) for v in direction]
        p.offset, p.mu, p.p_peak, p.t_0, p.sigma_t_squared = float(offset), float(mu), float(p_peak), float(t_0), float(sigma_t_squared)
        return p

    def laser_rescale_ttm(self, steps, **laser):
        """laser_rescale_ttm (imd_laser.c:715-788): source term of MD step `steps`; laser = dict(direction, offset, mu,
        p_peak, t_0, sigma_t_squared) -- the reference's laser_* globals after init_laser"""
        self._ck(self.L.imdgpu_ttm_laser_rescale(self.h, C.byref(self._laser(**laser)), int(steps)))

    def set_laser(self, **laser):
        """calc_ttm applies the laser source after update_fd in every step from now on (no arguments: off)"""
        self._ck(self.L.imdgpu_ttm_set_laser(self.h, C.byref(self._laser(**laser)) if laser else None))

    # -- 1D Maxwell solver (imd_fdtd.c) on a one-row lattice -------------------------------------------------------------
    AL_DRUDE_LORENTZ = (2.73, 1.1174e+15, 7.6595e+15, 2.4024e+15, 4.5199e+14)     # fitDL's constants for Al, imd_ttm_3D.c:2921-2928

    def init_fdtd(self, I0, lam, sigma_t, t_0, srcx, bw=15, Sc=1.0, DL=AL_DRUDE_LORENTZ):
        """init_fdtd / init_pml (imd_fdtd.c:594-859); SI units as in the reference's parameter file"""
        p = FdtdParamsC()
        p.I0, p.lam, p.sigma_t, p.t_0, p.srcx, p.bw, p.Sc = float(I0), float(lam), float(sigma_t), float(t_0), float(srcx), int(bw), float(Sc)
        p.DL[:] = [float(v) for v in DL]
        self._ck(self.L.imdgpu_ttm_fdtd_init(self.h, C.byref(p)))

    def fdtd_set_dl(self, DL):
        a = None if DL is None else np.ascontiguousarray(DL, dtype=np.float64)
        self._ck(self.L.imdgpu_ttm_fdtd_set_dl(self.h, None if a is None else _dp(a)))

    def fdtd_set_dl_tables(self, tabs):
        """fitDL's tables DL1..DL5 (imd_amd.tricub.TricubTable each, or paths of files in the reference's layout); None: constants"""
        from . import tricub
        if tabs is None:
            self._ck(self.L.imdgpu_ttm_fdtd_set_dl_tables(self.h, None)); return
        tabs = [tricub.read_table(t) if isinstance(t, str) else t for t in tabs]
        if len(tabs) != 5:
            raise ValueError("fitDL takes five tables (DL1 .. DL5)")
        arr = (TricubTableC * 5)()
        for a, t in zip(arr, tabs):
            a.is_, a.js, a.ks = t.dims
            a.range[:] = [float(v) for v in t.range]
            a.x, a.y, a.z, a.f = _dp(t.x), _dp(t.y), _dp(t.z), _dp(t.f)
        self._ck(self.L.imdgpu_ttm_fdtd_set_dl_tables(self.h, arr))

    def fdtd_dl_lookup(self, which, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        out = np.zeros(len(xyz))
        self._ck(self.L.imdgpu_ttm_fdtd_dl_lookup(self.h, int(which), len(xyz), _dp(xyz), _dp(out)))
        return out

    def fdtd_fitdl(self, temp, md_temp, dens, ne):
        a = [np.ascontiguousarray(v, dtype=np.float64) for v in (temp, md_temp, dens, ne)]
        out = np.zeros((len(a[0]), 6))
        self._ck(self.L.imdgpu_ttm_fdtd_fitdl(self.h, len(a[0]), _dp(a[0]), _dp(a[1]), _dp(a[2]), _dp(a[3]), _dp(out)))
        return out

    def fdtd_cell_dl(self):
        """DL[nx, 6] of the row's cells as the last solver loop left them"""
        out = np.zeros((self.fd_dim[0], 6))
        self._ck(self.L.imdgpu_ttm_fdtd_fields(self.h, 8, _dp(out)))
        return out

    def do_fdtd(self, t_SI, nsub, tau_FDTD, laser_timefun):
        """nsub x do_fdtd(t) (imd_fdtd.c:128-306), t advancing by tau_FDTD"""
        self._ck(self.L.imdgpu_ttm_fdtd_substeps(self.h, float(t_SI), int(nsub), float(tau_FDTD), float(laser_timefun)))

    def fdtd_fields(self):
        out = {}
        for q, n in enumerate(("Ezx", "Hy", "Jzx", "Jlzx", "Pzx", "sigmay", "Jzx_old", "Jlzx_old")):
            out[n] = np.zeros(self.fd_dim[0])
            self._ck(self.L.imdgpu_ttm_fdtd_fields(self.h, q, _dp(out[n])))
        return out

    def fdtd_info(self):
        dt, e0, t = C.c_double(), C.c_double(), C.c_double()
        sx = C.c_int()
        self._ck(self.L.imdgpu_ttm_fdtd_get(self.h, C.byref(dt), C.byref(e0), C.byref(sx), C.byref(t)))
        return dict(dt_FDTD=dt.value, E0=e0.value, srcx_int=sx.value, t_SI=t.value)

    def get_eabs(self):
        """Eabs_local: absorbed laser energy [eV] booked by do_DIFF so far (imd_ttm_3D.c:1647-1651)"""
        v = C.c_double()
        self._ck(self.L.imdgpu_ttm_get_eabs(self.h, C.byref(v)))
        return v.value

    def ttm_fill_ghost_layers(self):
        self._ck(self.L.imdgpu_ttm_fill_ghost_layers(self.h))

    def do_DIFF(self, tau):
        self._ck(self.L.imdgpu_ttm_diff(self.h, float(tau)))

    def CFL_maxdt(self, steps=0):
        v = C.c_double()
        self._ck(self.L.imdgpu_ttm_cfl_maxdt(self.h, int(steps), C.byref(v)))
        return v.value

    def calc_ttm(self, steps=0):
        self._ck(self.L.imdgpu_calc_ttm(self.h, int(steps)))

    def move_atoms_ttm(self):
        e = C.c_double()
        self._ck(self.L.imdgpu_move_atoms_ttm(self.h, C.byref(e)))
        return e.value

    # -- profiling -----------------------------------------------------------------
    def profile(self, on=True):
        self._ck(self.L.imdgpu_profile_enable(self.h, 1 if on else 0))

    def profile_reset(self):
        self._ck(self.L.imdgpu_profile_reset(self.h))

    def profile_get(self, name):
        n = C.c_longlong()
        ms = C.c_double()
        self._ck(self.L.imdgpu_profile_get(self.h, name.encode(), C.byref(n), C.byref(ms)))
        return n.value, ms.value
