"""TEST INFRASTRUCTURE -- ctypes front end of the CPU oracle (oracle/imd_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package imd_amd never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liborcimd.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))
            and not f.startswith("ref_")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liborcimd.so"])
    return so


class _Table(C.Structure):
    _fields_ = [("ncols", C.c_int), ("maxsteps", C.c_int),
                ("begin", C.POINTER(C.c_double)), ("end", C.POINTER(C.c_double)),
                ("step", C.POINTER(C.c_double)), ("invstep", C.POINTER(C.c_double)),
                ("len", C.POINTER(C.c_int)), ("table", C.POINTER(C.c_double))]


class _Sys(C.Structure):
    _fields_ = [("dim", C.c_int * 3), ("gdim", C.c_int * 3), ("pbc", C.c_int * 3),
                ("ntypes", C.c_int), ("cap", C.c_int), ("nall", C.c_int), ("ninner", C.c_int),
                ("box", C.c_double * 9), ("tbox", C.c_double * 9),
                ("cellsz", C.c_double), ("nbl_margin", C.c_double), ("timestep", C.c_double),
                ("n", C.POINTER(C.c_int)), ("inner", C.POINTER(C.c_int)), ("nq", C.POINTER(C.c_int)),
                ("ort", C.POINTER(C.c_double)), ("impuls", C.POINTER(C.c_double)),
                ("kraft", C.POINTER(C.c_double)), ("nbl_pos", C.POINTER(C.c_double)),
                ("masse", C.POINTER(C.c_double)), ("pot_eng", C.POINTER(C.c_double)),
                ("rho", C.POINTER(C.c_double)), ("dF", C.POINTER(C.c_double)),
                ("sorte", C.POINTER(C.c_int)), ("nummer", C.POINTER(C.c_int)),
                ("numneighs", C.POINTER(C.c_int)),
                ("cl_off", C.POINTER(C.c_int)), ("tl", C.POINTER(C.c_int)), ("tb", C.POINTER(C.c_int)),
                ("tl_cap", C.c_int), ("tb_cap", C.c_int), ("tb_len", C.c_int),
                ("have_valid_nbl", C.c_int), ("nbl_count", C.c_int),
                ("pair", _Table), ("rhotab", _Table), ("embed", _Table),
                ("have_pair", C.c_int), ("have_eam", C.c_int), ("count_ttm_neigh", C.c_int),
                ("tot_pot_energy", C.c_double), ("virial", C.c_double), ("tot_kin_energy", C.c_double),
                ("is_short", C.c_int), ("restrictions", C.c_double * 3),
                ("fdi", C.POINTER(C.c_int)), ("fdj", C.POINTER(C.c_int)), ("fdk", C.POINTER(C.c_int))]


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_new.restype = C.POINTER(_Sys)
        L.orc_new.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_double),
                              C.POINTER(C.c_int), C.c_double, C.c_double, C.c_double]
        L.orc_free.argtypes = [C.POINTER(_Sys)]
        L.orc_set_table.argtypes = [C.POINTER(_Sys), C.c_int, C.c_int, C.c_int] + \
            [C.POINTER(C.c_double)] * 4 + [C.POINTER(C.c_int), C.POINTER(C.c_double)]
        for f in ("orc_do_boundaries", "orc_fix_cells", "orc_make_nblist", "orc_check_nblist",
                  "orc_move_atoms_nve", "orc_make_cell_lists"):
            getattr(L, f).argtypes = [C.POINTER(_Sys)]
            getattr(L, f).restype = None
        L.orc_send_cells.argtypes = [C.POINTER(_Sys), C.c_int, C.c_int]
        L.orc_send_forces.argtypes = [C.POINTER(_Sys), C.c_int]
        L.orc_calc_forces.argtypes = [C.POINTER(_Sys), C.c_int]
        L.orc_global_cell_dim.argtypes = [C.POINTER(C.c_double), C.c_double,
                                          C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_pair_int.argtypes = [C.POINTER(_Table), C.c_int, C.c_int, C.c_double,
                                   C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.orc_val_func.restype = C.c_double
        L.orc_val_func.argtypes = [C.POINTER(_Table), C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int)]
        L.orc_deriv_func.restype = C.c_double
        L.orc_deriv_func.argtypes = L.orc_val_func.argtypes
        L.orc_set_nblist.argtypes = [C.POINTER(_Sys), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.c_int]
        L.orc_init_threepoint.argtypes = [C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def global_cell_dim(box, cellsz, cd=(1, 1, 1)):
    b = np.ascontiguousarray(box, dtype=np.float64).reshape(9)
    cdv = np.asarray(cd, dtype=np.int32)
    out = np.zeros(3, dtype=np.int32)
    lib().orc_global_cell_dim(_dp(b), float(cellsz), _ip(cdv), _ip(out))
    return tuple(int(v) for v in out)


class OracleSystem:
    """Padded-cell system; numpy views alias the C arrays (shape (nall, cap[,3]))."""

    def __init__(self, dim, cap, ntypes, box, pbc, cellsz, nbl_margin, timestep):
        L = lib()
        d = np.asarray(dim, dtype=np.int32)
        b = np.ascontiguousarray(box, dtype=np.float64).reshape(9)
        p = np.asarray(pbc, dtype=np.int32)
        self._p = L.orc_new(_ip(d), int(cap), int(ntypes), _dp(b), _ip(p),
                            float(cellsz), float(nbl_margin), float(timestep))
        self.s = self._p.contents
        s = self.s
        self.dim = tuple(int(v) for v in d)
        self.cap = int(cap)
        nall, ninner = s.nall, s.ninner
        as_arr = np.ctypeslib.as_array
        self.n = as_arr(s.n, (nall,))
        self.inner = as_arr(s.inner, (ninner,))
        self.nq = as_arr(s.nq, (ninner, 14))
        for name in ("ort", "impuls", "kraft", "nbl_pos"):
            setattr(self, name, as_arr(getattr(s, name), (nall, cap, 3)))
        for name in ("masse", "pot_eng", "rho", "dF", "sorte", "nummer", "numneighs"):
            setattr(self, name, as_arr(getattr(s, name), (nall, cap)))
        self.cl_off = as_arr(s.cl_off, (nall + 1,))
        self._keep = []

    def close(self):
        if self._p is not None:
            lib().orc_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_table(self, which, t):
        """t: dict with ncols,maxsteps,begin,end,step,invstep,len,table (after init_threepoint)."""
        a = {k: np.ascontiguousarray(t[k], dtype=np.float64) for k in ("begin", "end", "step", "invstep", "table")}
        ln = np.ascontiguousarray(t["len"], dtype=np.int32)
        lib().orc_set_table(self._p, which, int(t["ncols"]), int(t["maxsteps"]), _dp(a["begin"]), _dp(a["end"]),
                            _dp(a["step"]), _dp(a["invstep"]), _ip(ln), _dp(a["table"]))

    # thin wrappers -----------------------------------------------------------
    def fix_cells(self): lib().orc_fix_cells(self._p)
    def send_cells(self, what=0, full_shell=0): lib().orc_send_cells(self._p, what, full_shell)
    def send_forces(self, what): lib().orc_send_forces(self._p, what)
    def make_nblist(self): lib().orc_make_nblist(self._p)
    def calc_forces(self, steps=0): lib().orc_calc_forces(self._p, steps)
    def check_nblist(self): lib().orc_check_nblist(self._p)
    def move_atoms_nve(self): lib().orc_move_atoms_nve(self._p)

    def set_nblist(self, tl, tb):
        tl = np.ascontiguousarray(tl, dtype=np.int32); tb = np.ascontiguousarray(tb, dtype=np.int32)
        lib().orc_set_nblist(self._p, _ip(tl), len(tl), _ip(tb), len(tb))

    @property
    def tl(self):
        nin = int(sum(self.n[c] for c in self.inner))
        return np.ctypeslib.as_array(self.s.tl, (nin + 1,))

    @property
    def tb(self):
        return np.ctypeslib.as_array(self.s.tb, (max(self.s.tb_len, 1),))[: self.s.tb_len]

    # helpers -------------------------------------------------------------------
    def inner_mask(self):
        m = np.zeros(self.s.nall, dtype=bool)
        m[self.inner] = True
        return m

    def gather_inner(self, name):
        """Concatenate a per-atom field over inner cells in `cells` order."""
        arr = getattr(self, name)
        return np.concatenate([arr[c, : self.n[c]] for c in self.inner], axis=0)

    def fill_from_positions(self, pos, mom, mass, sorte=None, nummer=None):
        """Bin atoms like the generator does (cell_coord, then append in input order)."""
        pos = np.asarray(pos, dtype=np.float64); n = len(pos)
        t = np.array(self.s.tbox[:]).reshape(3, 3)
        gd = np.array(self.s.gdim[:])
        c = (gd * (pos @ t.T)).astype(np.int64)
        c = np.clip(c, 0, gd - 1) + 1
        lin = (c[:, 0] * self.dim[1] + c[:, 1]) * self.dim[2] + c[:, 2]
        order = np.argsort(lin, kind="stable")
        cnt = np.bincount(lin, minlength=self.s.nall)
        assert cnt.max() <= self.cap, "cell capacity %d < %d" % (self.cap, cnt.max())
        start = np.concatenate([[0], np.cumsum(cnt)])
        slot = np.arange(n) - start[lin[order]]
        cl = lin[order]
        self.n[:] = cnt
        self.ort[cl, slot] = pos[order]
        self.impuls[cl, slot] = np.asarray(mom, dtype=np.float64)[order]
        self.masse[cl, slot] = np.broadcast_to(np.asarray(mass, dtype=np.float64), (n,))[order]
        self.sorte[cl, slot] = 0 if sorte is None else np.asarray(sorte)[order]
        self.nummer[cl, slot] = (np.arange(n) if nummer is None else np.asarray(nummer))[order]
        self.s.have_valid_nbl = 0

    def load_cells(self, cell_n, per_cell_fields):
        """Fill cells from per-cell lists: per_cell_fields[name][c] -> array(n_c[,3])."""
        self.n[:] = cell_n
        for name, lst in per_cell_fields.items():
            arr = getattr(self, name)
            for c, v in enumerate(lst):
                if len(v):
                    arr[c, : len(v)] = v


# ---------------------------------------------------------------------------
# reader for the dumps written by oracle/ref_probe.c
# ---------------------------------------------------------------------------

def read_probe(path):
    buf = open(path, "rb").read()
    assert buf[:8] == b"IMDPROBE", "bad magic"
    assert buf[-8:] == b"ENDPROBE", "truncated probe dump"
    off = [8]

    def ri(n=1):
        v = np.frombuffer(buf, dtype="<i4", count=n, offset=off[0]); off[0] += 4 * n
        return v if n > 1 else int(v[0])

    def rd(n=1):
        v = np.frombuffer(buf, dtype="<f8", count=n, offset=off[0]); off[0] += 8 * n
        return v if n > 1 else float(v[0])

    P = {}
    P["version"] = ri(); P["flags"] = ri(); P["step"] = ri(); P["myid"] = ri()
    P["ntypes"] = ri(); nall = P["nallcells"] = ri(); ncells = P["ncells"] = ri()
    P["cell_dim"] = ri(3).copy(); P["global_cell_dim"] = ri(3).copy(); P["pbc_dirs"] = ri(3).copy()
    P["cpu_dim"] = ri(3).copy(); P["my_coord"] = ri(3).copy()
    P["natoms"] = ri(); P["have_valid_nbl"] = ri(); P["nbl_count"] = ri(); P["last_nbl_len"] = ri()
    P["box"] = rd(9).copy().reshape(3, 3)
    P["cellsz"] = rd(); P["nbl_margin"] = rd(); P["timestep"] = rd()
    P["tot_pot_energy"] = rd(); P["virial"] = rd(); P["tot_kin_energy"] = rd()

    def rtab():
        ncols = ri(); maxsteps = ri()
        if ncols == 0:
            return None
        t = {"ncols": ncols, "maxsteps": maxsteps}
        for k in ("begin", "end", "step", "invstep"):
            t[k] = np.atleast_1d(rd(ncols)).copy()
        t["len"] = np.atleast_1d(ri(ncols)).copy()
        t["table"] = rd((maxsteps + 2) * ncols).copy()
        return t

    P["pair_pot"] = rtab(); P["rho_h_tab"] = rtab(); P["embed_pot"] = rtab()
    cn = P["cell_n"] = np.atleast_1d(ri(nall)).copy()
    ntot = int(cn.sum())
    rec = np.dtype([("nummer", "<i4"), ("sorte", "<i4"), ("ort", "<f8", 3)])
    allat = np.frombuffer(buf, dtype=rec, count=ntot, offset=off[0]); off[0] += rec.itemsize * ntot
    P["all_nummer"] = allat["nummer"].copy(); P["all_sorte"] = allat["sorte"].copy()
    P["all_ort"] = allat["ort"].copy()
    P["cells"] = np.atleast_1d(ri(ncells)).copy()
    nin = P["ninner_atoms"] = ri()
    rec2 = np.dtype([("masse", "<f8"), ("impuls", "<f8", 3), ("kraft", "<f8", 3), ("pot_eng", "<f8"),
                     ("eam_rho", "<f8"), ("eam_dF", "<f8"), ("nbl_pos", "<f8", 3), ("numneighs", "<i4")])
    inn = np.frombuffer(buf, dtype=rec2, count=nin, offset=off[0]); off[0] += rec2.itemsize * nin
    for k in rec2.names:
        P[k] = inn[k].copy()
    if P["flags"] & 2:
        P["cl_off"] = np.atleast_1d(ri(nall)).copy()
        P["tl"] = np.atleast_1d(ri(nin + 1)).copy()
        P["tb"] = np.atleast_1d(ri(int(P["tl"][-1]))).copy() if P["tl"][-1] > 0 else np.zeros(0, np.int32)
        P["cnbrs"] = ri(ncells * 15).copy().reshape(ncells, 15)
    assert buf[off[0]:off[0] + 8] == b"ENDPROBE"
    return P


TTM1D_CELL = ("natoms", "natoms_old", "temp", "xi", "md_temp", "source", "vcomx", "vcomy", "vcomz", "dens", "fd_k", "fd_g", "Z",
              "Ce", "ne", "U", "temp2", "U2")
TTM1D_GLOB = ("natomsglobal", "fluxfromleft", "fluxfromright", "vcomxglobal", "vcomyglobal", "vcomzglobal", "xiarr_global")
TTM1D_VLAT = ("temp", "md_temp", "dens", "fd_k", "fd_g", "Z", "Ce", "ne")


def read_ttm1d_probe(path):
    """dump of oracle/ref_probe_ttm1d.c: the reference's 1D lattice (imd_ttm.c) of one rank, stage by stage"""
    buf = open(path, "rb").read()
    assert buf[:8] == b"IMDTT1DP" and buf[-8:] == b"ENDTT1DP", "bad / truncated 1D TTM probe dump"
    off = [8]

    def ri(n=1):
        v = np.frombuffer(buf, dtype="<i4", count=n, offset=off[0]); off[0] += 4 * n
        return v if n > 1 else int(v[0])

    def rd(n=1):
        v = np.frombuffer(buf, dtype="<f8", count=n, offset=off[0]); off[0] += 8 * n
        return v.copy() if n > 1 else float(v[0])
    T = dict(version=ri(), step=ri(), fd_min_atoms=ri(), fd_n_timesteps=ri(), vlatbuffer=ri())
    for k in ("fd_hx", "fd_hy", "fd_hz", "fd_vol", "neighvol", "atomic_weight", "atomic_charge", "timestep", "vlatdens", "Temin"):
        T[k] = rd()
    T["stages"] = {}
    while buf[off[0]:off[0] + 8] != b"ENDTT1DP":
        tag = buf[off[0]:off[0] + 16].split(b"\0")[0].decode(); off[0] += 16
        lx, nx, nv = ri(), ri(), ri()
        cells = rd(lx * len(TTM1D_CELL)).reshape(lx, len(TTM1D_CELL))
        glob = rd(nx * len(TTM1D_GLOB)).reshape(nx, len(TTM1D_GLOB))
        vlat = rd(nv * len(TTM1D_VLAT)).reshape(nv, len(TTM1D_VLAT)) if nv else np.zeros((0, len(TTM1D_VLAT)))
        sc = rd(4)
        T["stages"][tag] = dict(cells={k: cells[:, q].copy() for q, k in enumerate(TTM1D_CELL)},
                                glob={k: glob[:, q].copy() for q, k in enumerate(TTM1D_GLOB)},
                                vlat={k: vlat[:, q].copy() for q, k in enumerate(TTM1D_VLAT)},
                                last_active_cell=int(sc[0]), vlattice_proc=int(sc[1]), diff_substeps=int(sc[2]), max_dt_ttm=float(sc[3]))
        T["local_fd_dim"], T["global_fd_dim"], T["vlatdim"] = lx, nx, nv
    return T


def system_from_probe(P, cap=None, load_outputs=False):
    """Rebuild the reference's cell arrangement (same slots) inside the oracle."""
    cn = P["cell_n"]
    cap = cap or int(max(cn.max() + 8, 16))
    s = OracleSystem(P["cell_dim"], cap, P["ntypes"], P["box"], P["pbc_dirs"],
                     P["cellsz"], P["nbl_margin"], P["timestep"])
    start = np.concatenate([[0], np.cumsum(cn)])
    s.n[:] = cn
    for c in range(len(cn)):
        a, b = start[c], start[c + 1]
        if b > a:
            s.ort[c, : b - a] = P["all_ort"][a:b]
            s.sorte[c, : b - a] = P["all_sorte"][a:b]
            s.nummer[c, : b - a] = P["all_nummer"][a:b]
    # inner-only fields
    k = 0
    for c in P["cells"]:
        m = cn[c]
        s.masse[c, :m] = P["masse"][k:k + m]
        s.impuls[c, :m] = P["impuls"][k:k + m]
        s.nbl_pos[c, :m] = P["nbl_pos"][k:k + m]
        if load_outputs:
            s.kraft[c, :m] = P["kraft"][k:k + m]
            s.pot_eng[c, :m] = P["pot_eng"][k:k + m]
            s.rho[c, :m] = P["eam_rho"][k:k + m]
            s.dF[c, :m] = P["eam_dF"][k:k + m]
        k += m
    for which, key in ((0, "pair_pot"), (1, "rho_h_tab"), (2, "embed_pot")):
        if P.get(key) is not None:
            s.set_table(which, P[key])
    s.s.count_ttm_neigh = 1 if (P["flags"] & 4) else 0
    return s


# ---------------------------------------------------------------------------
# TTM oracle (oracle/ttm_oracle.c)
# ---------------------------------------------------------------------------

class _Eos(C.Structure):
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("xmin", C.c_double), ("xmax", C.c_double),
                ("ymin", C.c_double), ("ymax", C.c_double), ("z", C.POINTER(C.c_double)), ("diag", C.c_void_p)]


class _Ttm(C.Structure):
    _fields_ = [("dim", C.c_int * 3), ("ext", C.c_int * 3), ("pbc", C.c_int * 3), ("ncell", C.c_size_t),
                ("fd_min_atoms", C.c_int), ("cur", C.c_int),
                ("h", C.c_double * 3), ("vol", C.c_double), ("neighvol", C.c_double),
                ("atomic_weight", C.c_double), ("atomic_charge", C.c_double), ("omega_laser", C.c_double),
                ("Temin", C.c_double), ("max_dt_ttm", C.c_double),
                ("natoms", C.POINTER(C.c_int)), ("natoms_old", C.POINTER(C.c_int)),
                ("temp", C.POINTER(C.c_double) * 2), ("U", C.POINTER(C.c_double) * 2),
                ("xi", C.POINTER(C.c_double)), ("md_temp", C.POINTER(C.c_double)), ("source", C.POINTER(C.c_double)),
                ("vcom", C.POINTER(C.c_double)), ("dens", C.POINTER(C.c_double)), ("fd_k", C.POINTER(C.c_double)),
                ("fd_g", C.POINTER(C.c_double)), ("Z", C.POINTER(C.c_double)), ("Ce", C.POINTER(C.c_double)),
                ("ne", C.POINTER(C.c_double)), ("cve", _Eos), ("ee", _Eos), ("flux", C.POINTER(C.c_int)), ("adv", C.c_int), ("laser_active", C.c_int), ("eabs", C.c_double)]


_TTM_READY = False


def _ttm_lib():
    global _TTM_READY
    L = lib()
    if not _TTM_READY:
        tp, sp = C.POINTER(_Ttm), C.POINTER(_Sys)
        L.orc_ttm_new.restype = tp
        L.orc_ttm_new.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int),
                                  C.c_int, C.c_double, C.c_double, C.c_double]
        L.orc_ttm_free.argtypes = [tp]
        L.orc_ttm_set_eos.argtypes = [tp, C.c_int, C.c_int, C.c_int] + [C.c_double] * 4 + [C.POINTER(C.c_double)]
        L.orc_ttm_update_fd.argtypes = [tp, sp, C.c_int]
        L.orc_ttm_cell_activation.argtypes = [tp]
        L.orc_ttm_adv.argtypes = [tp, C.c_double, C.c_int]
        L.orc_ttm_laser_rescale.argtypes = [tp, C.POINTER(C.c_int)] + [C.c_double] * 6
        L.orc_ttm_fillmesh.argtypes = [tp, C.c_int]
        L.orc_ttm_fill_ghosts.argtypes = [tp]
        L.orc_ttm_diff.argtypes = [tp, C.c_double, C.c_int]
        L.orc_ttm_cfl.argtypes = [tp, C.c_int, C.c_double, C.c_int]
        L.orc_ttm_calc.argtypes = [tp, sp, C.c_int, C.c_int]
        L.orc_move_atoms_ttm.argtypes = [sp, tp]
        for f in ("orc_mean_charge", "orc_get_gamma", "orc_get_kappa"):
            getattr(L, f).restype = C.c_double
        L.orc_mean_charge.argtypes = [C.c_double] * 4
        L.orc_get_gamma.argtypes = [C.c_double] * 5
        L.orc_get_kappa.argtypes = [C.c_double] * 5
        L.orc_eos_cve.restype = C.c_double
        L.orc_eos_cve.argtypes = [tp, C.c_double, C.c_double]
        L.orc_eos_ee.restype = C.c_double
        L.orc_eos_ee.argtypes = [tp, C.c_double, C.c_double]
        L.orc_eos_lookup.restype = C.c_double
        L.orc_eos_lookup.argtypes = [C.POINTER(_Eos), C.c_double, C.c_double]
        L.orc_eos_te_from_r_ee.restype = C.c_double
        L.orc_eos_te_from_r_ee.argtypes = [tp, C.c_double, C.c_double]
        _TTM_READY = True
    return L


class OracleTTM:
    """FD lattice bound to an OracleSystem; numpy views of shape dim (ghosts incl.)."""

    def __init__(self, system, fd_ext, fd_min_atoms, atomic_weight, atomic_charge, r2_cut_pair, fd_h=None):
        L = _ttm_lib()
        self.sys = system
        md = np.asarray(system.dim, dtype=np.int32)
        ext = np.asarray(fd_ext, dtype=np.int32)
        box = np.array(system.s.box[:]).reshape(3, 3)
        if fd_h is None:      # fd_h = fd_ext * box/global_cell_dim (imd_geom_3d.c:243-246)
            fd_h = np.array([ext[d] * box[d, d] * (1.0 / (md[d] - 2)) for d in range(3)])
        h = np.ascontiguousarray(fd_h, dtype=np.float64)
        pbc = np.array(system.s.pbc[:], dtype=np.int32)
        self._p = L.orc_ttm_new(_ip(md), _ip(ext), _dp(h), _ip(pbc), int(fd_min_atoms), float(atomic_weight),
                                float(atomic_charge), float(r2_cut_pair))
        self.t = self._p.contents
        self.dim = tuple(int(v) for v in self.t.dim)
        n = int(self.t.ncell)
        aa = np.ctypeslib.as_array
        for name in ("natoms", "natoms_old", "xi", "md_temp", "source", "dens", "fd_k", "fd_g", "Z", "Ce", "ne"):
            setattr(self, name, aa(getattr(self.t, name), (n,)).reshape(self.dim))
        self.vcom = aa(self.t.vcom, (3 * n,)).reshape(self.dim + (3,))
        self.flux = aa(self.t.flux, (8 * n,)).reshape(self.dim + (8,))
        self._temp = [aa(self.t.temp[q], (n,)).reshape(self.dim) for q in (0, 1)]
        self._U = [aa(self.t.U[q], (n,)).reshape(self.dim) for q in (0, 1)]

    temp = property(lambda self: self._temp[self.t.cur])     # l1
    temp2 = property(lambda self: self._temp[self.t.cur ^ 1])
    U = property(lambda self: self._U[self.t.cur])
    U2 = property(lambda self: self._U[self.t.cur ^ 1])

    def close(self):
        if self._p is not None:
            _ttm_lib().orc_ttm_free(self._p)
            self._p = None

    def set_eos(self, which, nx, ny, xmin, xmax, ymin, ymax, z):
        z = np.ascontiguousarray(z, dtype=np.float64).reshape(-1)
        _ttm_lib().orc_ttm_set_eos(self._p, which, nx, ny, xmin, xmax, ymin, ymax, _dp(z))

    def update_fd(self, steps): _ttm_lib().orc_ttm_update_fd(self._p, self.sys._p, steps)
    def adv(self, tau, steps): _ttm_lib().orc_ttm_adv(self._p, float(tau), int(steps))
    def laser_rescale(self, direction, offset, mu, p_peak, t_0, sigma_t_squared, time):
        d = np.ascontiguousarray(direction, dtype=np.int32)
        _ttm_lib().orc_ttm_laser_rescale(self._p, _ip(d), float(offset), float(mu), float(p_peak), float(t_0),
                                         float(sigma_t_squared), float(time))

    def cell_activation(self): _ttm_lib().orc_ttm_cell_activation(self._p)
    def fillmesh(self, steps): _ttm_lib().orc_ttm_fillmesh(self._p, steps)
    def fill_ghosts(self): _ttm_lib().orc_ttm_fill_ghosts(self._p)
    def diff(self, tau, diff_substeps): _ttm_lib().orc_ttm_diff(self._p, tau, diff_substeps)
    def cfl(self, steps, timestep, n): _ttm_lib().orc_ttm_cfl(self._p, steps, timestep, n)
    def calc_ttm(self, steps, diff_substeps): _ttm_lib().orc_ttm_calc(self._p, self.sys._p, steps, diff_substeps)
    def move_atoms_ttm(self): _ttm_lib().orc_move_atoms_ttm(self.sys._p, self._p)
    def set_eos_diag(self, which, diag):
        L = _ttm_lib()
        L.orc_ttm_set_eos_diag.argtypes = [C.POINTER(_Ttm), C.c_int, C.c_char_p]
        L.orc_ttm_set_eos_diag(self._p, int(which), None if diag is None else np.ascontiguousarray(diag, np.uint8).tobytes())

    def eos_lookup(self, which, x, y):
        tab = self.t.ee if which else self.t.cve
        return _ttm_lib().orc_eos_lookup(C.byref(tab), float(x), float(y))

    def eos_ee(self, r, tK): return _ttm_lib().orc_eos_ee(self._p, float(r), float(tK))
    def eos_te_from_r_ee(self, r, e): return _ttm_lib().orc_eos_te_from_r_ee(self._p, float(r), float(e))


TTM_REC = np.dtype([("natoms", "<i4"), ("natoms_old", "<i4"), ("temp", "<f8"), ("xi", "<f8"), ("md_temp", "<f8"),
                    ("source", "<f8"), ("vcom", "<f8", 3), ("dens", "<f8"), ("fd_k", "<f8"), ("fd_g", "<f8"),
                    ("Z", "<f8"), ("Ce", "<f8"), ("ne", "<f8"), ("U", "<f8"), ("temp2", "<f8"), ("U2", "<f8")])


def read_ttm_probe(path):
    """Dumps of oracle/ref_probe_ttm.c: header + named lattice snapshots."""
    buf = open(path, "rb").read()
    assert buf[:8] == b"IMDTTMPR" and buf[-8:] == b"ENDTTMPR"
    off = 8
    hi = np.frombuffer(buf, "<i4", 11, off); off += 44
    hd = np.frombuffer(buf, "<f8", 10, off); off += 80
    H = dict(version=int(hi[0]), step=int(hi[1]), fd_dim=hi[2:5].copy(), fd_ext=hi[5:8].copy(),
             fd_min_atoms=int(hi[8]), diff_substeps=int(hi[9]), fd_n_timesteps=int(hi[10]),
             fd_h=hd[0:3].copy(), fd_vol=float(hd[3]), neighvol=float(hd[4]), atomic_weight=float(hd[5]),
             atomic_charge=float(hd[6]), timestep=float(hd[7]), tau=float(hd[8]), omega_laser=float(hd[9]))
    n = int(np.prod(H["fd_dim"]))
    stages = {}
    while off < len(buf) - 8:
        tag = buf[off:off + 16].split(b"\0")[0].decode(); off += 16
        mdt = float(np.frombuffer(buf, "<f8", 1, off)[0]); off += 8
        if mdt == -3.0:               # absorbed energy so far and laser_active
            v = np.frombuffer(buf, "<f8", 2, off).copy(); off += 16
            stages[tag] = dict(max_dt_ttm=mdt, eabs=float(v[0]), laser_active=int(v[1]))
            continue
        if mdt == -2.0:               # parameter record (dump_laser_params): 10 doubles
            v = np.frombuffer(buf, "<f8", 10, off).copy(); off += 80
            stages[tag] = dict(max_dt_ttm=mdt, params=dict(dir=v[0:3].astype(int), offset=float(v[3]), mu=float(v[4]),
                               p_peak=float(v[5]), t_0=float(v[6]), sigma_t_squared=float(v[7]), sigma_e=float(v[8]),
                               sigma_t=float(v[9])))
            continue
        if mdt == -1.0:               # int record: the 8 atom-flux counters per cell (dump_flux)
            fl = np.frombuffer(buf, "<i4", 8 * n, off).reshape(tuple(H["fd_dim"]) + (8,)).copy(); off += 32 * n
            stages[tag] = dict(max_dt_ttm=mdt, flux=fl)
            continue
        rec = np.frombuffer(buf, TTM_REC, n, off).reshape(tuple(H["fd_dim"])).copy(); off += TTM_REC.itemsize * n
        stages[tag] = dict(max_dt_ttm=mdt, lat=rec)
    H["stages"] = stages
    return H


# ---------------------------------------------------------------------------
# 1D Maxwell solver (oracle/fdtd_oracle.c) and the dumps of oracle/ref_probe_fdtd.c
# ---------------------------------------------------------------------------

class _Fdtd(C.Structure):
    _fields_ = [("nx", C.c_int), ("fd_min_atoms", C.c_int), ("bw", C.c_int), ("srcx_int", C.c_int), ("containssrc", C.c_int),
                ("laser_active", C.c_int)] + \
               [(n, C.c_double) for n in ("Imp0", "fd_dx", "Sc_FDTD", "dt", "omega_laser", "Nlambdax", "E0", "E0_init", "t0",
                                          "sigma_t_squared", "laser_timefun")] + \
               [(n, C.POINTER(C.c_double)) for n in ("Ezx", "Hy", "Jzx", "Jlzx", "Pzx", "source", "sigmay", "Jzx_old", "Jlzx_old",
                                                     "DL")] + [("natoms", C.POINTER(C.c_int))]


FDTD_FIELDS = ("Ezx", "Hy", "Jzx", "Jlzx", "Pzx", "source", "sigmay", "Jzx_old", "Jlzx_old")


class OracleFDTD:
    def __init__(self, nx, fd_h_x, I0, lam, sigma_t, t0, srcx, bw, Sc, fd_min_atoms=5):
        L = lib()
        L.orc_fdtd_new.restype = C.POINTER(_Fdtd)
        L.orc_fdtd_new.argtypes = [C.c_int] + [C.c_double] * 6 + [C.c_int, C.c_double, C.c_int]
        L.orc_fdtd_free.argtypes = [C.POINTER(_Fdtd)]
        L.orc_fdtd_begin_md_step.argtypes = [C.POINTER(_Fdtd), C.c_double, C.c_double, C.POINTER(C.c_double)]
        L.orc_fdtd_step.argtypes = [C.POINTER(_Fdtd), C.c_double]
        self._p = L.orc_fdtd_new(int(nx), float(fd_h_x), float(I0), float(lam), float(sigma_t), float(t0), float(srcx),
                                 int(bw), float(Sc), int(fd_min_atoms))
        self.f = self._p.contents
        aa = np.ctypeslib.as_array
        for n in FDTD_FIELDS:
            setattr(self, n, aa(getattr(self.f, n), (nx,)))
        self.DL = aa(self.f.DL, (6 * nx,)).reshape(nx, 6)
        self.natoms = aa(self.f.natoms, (nx,))

    def begin_md_step(self, t_SI, timestep):
        tau = C.c_double()
        n = lib().orc_fdtd_begin_md_step(self._p, float(t_SI), float(timestep), C.byref(tau))
        return n, tau.value

    def step(self, t): lib().orc_fdtd_step(self._p, float(t))

    def close(self):
        if self._p is not None:
            lib().orc_fdtd_free(self._p)
            self._p = None


def read_fdtd_probe(path):
    b = open(path, "rb").read()
    assert b[:8] == b"IMDFDTD1" and b[-8:] == b"ENDFDTD1"
    off = 8
    hi = np.frombuffer(b, "<i4", 4, off); off += 16
    hd = np.frombuffer(b, "<f8", 8, off); off += 64
    nx = int(hi[0])
    H = dict(nx=nx, bw=int(hi[1]), srcx_int=int(hi[2]), containssrc=int(hi[3]), dt_FDTD=float(hd[0]), fd_dx=float(hd[1]),
             E0=float(hd[2]), E0_init=float(hd[3]), Nlambdax=float(hd[4]), Imp0=float(hd[5]), sigma_t_squared=float(hd[6]),
             omega_laser=float(hd[7]))
    rec = np.dtype([("natoms", "<i4"), ("dens", "<f8"), ("temp", "<f8"), ("ne", "<f8"), ("DL", "<f8", 6)])
    H["cells"] = np.frombuffer(b, rec, nx, off).copy(); off += rec.itemsize * nx
    H["dumps"], H["md"] = {}, []

    def rd():
        nonlocal off
        tag = b[off:off + 16].split(b"\0")[0].decode(); off += 16
        t, ltf = np.frombuffer(b, "<f8", 2, off); off += 16
        a = np.frombuffer(b, "<f8", 9 * nx, off).reshape(nx, 9).copy(); off += 72 * nx
        H["dumps"][tag] = dict(t_SI=float(t), laser_timefun=float(ltf), **{n: a[:, q] for q, n in enumerate(FDTD_FIELDS)})
        return tag
    rd()
    while off < len(b) - 8:
        nsub, act = np.frombuffer(b, "<i4", 2, off); off += 8
        H["md"].append((int(nsub), int(act)))
        while not rd().startswith("md"):
            pass
    return H


# ---------------------------------------------------------------------------
# 1D TTM oracle (oracle/ttm1d_oracle.c): imd_ttm.c's own variant
# ---------------------------------------------------------------------------
class _Ttm1d(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("nx", "nranks", "vlatdim", "vlatbuffer", "fd_min_atoms", "fd_n_timesteps", "cur", "vcur",
                                       "last_active", "vlat_proc", "diff_substeps", "adv")] + \
               [(k, C.c_double) for k in ("hx", "vol", "neighvol", "aw", "ac", "vlatdens", "Temin", "max_dt_ttm", "omega_laser")] + \
               [("natoms", C.POINTER(C.c_int)), ("natoms_old", C.POINTER(C.c_int)),
                ("temp", C.POINTER(C.c_double) * 2), ("U", C.POINTER(C.c_double) * 2)] + \
               [(k, C.POINTER(C.c_double)) for k in ("xi", "md_temp", "source", "dens", "fd_k", "fd_g", "Z", "Ce", "ne")] + \
               [(k, C.POINTER(C.c_int)) for k in ("fluxl", "fluxr", "totneigh", "natomsglobal")] + \
               [("vcom", C.POINTER(C.c_double) * 3), ("xiarr", C.POINTER(C.c_double)),
                ("vtemp", C.POINTER(C.c_double) * 2), ("vmd", C.POINTER(C.c_double) * 2)] + \
               [(k, C.POINTER(C.c_double)) for k in ("vk", "vg", "vCe", "vZ", "vne")] + [("eos", C.c_void_p)] + \
               [(k, C.c_int) for k in ("tmm_on", "laser_active", "ecut", "eps_bb_rows", "K12_rows")] + \
               [(k, C.c_double) for k in ("I0", "lam", "t0", "sigma_t", "t1", "sigma_t1", "abs_threshold", "laser_threshold", "k0", "tmm_time",
                                          "I_t", "refl", "trans", "absorbed", "eabs")] + \
               [(k, C.POINTER(C.c_double)) for k in ("eps_re", "eps_im", "qabs", "eps_bb", "K12")] + \
               [("tmm_active", C.POINTER(C.c_int)), ("km", C.c_void_p), ("phase_nx", C.c_int), ("phase_ny", C.c_int)] + \
               [(k, C.c_double) for k in ("phase_xmin", "phase_xmax", "phase_ymin", "phase_ymax")] + \
               [("phase_z", C.POINTER(C.c_double)), ("phase_diag", C.c_void_p)] + \
               [("ep_on", C.c_int), ("box_xx", C.c_double), ("epress", C.POINTER(C.c_double)), ("epress_deriv", C.POINTER(C.c_double)),
                ("pe_nx", C.c_int), ("pe_ny", C.c_int)] + [(k, C.c_double) for k in ("pe_xmin", "pe_xmax", "pe_ymin", "pe_ymax")] + \
               [("pe_z", C.POINTER(C.c_double)), ("pe_diag", C.c_void_p)]


class OracleTTM1D:
    """the reference's 1D TTM (imd_ttm.c) on flat atom arrays; lattice arrays are views of length nx + 2 (ghost cells at both ends)"""

    def __init__(self, nx, box, nranks, vlatdim, vlatbuffer, vlatdens, fd_min_atoms, fd_n_timesteps, atomic_weight, atomic_charge,
                 r2_cut_pair, timestep, eos_cve, eos_ee, adv=True):
        L = _ttm_lib()
        ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
        L.orc_ttm1d_new.restype = C.POINTER(_Ttm1d)
        L.orc_ttm1d_new.argtypes = [C.c_int] + [C.c_double] * 3 + [C.c_int] * 3 + [C.c_double, C.c_int, C.c_int] + [C.c_double] * 4 + [C.c_void_p]
        L.orc_ttm1d_free.argtypes = [C.POINTER(_Ttm1d)]
        L.orc_ttm1d_update_fd.argtypes = [C.POINTER(_Ttm1d), C.c_int, C.c_int, dp, ip, dp, dp, ip, ip]
        L.orc_ttm1d_adv.argtypes = [C.POINTER(_Ttm1d), C.c_double, C.c_int]
        L.orc_ttm1d_cell_activation.argtypes = [C.POINTER(_Ttm1d)]
        L.orc_ttm1d_fillmesh.argtypes = [C.POINTER(_Ttm1d), C.c_int]
        L.orc_ttm1d_fill_ghosts.argtypes = [C.POINTER(_Ttm1d)]
        L.orc_ttm1d_cfl.argtypes = [C.POINTER(_Ttm1d), C.c_int, C.c_double]
        L.orc_ttm1d_diff.argtypes = [C.POINTER(_Ttm1d), C.c_double]
        L.orc_ttm1d_calc.argtypes = [C.POINTER(_Ttm1d), C.c_int, C.c_double, C.c_int, dp, ip, dp, dp, ip, ip]
        L.orc_ttm1d_move_atoms.restype = C.c_double
        L.orc_ttm1d_move_atoms.argtypes = [C.POINTER(_Ttm1d), C.c_double, C.c_int, dp, ip, dp, dp, dp]
        # a 3D oracle lattice object only to hold the two EOS tables
        one = np.array([3, 3, 3], np.int32); ext = np.array([1, 1, 1], np.int32); h = np.ones(3); pbc = np.array([1, 1, 1], np.int32)
        self._holder = L.orc_ttm_new(_ip(one), _ip(ext), _dp(h), _ip(pbc), 1, float(atomic_weight), float(atomic_charge), float(r2_cut_pair))
        for which, e in ((0, eos_cve), (1, eos_ee)):
            z = np.ascontiguousarray(e["z"], np.float64).reshape(-1)
            L.orc_ttm_set_eos(self._holder, which, len(e["x"]), len(e["y"]), float(e["x"][0]), float(e["x"][-1]), float(e["y"][0]),
                              float(e["y"][-1]), _dp(z))
        box = np.asarray(box, np.float64).reshape(3, 3)
        self._p = L.orc_ttm1d_new(int(nx), box[0, 0], box[1, 1], box[2, 2], int(nranks), int(vlatdim), int(vlatbuffer), float(vlatdens),
                                  int(fd_min_atoms), int(fd_n_timesteps), float(atomic_weight), float(atomic_charge), float(r2_cut_pair),
                                  float(timestep), C.cast(self._holder, C.c_void_p))
        self.t = self._p.contents
        self.t.adv = 1 if adv else 0
        self.timestep, self.nx, self.L = float(timestep), int(nx), L
        aa = np.ctypeslib.as_array
        n2 = self.nx + 2
        for k in ("natoms", "natoms_old", "xi", "md_temp", "source", "dens", "fd_k", "fd_g", "Z", "Ce", "ne"):
            setattr(self, k, aa(getattr(self.t, k), (n2,)))
        for k in ("fluxl", "fluxr", "natomsglobal", "xiarr"):
            setattr(self, k, aa(getattr(self.t, k), (self.nx,)))
        self.vcom = [aa(self.t.vcom[d], (self.nx,)) for d in range(3)]
        self._temp = [aa(self.t.temp[q], (n2,)) for q in (0, 1)]
        self._U = [aa(self.t.U[q], (n2,)) for q in (0, 1)]
        nv = int(vlatdim)
        self._vt = [aa(self.t.vtemp[q], (nv,)) for q in (0, 1)]
        self._vm = [aa(self.t.vmd[q], (nv,)) for q in (0, 1)]
        for k in ("vk", "vg", "vCe", "vZ", "vne"):
            setattr(self, k, aa(getattr(self.t, k), (nv,)))
        self.fdi = None

    temp = property(lambda self: self._temp[self.t.cur])
    temp2 = property(lambda self: self._temp[self.t.cur ^ 1])
    U = property(lambda self: self._U[self.t.cur])
    U2 = property(lambda self: self._U[self.t.cur ^ 1])
    vtemp = property(lambda self: self._vt[self.t.vcur])
    vmd = property(lambda self: self._vm[self.t.vcur])

    def _atoms(self, ort, sorte, impuls, masse, numneighs):
        self._a = (np.ascontiguousarray(ort, np.float64), np.ascontiguousarray(sorte, np.int32), np.ascontiguousarray(impuls, np.float64),
                   np.ascontiguousarray(masse, np.float64), np.ascontiguousarray(numneighs, np.int32))
        if self.fdi is None or len(self.fdi) != len(self._a[0]):
            self.fdi = np.full(len(self._a[0]), -1, np.int32)
        return self._a

    def update_fd(self, steps, ort, sorte, impuls, masse, numneighs):
        o, s, p, m, nn = self._atoms(ort, sorte, impuls, masse, numneighs)
        self.L.orc_ttm1d_update_fd(self._p, int(steps), len(o), _dp(o), _ip(s), _dp(p), _dp(m), _ip(nn), _ip(self.fdi))

    def tmm_init(self, I0, lam, sigma_t, t0, eps_bb, K12, phase, sigma_t1=0.0, t1=0.0, abs_threshold=20.0, laser_threshold=1e-5):
        """tmm_init (imd_tmm.c:98-191); eps_bb [rows, 3], K12 [rows, 2], phase = dict(x, y, z) like the EOS tables"""
        L = self.L
        dp = C.POINTER(C.c_double)
        L.orc_ttm1d_tmm_init.argtypes = [C.POINTER(_Ttm1d)] + [C.c_double] * 8 + [C.c_int, dp, C.c_int, dp, C.c_int, C.c_int, dp, dp]
        L.orc_ttm1d_do_tmm.argtypes = [C.POINTER(_Ttm1d), C.c_int]
        eb = np.ascontiguousarray(eps_bb, np.float64); k12 = np.ascontiguousarray(K12, np.float64)
        pz = np.ascontiguousarray(phase["z"], np.float64).reshape(-1)
        rng = np.array([phase["x"][0], phase["x"][-1], phase["y"][0], phase["y"][-1]], np.float64)
        L.orc_ttm1d_tmm_init(self._p, float(I0), float(lam), float(sigma_t), float(t0), float(sigma_t1), float(t1), float(abs_threshold),
                             float(laser_threshold), len(eb), _dp(eb), len(k12), _dp(k12), len(phase["x"]), len(phase["y"]), _dp(rng), _dp(pz))
        aa = np.ctypeslib.as_array
        for k in ("eps_re", "eps_im", "qabs"):
            setattr(self, k, aa(getattr(self.t, k), (self.nx,)))
        self.tmm_active = aa(self.t.tmm_active, (self.nx,))

    def set_epress(self, pe):
        """-DELECPRESS: the table of EOS_pe_from_r_te (Pa) as dict(x, y, z)"""
        L = self.L
        dp = C.POINTER(C.c_double)
        L.orc_ttm1d_set_epress.argtypes = [C.POINTER(_Ttm1d), C.c_int, C.c_int, dp, dp]
        L.orc_ttm1d_elec_pressure.argtypes = [C.POINTER(_Ttm1d), C.c_int]
        pz = np.ascontiguousarray(pe["z"], np.float64).reshape(-1)
        rng = np.array([pe["x"][0], pe["x"][-1], pe["y"][0], pe["y"][-1]], np.float64)
        L.orc_ttm1d_set_epress(self._p, len(pe["x"]), len(pe["y"]), _dp(rng), _dp(pz))
        aa = np.ctypeslib.as_array
        self.epress = aa(self.t.epress, (self.nx,)); self.epress_deriv = aa(self.t.epress_deriv, (self.nx,))

    def elec_pressure(self, steps): self.L.orc_ttm1d_elec_pressure(self._p, int(steps))

    def do_tmm(self, steps):
        rc = self.L.orc_ttm1d_do_tmm(self._p, int(steps))
        if rc:
            raise RuntimeError("Sample is too short for TMM")

    def adv(self, tau, steps): self.L.orc_ttm1d_adv(self._p, float(tau), int(steps))
    def cell_activation(self): self.L.orc_ttm1d_cell_activation(self._p)
    def fillmesh(self, steps): self.L.orc_ttm1d_fillmesh(self._p, int(steps))
    def fill_ghosts(self): self.L.orc_ttm1d_fill_ghosts(self._p)
    def cfl(self, steps): self.L.orc_ttm1d_cfl(self._p, int(steps), self.timestep)
    def diff(self, tau): self.L.orc_ttm1d_diff(self._p, float(tau))

    def calc(self, steps, ort, sorte, impuls, masse, numneighs):
        o, s, p, m, nn = self._atoms(ort, sorte, impuls, masse, numneighs)
        return self.L.orc_ttm1d_calc(self._p, int(steps), self.timestep, len(o), _dp(o), _ip(s), _dp(p), _dp(m), _ip(nn), _ip(self.fdi))

    def move_atoms(self, ort, sorte, impuls, kraft, masse):
        o = np.ascontiguousarray(ort, np.float64); p = np.ascontiguousarray(impuls, np.float64)
        s = np.ascontiguousarray(sorte, np.int32); k = np.ascontiguousarray(kraft, np.float64); m = np.ascontiguousarray(masse, np.float64)
        e = self.L.orc_ttm1d_move_atoms(self._p, self.timestep, len(o), _dp(o), _ip(s), _dp(p), _dp(k), _dp(m))
        return o, p, e

    def close(self):
        if self._p is not None:
            self.L.orc_ttm1d_free(self._p); self.L.orc_ttm_free(self._holder); self._p = None


# ---------------------------------------------------------------------------------------------------------------
# tricubic table look-up behind fitDL (oracle/tricub_oracle.c)
class OracleTricub:
    """do_tricubinterp on one table (imd_interpol.c:740-957); table = imd_amd.tricub.TricubTable or the same fields"""
    def __init__(self, rng, x, y, z, f):
        L = lib()
        L.orc_tricub_new.restype = C.c_void_p
        L.orc_tricub_new.argtypes = [C.c_int] * 3 + [C.c_void_p] * 5
        L.orc_tricub_free.argtypes = [C.c_void_p]
        L.orc_tricub_eval_n.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        f = np.ascontiguousarray(f, dtype=np.float64)
        a = [np.ascontiguousarray(v, dtype=np.float64) for v in (rng, x, y, z)]
        self.ymax, self.zmax = float(a[0][3]), float(a[0][5])
        self._p = L.orc_tricub_new(f.shape[0], f.shape[1], f.shape[2], *[v.ctypes.data for v in a], f.ctypes.data)

    def eval(self, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        out = np.empty(len(xyz))
        lib().orc_tricub_eval_n(self._p, len(xyz), xyz.ctypes.data, out.ctypes.data)
        return out

    def close(self):
        if self._p:
            lib().orc_tricub_free(self._p)
            self._p = None


def oracle_fitdl(tabs, temp, md_temp, dens, ne):
    """fitDL (imd_ttm_3D.c:2919-2968) for arrays of cells; tabs = five OracleTricub (DL1..DL5) -> (DL[n, 6], status[n])"""
    L = lib()
    L.orc_fitdl.restype = C.c_int
    L.orc_fitdl.argtypes = [C.c_void_p] + [C.c_double] * 4 + [C.c_void_p]
    ptrs = (C.c_void_p * 5)(*[t._p for t in tabs])
    temp, md_temp, dens, ne = (np.atleast_1d(np.asarray(v, dtype=np.float64)) for v in (temp, md_temp, dens, ne))
    DL = np.empty((len(temp), 6)); st = np.empty(len(temp), dtype=np.int32)
    row = (C.c_double * 6)()
    for q in range(len(temp)):
        st[q] = L.orc_fitdl(ptrs, temp[q], md_temp[q], dens[q], ne[q], row)
        DL[q] = row[:]
    return DL, st
