"""Potential tables: host-side mirror of IMD's ``pot_table_t`` interface.

The device kernels consume exactly what the reference keeps in memory
(reference types.h:537-549): per column ``begin/end/step/invstep/len`` and a
column-interleaved value table ``table[k*ncols+col]`` with two extrapolated
rows appended (reference imd_potential.c:1260-1276), equidistant in r**2 for
radial tables and in rho for the embedding table.

This module is host logic (file formats, analytic table creation); evaluation
happens on the GPU (imd_amd/csrc/imdgpu.hip) and, for tests, in oracle/.
"""
from dataclasses import dataclass

import numpy as np

POT_TAIL = 0.05          # reference config.h:124


@dataclass
class PotTable:
    """Same fields, same meaning as the reference's ``pot_table_t``."""
    begin: np.ndarray
    end: np.ndarray
    step: np.ndarray
    invstep: np.ndarray
    len: np.ndarray
    ncols: int
    maxsteps: int
    table: np.ndarray        # flat, (maxsteps+2)*ncols

    def as_dict(self):
        return dict(begin=self.begin, end=self.end, step=self.step, invstep=self.invstep,
                    len=self.len, ncols=self.ncols, maxsteps=self.maxsteps, table=self.table)

    @staticmethod
    def from_dict(d):
        return PotTable(np.ascontiguousarray(d["begin"], np.float64), np.ascontiguousarray(d["end"], np.float64),
                        np.ascontiguousarray(d["step"], np.float64), np.ascontiguousarray(d["invstep"], np.float64),
                        np.ascontiguousarray(d["len"], np.int32), int(d["ncols"]), int(d["maxsteps"]),
                        np.ascontiguousarray(d["table"], np.float64))


def init_threepoint(table, ncols, lens):
    """Append two rows continuing the last parabola (imd_potential.c:1260-1276)."""
    t = table.reshape(-1, ncols)
    for col in range(ncols):
        n = int(lens[col])
        a, b, c = t[n - 1, col], t[n - 2, col], t[n - 3, col]
        t[n, col] = 3 * a - 3 * b + c
        t[n + 1, col] = 6 * a - 8 * b + 3 * c


PSTEP = 50          # config.h:260


def _tokens(path):
    header, body = [], []
    with open(path) as f:
        for line in f:
            if line.startswith("#"):
                header.append(line)
            else:
                body.append(line)
    return header, " ".join(body).split()


def read_pot_table(path, ncols, radial):
    """Read an IMD potential file (format 1 or 2, imd_potential.c:161-466).

    Returns (PotTable, cellsz) where cellsz is max(end) over columns for radial
    tables (the r**2 the cell size derives from), else 0.
    """
    header, tok = _tokens(path)
    fmt = 2 if ncols else 1
    have_fmt = False
    for h in header:
        if h[1] == "F":
            p = h[2:].split()
            fmt, size = int(p[0]), int(p[1])
            if size != ncols:
                raise ValueError("Wrong number of data columns in file %s: should be %d, is %d" % (path, ncols, size))
            if fmt not in (1, 2):
                raise ValueError("Unrecognized format specified for file %s" % path)
            have_fmt = True
    if header and not have_fmt:
        raise ValueError("Format not specified in header of file %s" % path)
    vals = np.array(tok, dtype=np.float64)
    cellsz = 0.0
    if fmt == 2:
        info = vals[: 3 * ncols].reshape(ncols, 3)
        begin, end, step = info[:, 0].copy(), info[:, 1].copy(), info[:, 2].copy()
        numstep = 1 + (end - begin) / step
        lens = (numstep + 0.49).astype(np.int32)
        maxsteps = int(lens.max())
        table = np.zeros((maxsteps + 2, ncols))
        off = 3 * ncols
        for i in range(ncols):
            if off + lens[i] > len(vals):
                raise ValueError("wrong format in file %s." % path)
            table[: lens[i], i] = vals[off: off + lens[i]]
            off += lens[i]
        if radial:
            for i in range(ncols):
                delta = table[lens[i] - 1, i]
                if delta != 0.0:
                    table[: lens[i], i] -= delta
            cellsz = float(end.max())
    else:
        rows = vals.reshape(-1, ncols + 1)
        npot = rows.shape[0]
        r2_start, r2 = rows[0, 0], rows[-1, 0]
        r2_step = (r2 - r2_start) / (npot - 1)
        maxsteps = max(PSTEP, -(-npot // PSTEP) * PSTEP)       # allocated in blocks of PSTEP rows (:307-323)
        table = np.zeros((maxsteps + 2, ncols))
        table[:npot] = rows[:, 1:]
        begin = np.full(ncols, r2_start); step = np.full(ncols, r2_step)
        end = np.zeros(ncols); lens = np.zeros(ncols, np.int32)
        for i in range(ncols):
            nz = np.nonzero(rows[:, 1 + i])[0]
            if len(nz):
                end[i] = rows[nz[-1], 0]; lens[i] = nz[-1] + 1
            if radial:
                delta = table[npot - 1, i]
                if delta != 0.0:
                    table[:npot, i] -= delta
        if radial:
            cellsz = float(end.max())
    pt = PotTable(begin, end, step, 1.0 / step, lens.astype(np.int32), ncols, maxsteps, table.reshape(-1).copy())
    init_threepoint(pt.table, ncols, pt.len)
    return pt, cellsz


def write_pot_table2(path, begin, end, step, columns):
    """Write format 2 (``#F 2 ncols`` / ``#E`` / info lines / values per column)."""
    ncols = len(columns)
    with open(path, "w") as f:
        f.write("#F 2 %d\n#E\n" % ncols)
        for i in range(ncols):
            f.write("%.16e %.16e %.16e\n" % (begin[i], end[i], step[i]))
        f.write("\n")
        for col in columns:
            for v in col:
                f.write("%.16e\n" % v)
            f.write("\n")


def write_pot_table1(path, r2, columns):
    """Write format 1 (``#F 1 ncols`` / ``#E`` / one line `r2 v_0 v_1 ...` per sample, r2 equidistant)."""
    cols = np.asarray(columns, dtype=np.float64)
    with open(path, "w") as f:
        f.write("#F 1 %d\n#E\n" % cols.shape[0])
        for k, x in enumerate(r2):
            f.write(" ".join("%.16e" % v for v in [x] + list(cols[:, k])) + "\n")


def pair_int_lj(eps, sigma, r2):
    """IMD's Lennard-Jones form eps*(s^12 - 2 s^6) (imd_potential.c:1437-1449)."""
    s2 = sigma * sigma / r2
    s6 = s2 * s2 * s2
    s12 = s6 * s6
    return eps * (s12 - 2.0 * s6), -12.0 * eps / r2 * (s12 - s6)


def create_lj_table(eps, sigma, r_cut, r_begin=0.0, pot_res=1000):
    """Tabulate LJ in r**2 the way create_pot_table does for one atom type
    (imd_potential.c:476-705, shift/tail from init_pre_pot :853-872)."""
    if r_begin == 0.0:
        r_begin = 0.1 * sigma
    r2_cut = r_cut * r_cut
    shift, tmp = pair_int_lj(eps, sigma, (1.0 - POT_TAIL) * r2_cut)
    shift += 0.25 * tmp * POT_TAIL * r2_cut
    aaa = -0.25 * tmp / (POT_TAIL * r2_cut)
    r2_begin, r2_end = r_begin * r_begin, r2_cut
    r2_step = (r2_end - r2_begin) / (pot_res - 1)
    table = np.zeros((pot_res + 2, 1))
    for n in range(pot_res):
        r2 = r2_begin + n * r2_step
        val = 0.0
        if r2 < (1.0 - POT_TAIL) * r2_cut:
            pot, _ = pair_int_lj(eps, sigma, r2)
            val += pot - shift
        elif r2 <= r2_cut:
            val += aaa * (r2_cut - r2) ** 2
        table[n, 0] = val
    pt = PotTable(np.array([r2_begin]), np.array([r2_end]), np.array([r2_step]), np.array([1.0 / r2_step]),
                  np.array([pot_res], np.int32), 1, pot_res, table.reshape(-1).copy())
    init_threepoint(pt.table, 1, pt.len)
    return pt, r2_end


# ---------------------------------------------------------------------------
# synthetic EAM for Al-like FCC; no real Al tables ship
# with the reference, so every number in this repo is for these tables.
# ---------------------------------------------------------------------------

def _cut(r, rc=6.5, w=0.6):
    x = (r - rc) / w
    return x ** 4 / (1.0 + x ** 4)


def synthetic_eam_functions():
    phi = lambda r: 0.27 * (np.exp(-2 * 1.16 * (r - 2.86)) - 2 * np.exp(-1.16 * (r - 2.86))) * _cut(r)
    rho = lambda r: np.exp(-3.0 * (r / 2.86 - 1.0)) * _cut(r)
    emb = lambda p: -1.6 * np.sqrt(p) + 0.02 * p * p
    return phi, rho, emb


# grids of the one-type synthetic EAM: (r_begin, r_cut, rows) of phi and of rho, (rho_max, rows) of F.
# "shared": phi and rho on ONE r**2 grid (files written by one tool); "distinct": three files with three headers --
# different begin, end, step and row count each (hand-made EAM files; the manual's phi/rho/f.zhakov.pt are three files)
EAM_GRIDS = {
    "shared": dict(phi=(1.5, 6.5, 4000), rho=(1.5, 6.5, 4000), emb=(40.0, 4000)),
    "distinct": dict(phi=(1.5, 6.5, 4000), rho=(1.7, 6.3, 3500), emb=(45.0, 3000)),
}


def _eam_columns(grids):
    """analytic functions sampled on the grids of EAM_GRIDS[grids]: [(begin, end, step, values)] for phi, rho, F"""
    phi, rho, emb = synthetic_eam_functions()
    g = EAM_GRIDS[grids]
    out = []
    for fn, (rb, rc, n) in ((phi, g["phi"]), (rho, g["rho"])):
        b2, e2 = rb ** 2, rc ** 2
        step = (e2 - b2) / (n - 1)
        r = np.sqrt(b2 + step * np.arange(n))
        out.append((b2, e2, step, fn(r if rc == 6.5 else r * 6.5 / rc)))     # the analytic cut-off sits at 6.5: squeezed to rc
    rho_max, n = g["emb"]
    ps = rho_max / (n - 1)
    out.append((0.0, rho_max, ps, emb(ps * np.arange(n))))
    return out


def synthetic_eam_tables(n=4000, r_begin=1.5, r_cut=6.5, rho_max=40.0, grids="shared"):
    """(pair_pot, rho_h_tab, embed_pot, cellsz) tabulated like the reference would
    hold them after read_pot_table (shifted to zero at r_cut, 2 pad rows)."""
    if grids == "shared" and (n, r_begin, r_cut, rho_max) != (4000, 1.5, 6.5, 40.0):
        cols = _eam_columns_custom(n, r_begin, r_cut, rho_max)
    else:
        cols = _eam_columns(grids)
    out = []
    for which, (b, e, step, v) in enumerate(cols):
        m = len(v)
        t = np.zeros((m + 2, 1))
        t[:m, 0] = v - v[-1] if (which < 2 and v[-1] != 0.0) else v
        pt = PotTable(np.array([b]), np.array([e]), np.array([step]), np.array([1.0 / step]),
                      np.array([m], np.int32), 1, m, t.reshape(-1).copy())
        init_threepoint(pt.table, 1, pt.len)
        out.append(pt)
    return out[0], out[1], out[2], max(cols[0][1], cols[1][1])


def _eam_columns_custom(n, r_begin, r_cut, rho_max):
    phi, rho, emb = synthetic_eam_functions()
    b2, e2 = r_begin ** 2, r_cut ** 2
    step = (e2 - b2) / (n - 1)
    r = np.sqrt(b2 + step * np.arange(n))
    ps = rho_max / (n - 1)
    return [(b2, e2, step, phi(r)), (b2, e2, step, rho(r)), (0.0, rho_max, ps, emb(ps * np.arange(n)))]


def write_synthetic_eam_files(dirname, n=4000, r_begin=1.5, r_cut=6.5, rho_max=40.0, grids="shared"):
    """phi.pt / rho.pt / F.pt in IMD format 2 for the reference binary."""
    import os
    if grids == "shared" and (n, r_begin, r_cut, rho_max) != (4000, 1.5, 6.5, 40.0):
        cols = _eam_columns_custom(n, r_begin, r_cut, rho_max)
    else:
        cols = _eam_columns(grids)
    for fn, (b, e, step, v) in zip(("phi.pt", "rho.pt", "F.pt"), cols):
        write_pot_table2(os.path.join(dirname, fn), [b], [e], [step], [v])


def write_synthetic_eam_files_2types(dirname, rows=1.0):
    """Two atom types (A = Al-like, B heavier): pair and density tables with ntypes^2 = 4 columns
    (col = type_i * ntypes + type_j, types.h:537-549), each with ITS OWN begin / end / step / length, the
    embedding table with 2 columns.  Exercises the per-column indexing of PAIR_INT2 / VAL_FUNC2 / DERIV_FUNC2.
    rows < 1 shrinks every column (coarse tables that fit the LDS next to a tile)."""
    import os
    phi, rho, emb = synthetic_eam_functions()
    spec = {0: (1.5, 6.5, 4000, 1.00, 1.00), 1: (1.6, 6.2, 3500, 0.85, 1.03), 2: (1.6, 6.2, 3500, 0.85, 1.03),
            3: (1.7, 6.4, 3000, 1.25, 0.97)}             # r_begin, r_cut, n, amplitude, radial stretch
    rspec = {0: (1.5, 6.5, 4000, 1.00), 1: (1.5, 6.3, 3800, 0.80), 2: (1.6, 6.1, 3600, 1.30), 3: (1.5, 6.4, 3900, 1.10)}
    def cut_at(fn, rc):
        return lambda r: fn(r * 6.5 / rc)                # the analytic cut-off of fn sits at 6.5: squeeze it to rc
    b, e, st, cols = [], [], [], []
    for c in range(4):
        rb, rc, n, amp, stretch = spec[c]
        n = int(n * rows)
        b2, e2 = rb * rb, rc * rc
        step = (e2 - b2) / (n - 1)
        r = np.sqrt(b2 + step * np.arange(n))
        b.append(b2); e.append(e2); st.append(step); cols.append(amp * cut_at(phi, rc)(r * stretch / stretch))
    write_pot_table2(os.path.join(dirname, "phi.pt"), b, e, st, cols)
    b, e, st, cols = [], [], [], []
    for c in range(4):
        rb, rc, n, amp = rspec[c]
        n = int(n * rows)
        b2, e2 = rb * rb, rc * rc
        step = (e2 - b2) / (n - 1)
        r = np.sqrt(b2 + step * np.arange(n))
        b.append(b2); e.append(e2); st.append(step); cols.append(amp * cut_at(rho, rc)(r))
    write_pot_table2(os.path.join(dirname, "rho.pt"), b, e, st, cols)
    ne = [int(4000 * rows), int(3500 * rows)]
    ps = [40.0 / (ne[0] - 1), 55.0 / (ne[1] - 1)]
    write_pot_table2(os.path.join(dirname, "F.pt"), [0.0, 0.0], [40.0, 55.0], ps,
                     [emb(ps[0] * np.arange(ne[0])), 1.15 * emb(0.9 * ps[1] * np.arange(ne[1]))])


def write_synthetic_eam_files_ntypes(dirname, ntypes, rows=1.0):
    """ntypes atom types (3 or 4 here; 2 has its own writer above): pair and density tables with ntypes^2 columns, every column with ITS
    OWN begin / end / step / length (col = type_i * ntypes + type_j, types.h:537-549); the pair columns are symmetric in (i, j) like a
    physical pair potential, the density columns are all different; ntypes embedding columns."""
    import os
    phi, rho, emb = synthetic_eam_functions()
    def cut_at(fn, rc):
        return lambda r: fn(r * 6.5 / rc)
    def columns(fn, sym):
        b, e, st, cols = [], [], [], []
        for i in range(ntypes):
            for j in range(ntypes):
                lo, hi = (min(i, j), max(i, j)) if sym else (i, j)
                k = lo * ntypes + hi
                rb, rc = 1.5 + 0.05 * ((3 * k) % 5), 6.5 - 0.07 * ((5 * k + 2) % 7)
                n = int((4000 - 150 * ((7 * k) % 6)) * rows)
                amp = 1.0 + 0.11 * ((k * 3) % 5) - 0.2
                b2, e2 = rb * rb, rc * rc
                step = (e2 - b2) / (n - 1)
                r = np.sqrt(b2 + step * np.arange(n))
                b.append(b2); e.append(e2); st.append(step); cols.append(amp * cut_at(fn, rc)(r))
        return b, e, st, cols
    write_pot_table2(os.path.join(dirname, "phi.pt"), *columns(phi, True))
    write_pot_table2(os.path.join(dirname, "rho.pt"), *columns(rho, False))
    ne = [int((4000 - 300 * t) * rows) for t in range(ntypes)]
    hi = [40.0 + 6.0 * t for t in range(ntypes)]
    ps = [hi[t] / (ne[t] - 1) for t in range(ntypes)]
    write_pot_table2(os.path.join(dirname, "F.pt"), [0.0] * ntypes, hi, ps,
                     [(1.0 + 0.08 * t) * emb((1.0 - 0.04 * t) * ps[t] * np.arange(ne[t])) for t in range(ntypes)])
