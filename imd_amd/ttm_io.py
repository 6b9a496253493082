"""`.ttm` lattice files of the 3D two-temperature model: writer and reader.

ttm_writeout (imd_ttm_3D.c:1836-2118, the plain-text branch) prints one line per inner FD cell of the GLOBAL lattice,
x outermost, 0-based indices:

    #x y z natoms temp md_temp U xi source dens vx vy vz fd_k fd_g Z proc Ce
    %d %d %d %d %e %e %e %e %e %e %e %e %e %e %e %e %d %f

(file name `<outfiles>.<n>.ttm`, n = steps / ttm_int, imd_main_3d.c:735).  ttm_read (:2127-2330) takes the same file for
a restart.  Arrays here are the SoA lattice arrays of imdgpu_ttm_download *including* the ghost shell, as everywhere
in this package.  Pinned by tests/test_config_io.py against a file the reference wrote (tests/golden/ttm_file.npz)."""
import numpy as np

HEADER = "#x y z natoms temp md_temp U xi source dens vx vy vz fd_k fd_g Z proc Ce\n"
_FLOATS = ("temp", "md_temp", "U", "xi", "source", "dens", "vx", "vy", "vz", "fd_k", "fd_g", "Z")


def write_ttm(path, lat, proc=0):
    """lat: dict of arrays over the lattice incl. ghosts (natoms, temp, md_temp, U, xi, source, dens, vcom[...,3],
    fd_k, fd_g, Z, Ce); proc: rank id printed in the `proc` column (scalar or array)."""
    nat = np.asarray(lat["natoms"])
    inner = (slice(1, -1),) * 3
    dim = tuple(d - 2 for d in nat.shape)
    v = np.asarray(lat["vcom"])
    cols = dict(vx=v[..., 0], vy=v[..., 1], vz=v[..., 2])
    f = {k: np.asarray(cols[k] if k in cols else lat[k], dtype=np.float64)[inner].reshape(-1) for k in _FLOATS}
    ce = np.asarray(lat["Ce"], dtype=np.float64)[inner].reshape(-1)
    na = nat[inner].reshape(-1)
    pr = np.broadcast_to(np.asarray(proc), nat.shape)[inner].reshape(-1) if np.ndim(proc) else np.full(na.shape, int(proc))
    i, j, k = (a.reshape(-1) for a in np.meshgrid(*(np.arange(d) for d in dim), indexing="ij"))
    fmt = "%d %d %d %d" + " %e" * len(_FLOATS) + " %d %f\n"
    with open(path, "w") as out:
        out.write(HEADER)
        out.write("".join(fmt % ((i[q], j[q], k[q], na[q]) + tuple(f[n][q] for n in _FLOATS) + (pr[q], ce[q]))
                          for q in range(len(na))))


def read_ttm(path):
    """-> dict of arrays over the INNER global lattice [nx, ny, nz] (natoms, temp, ..., vcom[...,3], proc, Ce)"""
    a = np.loadtxt(path, comments="#", ndmin=2)
    idx = a[:, :3].astype(int)
    dim = tuple(idx.max(axis=0) + 1)
    if len(a) != int(np.prod(dim)):
        raise ValueError("%s: %d lines for a %s lattice" % (path, len(a), "x".join(map(str, dim))))
    out = {"natoms": np.zeros(dim, np.int32), "proc": np.zeros(dim, np.int32), "Ce": np.zeros(dim)}
    sel = (idx[:, 0], idx[:, 1], idx[:, 2])
    out["natoms"][sel] = a[:, 3].astype(np.int32)
    vals = {n: a[:, 4 + q] for q, n in enumerate(_FLOATS)}
    for n in _FLOATS:
        if n in ("vx", "vy", "vz"):
            continue
        out[n] = np.zeros(dim); out[n][sel] = vals[n]
    out["vcom"] = np.zeros(dim + (3,))
    for q, n in enumerate(("vx", "vy", "vz")):
        out["vcom"][..., q][sel] = vals[n]
    out["proc"][sel] = a[:, 16].astype(np.int32); out["Ce"][sel] = a[:, 17]
    return out
