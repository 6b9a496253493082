"""Host side of the transfer-matrix laser source of the 1D two-temperature model (imd_tmm.c): the two input files
`tmm_init` reads (`../alu_eps_bb.dat`: lambda [um], Re, Im of the interband permittivity, rows ascending in lambda;
`../K12.dat`: K1, K2 of the hot-plasma permittivity at nu = (row + 1) / 30, 1500 rows; `tmm_read_arr`, imd_tmm.c:592-655)
and synthetic stand-ins for them -- the real files are not part of the reference tree."""
import numpy as np


def read_arr(path, cols):
    """tmm_read_arr: as many rows as the file has newline characters, `cols` numbers per row"""
    with open(path) as fh:
        txt = fh.read()
    rows = txt.count("\n")
    a = np.array(txt.split()[:rows * cols], dtype=np.float64)
    if a.size != rows * cols:
        raise ValueError("%s: %d numbers for %d rows of %d" % (path, a.size, rows, cols))
    return a.reshape(rows, cols)


def write_arr(path, a):
    with open(path, "w") as fh:
        for row in np.asarray(a, dtype=np.float64):
            fh.write(" ".join("%.17g" % v for v in row) + "\n")


def synthetic_eps_bb():
    """an interband bump near 0.83 um (Al's parallel-band transition) on 0.2 .. 2.0 um; synthetic, not material data"""
    lam = np.arange(0.2, 2.0001, 0.05)
    x = (lam - 0.83) / 0.12
    return np.stack([lam, 4.0 + 18.0 * x / (1.0 + x * x), 2.0 + 36.0 / (1.0 + x * x)], axis=1)


def synthetic_K12():
    """smooth stand-ins with the limits do_tmm's closed forms use outside the table (1 below nu = 1/30; 315/8/nu^2 and
    6/nu^2 beyond 50, imd_tmm.c:662-695)"""
    nu = (np.arange(1500) + 1.0) / 30.0
    return np.stack([1.0 / (1.0 + 8.0 / 315.0 * nu * nu), 1.0 / (1.0 + nu * nu / 6.0)], axis=1)
