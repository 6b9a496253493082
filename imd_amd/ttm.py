"""TTM helpers (host logic): EOS table files and the synthetic EOS of the bench.

The reference reads two tables in the working directory (imd_ttm_3D.c:1236-1237,
format of nn_read_table, imd_interpol.c:1174-1232): ``is js`` / ``xmin xmax ymin
ymax`` / is*js lines ``x y z`` with x = density [kg/m^3] the slow index.
The real Al tables are not part of the reference tree; everything here uses
synthetic tables and says so.
"""
import numpy as np


def read_eos_table(path):
    with open(path) as f:
        tok = f.read().split()
    nx, ny = int(tok[0]), int(tok[1])
    vals = np.array(tok[6:6 + 3 * nx * ny], dtype=np.float64).reshape(nx, ny, 3)
    x, y = vals[:, 0, 0].copy(), vals[0, :, 1].copy()
    if not (np.allclose(np.diff(x), np.diff(x)[0]) and np.allclose(np.diff(y), np.diff(y)[0])):
        raise ValueError("%s: table is not on a regular grid" % path)
    return dict(x=x, y=y, z=vals[:, :, 2].copy())


def write_eos_table(path, eos):
    x, y, z = eos["x"], eos["y"], eos["z"]
    with open(path, "w") as f:
        f.write("%d %d\n%.16e %.16e %.16e %.16e\n" % (len(x), len(y), x[0], x[-1], y[0], y[-1]))
        for i in range(len(x)):
            for j in range(len(y)):
                f.write("%.16e %.16e %.16e\n" % (x[i], y[j], z[i, j]))


def synthetic_eos(nr=24, nt=40):
    """Planar synthetic EOS: cve(rho,T) = 0.05*T J/(kg K)  (C_e = gamma*T with
    gamma = 135 J/(m^3 K^2) at 2700 kg/m^3) and ee(rho, sqrt T) = 4000*sqrt(T) J/kg.
    Planar in the table's own arguments, so any linear interpolation rule (the
    reference's Delaunay/lpi or this package's bilinear rule) reproduces it."""
    r = np.linspace(500.0, 4500.0, nr)
    t = np.linspace(1.0, 60000.0, nt)
    ts = np.linspace(1.0, 245.0, nt)
    cve = dict(x=r, y=t, z=np.outer(np.ones(nr), 0.05 * t))
    ee = dict(x=r, y=ts, z=np.outer(np.ones(nr), 4000.0 * ts))
    return cve, ee


def attach_default(sysm, fd_ext=(2, 2, 2), diff_substeps=100, fd_min_atoms=5):
    """3D TTM as in BASELINE config 3: Al, fd_ext 2 2 2, fixed diff_substeps, no laser."""
    cve, ee = synthetic_eos()
    sysm.init_ttm(fd_ext, fd_min_atoms, diff_substeps, 26.9815, 13.0, cve, ee, omega_laser=0.0)


# ---------------------------------------------------------------------------
# laser set-up on the host (imd_laser.c): the parameters laser_rescale_ttm reads
# ---------------------------------------------------------------------------

def get_surface(x, box_xx):
    """laser_offset as get_surface() finds it (imd_laser.c:117-282): histogram of the atoms' x in 2.5 A bins (strict
    inequalities on both bin edges, :162), the sample's right end = outermost occupied bin, its surface = the bin after
    the first two adjacent empty bins met when walking left from there, nudged inwards past bins with < 500 atoms;
    returns (leftside + 0.5) * 2.5."""
    deltax = 2.5
    nd = int(box_xx / deltax)
    x = np.asarray(x, dtype=np.float64)
    dens = np.zeros(nd, dtype=np.int64)
    for l in range(nd):
        dens[l] = int(((x > l * deltax) & (x < (l + 1) * deltax)).sum())
    rightside, leftside = nd, 0
    for l in range(nd - 1, 0, -1):
        if dens[l] == 0 and dens[l - 1] != 0:
            rightside = l - 1
            break
    l = rightside
    while l > 0:
        if dens[l] == 0 and dens[l - 1] == 0:
            break
        l -= 1
    leftside = l + 1
    if dens[leftside] < 500:
        leftside = l + 3 if dens[leftside + 1] < 500 else l + 2
    return (leftside + 0.5) * deltax


def init_laser(mu, sigma_e, sigma_t, t_0, direction=(1, 0, 0), offset=0.0):
    """The globals after init_laser (imd_laser.c:283-327, no LASERYZ) as keyword arguments of
    GpuSystem.set_laser / laser_rescale_ttm: p_peak = mu * sigma_e / sigma_t / sqrt(2 pi), sigma_t_squared."""
    return dict(direction=tuple(int(v) for v in direction), offset=float(offset), mu=float(mu),
                p_peak=mu * sigma_e / sigma_t / np.sqrt(2 * np.pi), t_0=float(t_0), sigma_t_squared=sigma_t * sigma_t)
