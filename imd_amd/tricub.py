"""Tables over (x, y, z) in the reference's text layout for `read_tricub_interp` (imd_interpol.c:959-1084), the format of
fitDL's DL1.txt .. DL5.txt (imd_ttm_3D.c:1245-1249): line 1 `is js ks`, line 2 `xmin xmax ymin ymax zmin zmax`, then
is*js*ks lines `x y z f` with x slowest and z fastest.  The node coordinates are the ones of the LAST line read for each
index, as the reference keeps them (x[i] from the last line of block i, y[j] from the last k of the last i, ...)."""
import numpy as np


class TricubTable:
    def __init__(self, rng, x, y, z, f):
        self.range = np.asarray(rng, dtype=np.float64)            # xmin xmax ymin ymax zmin zmax
        self.x, self.y, self.z = (np.ascontiguousarray(v, dtype=np.float64) for v in (x, y, z))
        self.f = np.ascontiguousarray(f, dtype=np.float64)        # f[i, j, k]
        if self.f.shape != (len(self.x), len(self.y), len(self.z)):
            raise ValueError("tricubic table: f has shape %s for %d x %d x %d nodes" % (self.f.shape, len(self.x), len(self.y), len(self.z)))

    @property
    def dims(self):
        return self.f.shape


def write_table(path, tab):
    is_, js, ks = tab.dims
    with open(path, "w") as fh:
        fh.write("%d %d %d\n" % (is_, js, ks))
        fh.write(" ".join("%.17g" % v for v in tab.range) + "\n")
        for i in range(is_):
            for j in range(js):
                for k in range(ks):
                    fh.write("%.17g %.17g %.17g %.17g\n" % (tab.x[i], tab.y[j], tab.z[k], tab.f[i, j, k]))


def read_table(path):
    with open(path) as fh:
        tok = fh.read().split()
    is_, js, ks = (int(t) for t in tok[:3])
    rng = [float(t) for t in tok[3:9]]
    body = np.array(tok[9:9 + 4 * is_ * js * ks], dtype=np.float64)
    if body.size != 4 * is_ * js * ks:
        raise ValueError("%s: %d values for a %d x %d x %d table" % (path, body.size, is_, js, ks))
    body = body.reshape(is_, js, ks, 4)
    # read_tricub_interp overwrites z[k] at every line, y[j] after every k loop, x[i] after every j loop: the last wins
    return TricubTable(rng, body[:, -1, -1, 0], body[-1, :, -1, 1], body[-1, -1, :, 2], body[..., 3])


def synthetic_dl_tables(dens_lo=1500.0, dens_hi=4000.0, base=(2.73, 1.1174e+15, 7.6595e+15, 2.4024e+15, 4.5199e+14), n=(9, 24, 8)):
    """Five stand-ins for DL1..DL5.txt (not part of the reference tree) over (rho, lg Te, lg Ti), nodes at the cell centres:
    fitDL's constants for Al (imd_ttm_3D.c:2921-2928; DL2..5 in eV, the unit fitDL divides by hbar) bent by up to tens of
    per cent with density and both temperatures.  Synthetic data for tests and bench.py, not material data."""
    rng = (dens_lo, dens_hi, -1.6, 1.5, -1.6, 0.5)
    ax = [rng[2 * a] + (np.arange(n[a]) + 0.5) * (rng[2 * a + 1] - rng[2 * a]) / n[a] for a in range(3)]
    X, Y, Z = np.meshgrid(*ax, indexing="ij")
    u = (X - dens_lo) / (dens_hi - dens_lo)
    hb = 6.582119487171653e-16
    val = [v * (1.0 if q == 0 else hb) for q, v in enumerate(base)]
    bend = [1.0 + 0.10 * u - 0.04 * Y + 0.02 * Z, 1.0 + 0.5 * (Y + 1.6) + 0.2 * (Z + 1.6) + 0.1 * u, 1.0 - 0.05 * Y + 0.1 * u,
            1.0 + 0.03 * Y * Z, 1.0 + 0.3 * (Y + 1.6) - 0.05 * u]
    return [TricubTable(rng, ax[0], ax[1], ax[2], b * w) for b, w in zip(val, bend)]
