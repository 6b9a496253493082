"""Row f2 (config 5's 1D Maxwell laser source): the oracle's restatement of the reference's FDTD solver
(oracle/fdtd_oracle.c <- imd_fdtd.c:63-126,128-306,594-859) against the fields the reference's OWN do_fdtd produced
(tests/golden/fdtd_1d.npz, made by oracle/ref_probe_fdtd.c: six MD steps x 437 sub-steps on a 158-cell row with an
80-cell Al slab, PML at both ends, TF/SF source).  Same C arithmetic, same order: bit-exact."""
import numpy as np

from oracle import pyoracle as po
from tests import golden_util as gu


def oracle_from_golden(G):
    a = {k[4:]: G[k] for k in G.files if k.startswith("arg_")}
    f = po.OracleFDTD(int(a["nx"]), float(a["hx"]), float(a["I0"]), float(a["lam"]), float(a["sigma_t"]), float(a["t0"]),
                      float(a["srcx"]), int(a["bw"]), float(a["Sc"]))
    f.natoms[:] = G["cell_natoms"]
    f.DL[:] = G["cell_DL"]
    return f, a


def test_setup_matches_init_fdtd_and_init_pml():
    G = gu.load("fdtd_1d")
    f, a = oracle_from_golden(G)
    assert f.f.srcx_int == int(G["srcx_int"]) and f.f.containssrc == int(G["containssrc"]) and f.f.bw == int(G["bw"])
    for k, v in (("dt", "dt_FDTD"), ("fd_dx", "fd_dx"), ("E0", "E0"), ("Nlambdax", "Nlambdax"), ("Imp0", "Imp0"),
                 ("omega_laser", "omega_laser"), ("sigma_t_squared", "sigma_t_squared")):
        assert getattr(f.f, k) == float(G[v]), k
    assert np.array_equal(f.sigmay, G["init_sigmay"])            # the PML profile, cell by cell
    assert (G["init_sigmay"] > 0).sum() == 2 * int(G["bw"]) + 1
    f.close()


def test_fields_equal_the_reference_solver():
    G = gu.load("fdtd_1d")
    f, a = oracle_from_golden(G)
    t = 0.0
    for s, (nsub_ref, act_ref) in enumerate(zip(G["md_nsub"], G["md_active"])):
        nsub, tau = f.begin_md_step(t, float(a["timestep"]))
        assert nsub == int(nsub_ref) and f.f.laser_active == int(act_ref)
        for sub in range(nsub):
            if f.f.laser_active:
                f.step(t)
            t += tau
            tag = "s0_sub%d" % sub
            if s == 0 and tag + "_Ezx" in G:
                for k in ("Ezx", "Hy", "Jzx", "Jlzx", "Pzx", "source"):
                    assert np.array_equal(getattr(f, k), G["%s_%s" % (tag, k)]), (tag, k)
        tag = "md%d" % s
        assert t == float(G[tag + "_t_SI"]) and f.f.laser_timefun == float(G[tag + "_laser_timefun"])
        for k in ("Ezx", "Hy", "Jzx", "Jlzx", "Pzx", "source", "Jzx_old", "Jlzx_old"):
            assert np.array_equal(getattr(f, k), G["%s_%s" % (tag, k)]), (tag, k)
    # the run did something: the pulse entered the slab and deposited power there, nothing in the vacuum cells
    assert G["md2_source"].max() > 1e-3 and np.all(G["md2_source"][G["cell_natoms"] < 5] == 0.0)
    f.close()
