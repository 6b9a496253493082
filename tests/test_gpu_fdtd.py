"""GPU parity for BASELINE config 5's laser source (run with -m gpu): the 1D Maxwell solver on a one-row lattice
(imdgpu_ttm_fdtd_*; imd_fdtd.c:63-126,128-306,594-859) against

 (a) the fields the reference's OWN do_fdtd produced (tests/golden/fdtd_1d.npz, oracle/ref_probe_fdtd.c), and
 (b) the solver fused with the diffusion sub-steps (calc_ttm's loop, imd_ttm_3D.c:96-122,163-179: per FDTD sub-step
     do_fdtd, then do_DIFF + do_FILLMESH) against the CPU oracle running the same sequence function by function.

Tolerance: 1e-9 of the field's maximum (device sin/exp differ from glibc in the last bit; the scheme is linear and
stable, so the differences stay at rounding level over the 2 600 sub-steps of the fixture); integers exact.  The same bar covers
the three places where the fused loop multiplies by a reciprocal hoisted out of the sub-step loop instead of dividing inside it
(1 / natoms, 1 / (Tl rho), 1 / dx in k_ttm_fdtd_chain: one rounding each, ~1e-16 relative per sub-step)."""
import numpy as np
import pytest

from imd_amd import lattice, potential, tricub, ttm as ttm_mod
from imd_amd.system import GpuSystem
from oracle import pyoracle as po
from tests import golden_util as gu
from tests.test_oracle_fdtd_vs_reference import oracle_from_golden

pytestmark = pytest.mark.gpu


def _row_system(nxc, hx, atoms=None, hy=21.0, count_ttm=True):
    """a box of nxc x 3 x 3 MD cells (hx x hy/3 x hy/3), open in x, with ONE row of FD cells (fd_ext 1 3 3)"""
    pair, rho, emb, r2cut = potential.synthetic_eam_tables()
    margin = 0.4
    cellsz = (np.sqrt(r2cut) + margin) ** 2
    box = np.diag([nxc * hx, hy, hy])
    s = GpuSystem(box, (0, 1, 1), (nxc + 2, 5, 5), 1, cellsz, margin, 0.1, count_ttm_neighbours=count_ttm)
    s.upload_tables(pair, rho, emb)
    if atoms is None:
        s.upload_atoms(np.zeros((0, 3)), np.zeros((0, 3)), 26.9815)
    else:
        s.upload_atoms(atoms[0], atoms[1], 26.9815)
    cve, ee = ttm_mod.synthetic_eos()
    s.init_ttm((1, 3, 3), 5, 100, 26.9815, 13.0, cve, ee, omega_laser=0.0)
    assert s.fd_dim == (nxc + 2, 3, 3) and abs(s.fd_h[0] - hx) < 1e-12
    return s, (pair, cellsz, margin, box)


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_solver_vs_reference_fields():
    G = gu.load("fdtd_1d")
    a = {k[4:]: G[k] for k in G.files if k.startswith("arg_")}
    nx = int(a["nx"])
    s, _ = _row_system(nx - 2, float(a["hx"]))
    # the cells' material state of the fixture: natoms into the lattice row (i, 1, 1), per-cell Drude-Lorentz parameters
    nat = np.zeros(s.fd_dim, np.int32)
    nat[:, 1, 1] = G["cell_natoms"]
    s.ttm_upload(natoms=nat)
    s.init_fdtd(float(a["I0"]), float(a["lam"]), float(a["sigma_t"]), float(a["t0"]), float(a["srcx"]), bw=int(a["bw"]), Sc=float(a["Sc"]))
    s.fdtd_set_dl(G["cell_DL"])
    info = s.fdtd_info()
    assert info["srcx_int"] == int(G["srcx_int"]) and info["dt_FDTD"] == float(G["dt_FDTD"]) and info["E0"] == float(G["E0"])
    assert np.array_equal(s.fdtd_fields()["sigmay"], G["init_sigmay"])          # init_pml's profile, evaluated on the host
    f, _ = oracle_from_golden(G)           # only for the per-step envelope values (begin_md_step), checked against the fixture
    t = 0.0
    for step in range(len(G["md_nsub"])):
        nsub, tau = f.begin_md_step(t, float(a["timestep"]))
        assert nsub == int(G["md_nsub"][step])
        if step == 0:                       # the first sub-steps one by one (fixture dumps after sub-steps 0, 1, 7)
            done = 0
            for upto in (1, 2, 8):
                s.do_fdtd(t + done * tau, upto - done, tau, f.f.laser_timefun); done = upto
                F = s.fdtd_fields()
                for k in ("Ezx", "Hy"):
                    assert _rel(F[k], G["s0_sub%d_%s" % (upto - 1, k)]) < 1e-9, (upto, k)
            s.do_fdtd(t + done * tau, nsub - done, tau, f.f.laser_timefun)
        else:
            s.do_fdtd(t, nsub, tau, f.f.laser_timefun)
        t += nsub * tau
        F = s.fdtd_fields()
        src = s.ttm_download(fields=("source",))["source"][:, 1, 1]
        tag = "md%d" % step
        for k in ("Ezx", "Hy", "Jzx", "Jlzx", "Pzx", "Jzx_old", "Jlzx_old"):
            assert _rel(F[k], G["%s_%s" % (tag, k)]) < 1e-9, (tag, k)
        assert _rel(src, G[tag + "_source"]) < 1e-9, tag
        assert np.all(src[G["cell_natoms"] < 5] == 0.0)
    f.close(); s.close()


@pytest.mark.parametrize("dl", ["constants", "tables"])
def test_solver_loop_with_diffusion_vs_oracle(dl):
    """one MD step of calc_ttm's solver loop on a slab: update_fd .. FILLMESH as usual, then 437 x [do_fdtd; do_DIFF;
    do_FILLMESH] in ONE launch; the oracle runs the same functions one after the other on the host.  dl = tables: every
    FILLMESH takes the cells' Drude-Lorentz parameters from fitDL's tricubic tables (imd_ttm_3D.c:600, :2919-2968)"""
    hx, nxc = 7.0, 120
    # an Al slab of 60 cells behind 40 cells of vacuum (the source sits in the vacuum), 20 cells of vacuum behind it
    nb = (int(round(60 * hx / 4.05)), 5, 5)
    pos = lattice.fcc_positions(nb, 4.05) * np.array([60 * hx / (nb[0] * 4.05), 21.0 / (5 * 4.05), 21.0 / (5 * 4.05)])
    pos = pos + np.array([40 * hx + 0.3, 0.0, 0.0])
    rng = np.random.default_rng(2)
    pos = pos + rng.normal(scale=0.03, size=pos.shape)
    pos[:, 1:] %= 21.0
    mom = lattice.maxwell_momenta(len(pos), 26.9815, 0.03, seed=2)
    s, (pair, cellsz, margin, box) = _row_system(nxc, hx, (pos, mom))
    par = dict(I0=4e15, lam=800e-9, sigma_t=1.2e-15, t_0=0.6e-15, srcx=25 * hx * 1e-10, bw=10, Sc=1.0)
    s.init_fdtd(**par)
    s.calc_forces(0)
    # oracle: same atoms, same lattice
    o = po.OracleSystem((nxc + 2, 5, 5), 96, 1, box, (0, 1, 1), cellsz, margin, 0.1)
    tabs = potential.synthetic_eam_tables()
    for w, t_ in enumerate(tabs[:3]):
        o.set_table(w, t_.as_dict())
    o.fill_from_positions(pos, mom, 26.9815)
    o.s.count_ttm_neigh = 1
    o.calc_forces(0)
    cve, ee = ttm_mod.synthetic_eos()
    ot = po.OracleTTM(o, (1, 3, 3), 5, 26.9815, 13.0, float(pair.end[0]))
    ot.set_eos(0, len(cve["x"]), len(cve["y"]), cve["x"][0], cve["x"][-1], cve["y"][0], cve["y"][-1], cve["z"])
    ot.set_eos(1, len(ee["x"]), len(ee["y"]), ee["x"][0], ee["x"][-1], ee["y"][0], ee["y"][-1], ee["z"])
    of = po.OracleFDTD(nxc + 2, hx, par["I0"], par["lam"], par["sigma_t"], par["t_0"], par["srcx"], par["bw"], par["Sc"])
    ot.t.omega_laser = of.f.omega_laser                    # init_fdtd sets the global the wide-range model reads
    otabs = None
    if dl == "tables":
        tabs = tricub.synthetic_dl_tables(1500.0, 4000.0)
        s.fdtd_set_dl_tables(tabs)
        otabs = [po.OracleTricub(t_.range, t_.x, t_.y, t_.z, t_.f) for t_ in tabs]
    dl_seen = []
    AMU, ECH, ECONST, EMASS = 1.6605390402e-27, 1.60217662e-19, 8.841941282883075e-12, 9.10938356e-31
    for step in (0, 1):                                     # step 0 re-initialises U in FILLMESH; step 1 is the regular path
        ot.update_fd(step); ot.cell_activation(); ot.fillmesh(step); ot.fill_ghosts()
        nsub, tau = of.begin_md_step(of_t := (0.0 if step == 0 else of_t_end), 0.1)
        ndiff = max(1, int(max(nsub, 100) / nsub))
        tau_d = tau / ndiff / (10.18 * 1.0e-15)
        ot.t.laser_active = of.f.laser_active
        of.natoms[:] = ot.natoms[:, 1, 1]
        t = of_t
        for sub in range(nsub):
            act = of.natoms >= 5
            if otabs is None:
                of.DL[act, :5] = GpuSystem.AL_DRUDE_LORENTZ
                of.DL[act, 5] = np.sqrt(ot.ne[:, 1, 1][act] * ECH * ECH / ECONST / EMASS)
            else:                                               # fitDL at the end of the FILLMESH just done
                d6, st = po.oracle_fitdl(otabs, ot.temp[:, 1, 1][act], ot.md_temp[:, 1, 1][act], ot.dens[:, 1, 1][act], ot.ne[:, 1, 1][act])
                assert (st == 0).all()
                of.DL[act] = d6
                if sub % 100 == 0:
                    dl_seen.append(d6[:, 1].copy())
            if of.f.laser_active:
                of.step(t)
                ot.source[:, 1, 1] = of.source
            for q in range(ndiff):
                ot.diff(tau_d, ndiff); ot.fillmesh(step); ot.fill_ghosts()
            t += tau
        of_t_end = t
        s.calc_ttm(step)
        F = s.fdtd_fields()
        lat = s.ttm_download()
        assert of.f.laser_active == 1 and of.source.max() > 0
        for k in ("Ezx", "Hy", "Jzx", "Jlzx", "Pzx"):
            assert _rel(F[k], getattr(of, k)) < 1e-9, (step, k)
        row = (slice(1, -1), 1, 1)
        assert np.array_equal(lat["natoms"][row], ot.natoms[row])
        for k in ("temp", "U", "xi", "Ce", "fd_k", "fd_g", "source"):
            assert _rel(lat[k][row], getattr(ot, k)[row]) < 1e-9, (step, k)
        assert abs(s.get_eabs() - ot.t.eabs) <= 1e-9 * abs(ot.t.eabs)
        assert abs(s.fdtd_info()["t_SI"] - t) < 1e-9 * t
        if otabs is not None:
            # the cells' parameters after the last FILLMESH of the loop: table nodes, so equal or one node apart
            act = of.natoms >= 5
            d6, _ = po.oracle_fitdl(otabs, ot.temp[:, 1, 1][act], ot.md_temp[:, 1, 1][act], ot.dens[:, 1, 1][act], ot.ne[:, 1, 1][act])
            got = s.fdtd_cell_dl()[act]
            assert (got[:, :5] == d6[:, :5]).mean() > 0.98 and _rel(got[:, 5], d6[:, 5]) < 1e-9
        if step == 0:
            # the electrons did heat up where the field got in, and only there
            assert lat["temp"][row].max() > 1.05 * ot.md_temp[row].max()
    if otabs is not None:
        # the look-up did move with the electrons heating up: the Drude damping of some cell changed during the pulse
        assert max(np.abs(a_ - dl_seen[0]).max() for a_ in dl_seen[1:]) > 0.0
        for t_ in otabs:
            t_.close()
    of.close(); ot.close(); o.close(); s.close()


@pytest.mark.parametrize("nyz", [9, 46])
def test_update_fd_for_fd_cells_of_many_md_cells(nyz):
    """the one-row lattice of a wide slab puts thousands of MD cells under one FD cell (fd_ext 1 ny nz): update_fd then runs a
    workgroup per FD cell (k_ttm_update_fd_wide; 81 MD cells per FD cell) or, from 2048 MD cells per FD cell on, 64 chunk
    workgroups per FD cell and a fixed-order combine (k_ttm_ufd_part / _combine; 2116 cells) -- same numbers as the oracle's
    sequential loop (integers exact, sums to rounding)"""
    hx, nxc = 7.0, 12
    pair, rho, emb, r2cut = potential.synthetic_eam_tables()
    margin = 0.4
    cellsz = (np.sqrt(r2cut) + margin) ** 2
    box = np.diag([nxc * hx, nyz * 7.0, nyz * 7.0])
    nuc = int(round(nyz * 7.0 / 4.05))
    nb = (int(round(8 * hx / 4.05)), nuc, nuc)
    pos = lattice.fcc_positions(nb, 4.05) * np.array([8 * hx / (nb[0] * 4.05), box[1, 1] / (nuc * 4.05), box[2, 2] / (nuc * 4.05)])
    pos = pos + np.array([2 * hx + 0.3, 0.0, 0.0])
    rng = np.random.default_rng(4)
    pos = pos + rng.normal(scale=0.03, size=pos.shape)
    pos[:, 1:] %= box[1, 1]
    mom = lattice.maxwell_momenta(len(pos), 26.9815, 0.05, seed=4)
    dim = (nxc + 2, nyz + 2, nyz + 2)
    s = GpuSystem(box, (0, 1, 1), dim, 1, cellsz, margin, 0.1, count_ttm_neighbours=True)
    s.upload_tables(pair, rho, emb)
    s.upload_atoms(pos, mom, 26.9815)
    cve, ee = ttm_mod.synthetic_eos()
    s.init_ttm((1, nyz, nyz), 5, 100, 26.9815, 13.0, cve, ee)
    assert s.fd_dim == (nxc + 2, 3, 3)
    s.calc_forces(0)
    s.update_fd(0)
    lat = s.ttm_download()
    o = po.OracleSystem(dim, 96, 1, box, (0, 1, 1), cellsz, margin, 0.1)
    for w, t_ in enumerate((pair, rho, emb)):
        o.set_table(w, t_.as_dict())
    o.fill_from_positions(pos, mom, 26.9815)
    o.s.count_ttm_neigh = 1
    o.calc_forces(0)
    ot = po.OracleTTM(o, (1, nyz, nyz), 5, 26.9815, 13.0, float(pair.end[0]))
    ot.update_fd(0)
    row = (slice(1, -1), 1, 1)
    assert np.array_equal(lat["natoms"][row], ot.natoms[row]) and int(ot.natoms[row].sum()) == len(pos)
    for k in ("md_temp", "dens", "temp"):
        assert _rel(lat[k][row], getattr(ot, k)[row]) < 1e-12, k
    assert np.abs(lat["vcom"][row] - ot.vcom[row]).max() < 1e-12 * np.abs(ot.vcom[row]).max() + 1e-18
    ot.close(); o.close(); s.close()
