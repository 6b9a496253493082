"""GPU parity AT SCALE (run with -m gpu): BASELINE.json config 2's own criterion -- "1M-atom FCC Al, EAM, forces vs
CPU to 1e-9" -- checked literally: the HIP path through the C ABI against the CPU oracle (oracle/imd_oracle.c, the
bit-exact restatement of imd_forces_nbl.c:136-287,419-985) on the SAME 1 048 576-atom thermalised, perturbed
configuration.  At this size the tile kernels run thousands of tiles through the persistent block ranges, the
XCD-contiguous walk and the staging plan, which the 2 048..6 912-atom goldens cannot exercise.

Bars: neighbour list (tl, tb, cl_off in the reference's format) bit-exact; numneighs exact; forces <= 1e-9 of max|F|;
rho, dF, per-atom energies, total energy <= 1e-11 relative; virial <= 1e-10; TTM lattice after one calc_ttm
(update_fd + 100 fused sub-steps, imd_ttm_3D.c:50-77) integers exact, fields <= 1e-10."""
import numpy as np
import pytest

from imd_amd import lattice, potential, ttm as ttm_mod
from imd_amd.system import GpuSystem, global_cell_dim
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _relmax(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


def _setup(nbn, ttmn, seed=3, cd=(1, 1, 1), temp=0.02, grids="shared"):
    nb = (nbn,) * 3
    box = lattice.fcc_box(nb)
    pos = lattice.fcc_positions(nb)
    rng = np.random.default_rng(seed)
    pos = (pos + rng.normal(scale=0.06, size=pos.shape)) % np.diag(box)
    mom = lattice.maxwell_momenta(len(pos), 26.9815, temp, seed=seed)
    pair, rho, emb, r2cut = potential.synthetic_eam_tables(grids=grids)
    cellsz = (np.sqrt(r2cut) + 0.5) ** 2
    dim = [g + 2 for g in global_cell_dim(box, cellsz, cd)]
    o = po.OracleSystem(dim, 64, 1, box, (1, 1, 1), cellsz, 0.5, 0.1)
    for w, t in enumerate((pair, rho, emb)):
        o.set_table(w, t.as_dict())
    o.fill_from_positions(pos, mom, 26.9815)
    o.s.count_ttm_neigh = 1 if ttmn else 0
    g = GpuSystem(box, (1, 1, 1), dim, 1, cellsz, 0.5, 0.1, count_ttm_neighbours=ttmn)
    g.upload_tables(pair, rho, emb)
    # hand the device exactly the oracle's cell arrangement (same atoms in the same slots of the same cells)
    g.upload_cells(o.n.copy(), o.gather_inner("ort"), o.gather_inner("impuls"), o.gather_inner("masse"),
                   o.gather_inner("sorte").astype(np.int32), o.gather_inner("nummer").astype(np.int32))
    return o, g, pair


def _compare_forces(o, g, ttmn):
    o.calc_forces(0)
    r = g.calc_forces(0)
    assert g.get_info("list_mode") == 1 and g.get_info("fused_embedding") == 1      # the tile family, no fall-back
    d = g.download(fields=("kraft", "pot_eng", "eam_rho", "eam_dF", "numneighs", "nummer"))
    assert np.array_equal(d["nummer"], o.gather_inner("nummer"))
    fo = o.gather_inner("kraft")
    assert _relmax(d["kraft"], fo) < 1e-9
    assert _relmax(d["pot_eng"], o.gather_inner("pot_eng")) < 1e-11
    assert _relmax(d["eam_rho"], o.gather_inner("rho")) < 1e-11
    assert _relmax(d["eam_dF"], o.gather_inner("dF")) < 1e-11
    if ttmn:
        assert np.array_equal(d["numneighs"], o.gather_inner("numneighs"))
        assert d["numneighs"].min() > 20
    assert abs(r["tot_pot_energy"] - o.s.tot_pot_energy) < 1e-11 * abs(o.s.tot_pot_energy)
    assert abs(r["virial"] - o.s.virial) < 1e-10 * abs(o.s.virial)
    assert r["is_short"] == 0 == o.s.is_short
    return float(np.abs(d["kraft"] - fo).max() / np.abs(fo).max())


@pytest.mark.parametrize("ttmn,grids", [(False, "shared"), (True, "shared"), (False, "distinct"), (True, "distinct")])
def test_config2_forces_and_list_vs_oracle_1M(ttmn, grids):
    """64^3 unit cells = 1 048 576 atoms (BASELINE config 2), tile kernels with the embedding pass fused; grids = "distinct":
    phi, rho and F tabulated on three different grids (begin, end, step, rows: potential.EAM_GRIDS), i.e. the distinct-grids
    form of pass 1 and the coefficient form of the force pass on rho_h_tab's own grid"""
    o, g, _ = _setup(64, ttmn, grids=grids)
    err = _compare_forces(o, g, ttmn)
    assert g.get_info("pass1_form") == (2 if grids == "shared" else 1)
    assert g.get_info("p3_coef_rows") > 0 and g.get_info("p3_standin_passes") == 0
    print("1M atoms, ttm neighbours %s, %s grids: max force error / max|F| = %.3e" % (ttmn, grids, err))
    # the device's symmetric LDS-slot list, exported in the reference's half-list format, is the oracle's list
    tl, tb, cl_off = g.export_nblist()
    assert np.array_equal(cl_off, o.cl_off[:-1])
    assert np.array_equal(tl, o.tl)
    assert np.array_equal(tb, o.tb)
    ent, _ = g.nblist_stats()
    assert ent == 2 * len(o.tb)
    g.close(); o.close()


def test_config2_global_gather_family_vs_oracle_1M():
    """the second kernel family (ELL int32 list, positions through L2) against the oracle at the same size"""
    o, g, _ = _setup(64, False, seed=5)
    g.set_option("mode", 0)
    o.calc_forces(0)
    r = g.calc_forces(0)
    assert g.get_info("list_mode") == 0
    d = g.download(fields=("kraft", "eam_rho", "nummer"))
    assert np.array_equal(d["nummer"], o.gather_inner("nummer"))
    assert _relmax(d["kraft"], o.gather_inner("kraft")) < 1e-9
    assert _relmax(d["eam_rho"], o.gather_inner("rho")) < 1e-11
    assert abs(r["tot_pot_energy"] - o.s.tot_pot_energy) < 1e-11 * abs(o.s.tot_pot_energy)
    g.close(); o.close()


def test_config2_calc_ttm_vs_oracle_1M():
    """one calc_ttm at 1M atoms (fd_ext 2 2 2 like config 3: 18^3 active FD cells, 100 sub-steps)"""
    o, g, pair = _setup(64, True, seed=11, cd=(2, 2, 2))
    _compare_forces(o, g, True)
    cve, ee = ttm_mod.synthetic_eos()
    ot = po.OracleTTM(o, (2, 2, 2), 5, 26.9815, 13.0, float(pair.end[0]))
    ot.set_eos(0, len(cve["x"]), len(cve["y"]), cve["x"][0], cve["x"][-1], cve["y"][0], cve["y"][-1], cve["z"])
    ot.set_eos(1, len(ee["x"]), len(ee["y"]), ee["x"][0], ee["x"][-1], ee["y"][0], ee["y"][-1], ee["z"])
    ot.calc_ttm(0, 100)
    ttm_mod.attach_default(g, fd_ext=(2, 2, 2), diff_substeps=100, fd_min_atoms=5)
    g.calc_ttm(0)
    lat = g.ttm_download()
    inner = (slice(1, -1),) * 3
    assert tuple(lat["natoms"].shape) == ot.dim
    assert np.array_equal(lat["natoms"][inner], ot.natoms[inner])
    assert int(lat["natoms"][inner].sum()) == 1048576
    for f in ("temp", "xi", "md_temp", "dens", "fd_k", "fd_g", "Ce", "U"):
        ref = getattr(ot, f)[inner]
        assert np.isfinite(ref).all()
        err = np.abs(lat[f][inner] - ref).max() / np.abs(ref).max()
        assert err < 1e-10, "%s rel err %.3e" % (f, err)
    assert np.abs(lat["vcom"][inner] - ot.vcom[inner]).max() < 1e-12 * np.abs(ot.vcom[inner]).max() + 1e-18
    mdt = g.CFL_maxdt(0)
    assert abs(mdt - ot.t.max_dt_ttm) < 1e-10 * mdt
    # and the coupled integrator on top: one move_atoms_ttm on both sides
    ot.move_atoms_ttm()
    g.move_atoms_ttm()
    d = g.download(fields=("ort", "impuls", "nummer"))
    assert _relmax(d["impuls"], o.gather_inner("impuls")) < 1e-11
    assert np.abs(d["ort"] - o.gather_inner("ort")).max() < 1e-11
    g.close(); ot.close(); o.close()


@pytest.mark.parametrize("grids", ["shared", "distinct"])
def test_trajectory_vs_oracle_442k(grids):
    """48^3 unit cells = 442 368 atoms, 25 device-resident NVE steps (forces, leap-frog, list checks, two
    re-binnings + list rebuilds) against the oracle's loop of the same reference functions; compared by atom number."""
    o, g, _ = _setup(48, False, seed=7, temp=0.08, grids=grids)      # hot enough for list rebuilds at steps 9 and 23
    nsteps = 25
    for st in range(nsteps):
        # main loop of the reference (imd_main_3d.c): calc_forces [fix_cells + make_nblist when the list is invalid,
        # imd_forces_nbl.c:300-315] -> move_atoms -> check_nblist
        o.calc_forces(st)
        o.move_atoms_nve()
        o.check_nblist()
        g.md_step(st)
    o.calc_forces(nsteps)
    r = g.calc_forces(nsteps)
    assert o.s.nbl_count >= 3                      # the run did rebuild (twice after the initial build)
    assert g.get_sums()["nbl_count"] == o.s.nbl_count
    d = g.download(fields=("ort", "impuls", "kraft", "nummer"))
    og, oo = np.argsort(d["nummer"]), np.argsort(o.gather_inner("nummer"))
    box = np.array(o.s.box[:]).reshape(3, 3).diagonal()
    dx = d["ort"][og] - o.gather_inner("ort")[oo]
    dx -= np.round(dx / box) * box
    assert np.abs(dx).max() < 1e-9
    assert _relmax(d["impuls"][og], o.gather_inner("impuls")[oo]) < 1e-9
    assert _relmax(d["kraft"][og], o.gather_inner("kraft")[oo]) < 1e-8
    assert abs(r["tot_pot_energy"] - o.s.tot_pot_energy) < 1e-10 * abs(o.s.tot_pot_energy)
    g.close(); o.close()
