"""GPU edge cases (run with -m gpu): arrangements that take the less travelled branches of the tile kernels --
crowded cells (more than 32 atoms: second staging sweep, tile_cap above the 768-thread workgroup), neighbour rows that
outgrow the default width (list regrown), boxes whose cell counts are not multiples of the tile, a slab with empty
cells, a box of a single tile.  Every case: both kernel families against the CPU oracle on the same input."""
import numpy as np
import pytest

from imd_amd import lattice, potential
from imd_amd.system import GpuSystem, global_cell_dim
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _case(nb, a, pbc=(1, 1, 1), jitter=0.05, vacuum=0.0, seed=1, grids="shared"):
    pair, rho, emb, r2cut = potential.synthetic_eam_tables(grids=grids)
    cellsz = (np.sqrt(r2cut) + 0.5) ** 2
    box = lattice.fcc_box(nb, a)
    pos = lattice.fcc_positions(nb, a)
    if vacuum > 0:                                     # open x: slab in the middle of a longer box
        box = box.copy(); box[0, 0] += 2 * vacuum; pos = pos + np.array([vacuum, 0, 0])
    rng = np.random.default_rng(seed)
    pos = pos + rng.normal(scale=jitter, size=pos.shape)
    mom = lattice.maxwell_momenta(len(pos), 26.9815, 0.03, seed=seed)
    gd = global_cell_dim(box, cellsz)
    dim = [g + 2 for g in gd]
    return dict(box=box, pos=pos, mom=mom, dim=dim, cellsz=cellsz, tabs=(pair, rho, emb), pbc=pbc)


def _oracle(c, cap):
    o = po.OracleSystem(c["dim"], cap, 1, c["box"], c["pbc"], c["cellsz"], 0.5, 0.1)
    for w, t in enumerate(c["tabs"]):
        o.set_table(w, t.as_dict())
    L = np.diag(c["box"])
    p = c["pos"].copy()
    for d in range(3):
        if c["pbc"][d]:
            p[:, d] %= L[d]
    o.fill_from_positions(p, c["mom"], 26.9815)
    o.calc_forces(0)
    idn = o.gather_inner("nummer")
    return o.gather_inner("kraft")[np.argsort(idn)], o.gather_inner("rho")[np.argsort(idn)], o.s.tot_pot_energy


def _gpu(c, mode):
    s = GpuSystem(c["box"], c["pbc"], c["dim"], 1, c["cellsz"], 0.5, 0.1)
    s.set_option("mode", mode)
    s.upload_tables(*c["tabs"])
    s.upload_atoms(c["pos"], c["mom"], 26.9815)
    r = s.calc_forces(0)
    d = s.download(fields=("kraft", "nummer", "eam_rho"))
    o = np.argsort(d["nummer"])
    info = {k: s.get_info(k) for k in ("list_mode", "tile_cap", "tile_threads", "W", "tile_x", "tile_y", "tile_z")}
    for st in range(3):                                # and the device loop keeps running on it
        s.md_step(st)
    assert np.isfinite(s.download(fields=("ort",))["ort"]).all()
    s.close()
    return d["kraft"][o], d["eam_rho"][o], r["tot_pot_energy"], info


CASES = {
    # a = 3.3 A: 1.85x the density -> ~40 atoms per cell, ~150 neighbours: rows regrow past W = 96, cells > 32 atoms
    "crowded": dict(nb=(14, 14, 14), a=3.3, cap=96),
    # 11 x 9 x 7 cells: no tile shape divides all three
    "ragged_box": dict(nb=(20, 16, 13), a=4.05, cap=64),
    # open x with vacuum: empty cells inside the tile halos
    "slab": dict(nb=(10, 12, 12), a=4.05, cap=64, pbc=(0, 1, 1), vacuum=9.0),
    # 3 x 3 x 3 cells: one tile, every halo cell is a periodic image
    "single_tile": dict(nb=(6, 6, 6), a=4.05, cap=64),
}


@pytest.mark.parametrize("grids", ["shared", "distinct"])
@pytest.mark.parametrize("name", list(CASES))
def test_edge_case_both_families_vs_oracle(name, grids):
    k = dict(CASES[name])
    cap = k.pop("cap")
    c = _case(grids=grids, **k)
    fo, ro, eo = _oracle(c, cap)
    scale = np.abs(fo).max()
    for mode in (0, 1):
        f, r, e, info = _gpu(c, mode)
        assert info["list_mode"] == mode, info
        assert np.abs(f - fo).max() < 1e-9 * scale, (name, mode, info)
        assert np.abs(r - ro).max() < 1e-11 * np.abs(ro).max()
        assert abs(e - eo) < 1e-11 * abs(eo)
        if name == "crowded" and mode == 1:
            assert info["W"] > 96 and info["tile_cap"] >= 64


@pytest.mark.parametrize("grids", ["shared", "distinct"])
@pytest.mark.parametrize("dist,short", [(1.2, 1), (1.6, 0)])
def test_a_pair_far_inside_the_first_shell(dist, short, grids):
    """Two atoms at 1.2 A: below the first table row (r_begin = 1.5 A) -- PAIR_INT2 clamps the argument and raises is_short
    (potaccess.h:331, imd_forces_nbl.c:673,986-993).  At 1.6 A: inside the table, but below the rows the coefficient form of
    the EAM force pass keeps in LDS (r < 1.71 A with the 4000-row table and 3136 slots), so that step is the three-row kernel's.
    Forces, densities, energy against the oracle either way; the flag as the reference sets it."""
    # (distinct grids: rho_h_tab begins at 1.7 A, so the 1.6 A pair raises is_short through the DENSITY look-up alone)
    if grids == "distinct" and dist == 1.6:
        short = 1
    c = _case((12, 12, 12), 4.05, jitter=0.02, seed=3, grids=grids)
    c["pos"][1] = c["pos"][0] + np.array([dist, 0.0, 0.0])
    o = po.OracleSystem(c["dim"], 64, 1, c["box"], c["pbc"], c["cellsz"], 0.5, 0.1)
    for w, t in enumerate(c["tabs"]):
        o.set_table(w, t.as_dict())
    L = np.diag(c["box"])
    o.fill_from_positions(c["pos"] % L, c["mom"], 26.9815)
    o.calc_forces(0)
    idn = np.argsort(o.gather_inner("nummer"))
    fo, ro = o.gather_inner("kraft")[idn], o.gather_inner("rho")[idn]
    assert o.s.is_short == short
    for mode in (0, 1):
        s = GpuSystem(c["box"], c["pbc"], c["dim"], 1, c["cellsz"], 0.5, 0.1)
        s.set_option("mode", mode)
        s.upload_tables(*c["tabs"])
        s.upload_atoms(c["pos"], c["mom"], 26.9815)
        r = s.calc_forces(0)
        d = s.download(fields=("kraft", "nummer", "eam_rho"))
        if mode == 1 and grids == "shared":
            assert s.get_info("p3_coef_klo") > 31, "the 1.6 A pair must lie below the rows held in LDS for this test to mean anything"
        s.close()
        k = np.argsort(d["nummer"])
        assert r["is_short"] == short, mode
        assert np.abs(d["kraft"][k] - fo).max() < 1e-9 * np.abs(fo).max(), mode
        assert np.abs(d["eam_rho"][k] - ro).max() < 1e-11 * np.abs(ro).max()
        assert abs(r["tot_pot_energy"] - o.s.tot_pot_energy) < 1e-11 * abs(o.s.tot_pot_energy)


def test_host_integrator_hand_over_of_positions_and_momenta():
    """imdgpu_update_positions / imdgpu_update_momenta: a host that integrates itself (shim, level 1) hands over the
    arrays in the cell order download() reports; the device holds exactly those values at the same atoms."""
    c = _case((8, 8, 8), 4.05)
    s = GpuSystem(c["box"], c["pbc"], c["dim"], 1, c["cellsz"], 0.5, 0.1)
    s.upload_tables(*c["tabs"])
    s.upload_atoms(c["pos"], c["mom"], 26.9815)
    s.calc_forces(0)
    d = s.download(fields=("ort", "impuls", "nummer"))
    rng = np.random.default_rng(5)
    new_x = d["ort"] + 1e-3 * rng.standard_normal(d["ort"].shape)
    new_p = rng.standard_normal(d["impuls"].shape)
    s.update_positions(new_x); s.update_momenta(new_p)
    e = s.download(fields=("ort", "impuls", "nummer"))
    assert np.array_equal(e["nummer"], d["nummer"])
    assert np.array_equal(e["ort"], new_x) and np.array_equal(e["impuls"], new_p)
    s.close()


def test_tiny_and_empty_systems():
    """The smallest inputs: forces before any upload are refused loudly, an upload of zero atoms is a legal empty rank
    (test_rank_without_atoms); one atom feels nothing; two atoms feel equal and opposite forces that match the oracle; the device loop runs on them."""
    from imd_amd.lib import ImdGpuError
    pair, rho, emb, r2cut = potential.synthetic_eam_tables()
    cellsz = (np.sqrt(r2cut) + 0.5) ** 2
    box = lattice.fcc_box((8, 8, 8), 4.05)
    dim = [g + 2 for g in global_cell_dim(box, cellsz)]
    both = np.array([[1.0, 1.0, 1.0], [3.2, 1.4, 0.9]])
    for n in (0, 1, 2):
        s = GpuSystem(box, (1, 1, 1), dim, 1, cellsz, 0.5, 0.1)
        s.upload_tables(pair, rho, emb)
        if n == 0:
            with pytest.raises(ImdGpuError):
                s.calc_forces(0)                       # nothing uploaded yet
            s.upload_atoms(both[:0], np.zeros((0, 3)), 26.9815)
            assert s.calc_forces(0)["tot_pot_energy"] == 0.0
            s.close()
            continue
        s.upload_atoms(both[:n], np.zeros((n, 3)), 26.9815)
        r = s.calc_forces(0)
        d = s.download(fields=("kraft", "nummer"))
        f = d["kraft"][np.argsort(d["nummer"])]
        o = po.OracleSystem(dim, 16, 1, box, (1, 1, 1), cellsz, 0.5, 0.1)
        for w, t in enumerate((pair, rho, emb)):
            o.set_table(w, t.as_dict())
        o.fill_from_positions(both[:n], np.zeros((n, 3)), 26.9815)
        o.calc_forces(0)
        fo = o.gather_inner("kraft")[np.argsort(o.gather_inner("nummer"))]
        assert np.abs(f - fo).max() <= 1e-12 * max(np.abs(fo).max(), 1.0)
        assert abs(r["tot_pot_energy"] - o.s.tot_pot_energy) <= 1e-12 * max(abs(o.s.tot_pot_energy), 1.0)
        if n == 2:
            assert np.abs(f[0] + f[1]).max() < 1e-14 and np.abs(f[0]).max() > 0.1
        for st in range(3):
            s.md_step(st)
        s.close()


def test_rank_without_atoms():
    """a rank whose domain is vacuum (laser-ablation slab split along x) owns no atoms -- the reference runs
    such ranks.  Every entry point must accept nlocal = 0: no launches fault, sums are zero, the TTM lattice of that rank is
    all inactive."""
    from imd_amd import ttm as ttm_mod
    c = _case((8, 8, 8), 4.05)
    s = GpuSystem(c["box"], c["pbc"], c["dim"], 1, c["cellsz"], 0.5, 0.1, count_ttm_neighbours=True)
    s.upload_tables(*c["tabs"])
    s.upload_atoms(np.zeros((0, 3)), np.zeros((0, 3)), 26.9815)
    assert s.num_atoms()[0] == 0
    r = s.calc_forces(0)
    assert r["tot_pot_energy"] == 0.0 and r["virial"] == 0.0 and r["is_short"] == 0
    ttm_mod.attach_default(s, fd_ext=(1, 1, 1), diff_substeps=10, fd_min_atoms=5)
    for st in range(3):
        s.md_step(st)
    lat = s.ttm_download(fields=("natoms", "temp", "xi"))
    assert int(lat["natoms"].sum()) == 0 and np.all(lat["xi"] == 0.0)
    d = s.download()
    assert d["ort"].shape == (0, 3) and int(d["cell_n"].sum()) == 0
    assert s.get_sums()["tot_kin_energy"] == 0.0
    s.close()


def test_restrictions_scale_the_forces_in_the_integrator():
    """move_atoms_nve: KRAFT *= restrictions before the momenta are advanced (imd_integrate.c:98-104 via the oracle's
    restatement); with every factor 1 the integrator skips the write-back, with others it must not"""
    from imd_amd import lattice, potential
    from imd_amd.system import global_cell_dim
    from oracle import pyoracle as po
    nb = (8, 8, 8)
    box = lattice.fcc_box(nb)
    pos = lattice.fcc_positions(nb)
    rng = np.random.default_rng(21)
    pos = (pos + rng.normal(scale=0.05, size=pos.shape)) % np.diag(box)
    mom = lattice.maxwell_momenta(len(pos), 26.9815, 0.03, seed=21)
    pair, rho, emb, r2cut = potential.synthetic_eam_tables()
    cellsz = (np.sqrt(r2cut) + 0.5) ** 2
    dim = [g + 2 for g in global_cell_dim(box, cellsz, (1, 1, 1))]
    for restr in ((1.0, 1.0, 1.0), (1.0, 0.0, 0.5)):
        o = po.OracleSystem(dim, 64, 1, box, (1, 1, 1), cellsz, 0.5, 0.1)
        for w, t in enumerate((pair, rho, emb)):
            o.set_table(w, t.as_dict())
        o.fill_from_positions(pos, mom, 26.9815)
        for q in range(3):
            o.s.restrictions[q] = restr[q]
        g = GpuSystem(box, (1, 1, 1), dim, 1, cellsz, 0.5, 0.1)
        g.upload_tables(pair, rho, emb)
        g.upload_cells(o.n.copy(), o.gather_inner("ort"), o.gather_inner("impuls"), o.gather_inner("masse"),
                       o.gather_inner("sorte").astype(np.int32), o.gather_inner("nummer").astype(np.int32))
        g.set_restrictions([restr])
        for st in range(4):
            o.calc_forces(st); o.move_atoms_nve(); o.check_nblist()
            g.md_step(st)
        d = g.download(fields=("ort", "impuls", "kraft", "nummer"))
        og, oo = np.argsort(d["nummer"]), np.argsort(o.gather_inner("nummer"))
        fo = o.gather_inner("kraft")[oo]
        assert np.abs(d["kraft"][og] - fo).max() < 1e-9 * np.abs(fo).max()
        assert np.abs(d["impuls"][og] - o.gather_inner("impuls")[oo]).max() < 1e-10 * np.abs(mom).max()
        assert np.abs(d["ort"][og] - o.gather_inner("ort")[oo]).max() < 1e-10
        if restr[1] == 0.0:
            assert np.abs(d["kraft"][:, 1]).max() == 0.0          # written back scaled
        g.close(); o.close()


def _two_type_case(seed=5):
    """Cu3Au (L1_2) decoration, the full-size two-type tables (4 columns each with its own begin / end / step / rows)"""
    import os
    import tempfile
    nb, a = (12, 12, 12), 4.05
    with tempfile.TemporaryDirectory() as d:
        potential.write_synthetic_eam_files_2types(d)
        pair, c1 = potential.read_pot_table(os.path.join(d, "phi.pt"), 4, True)
        rho, c2 = potential.read_pot_table(os.path.join(d, "rho.pt"), 4, True)
        emb, _ = potential.read_pot_table(os.path.join(d, "F.pt"), 2, False)
    cellsz = (np.sqrt(max(c1, c2)) + 0.5) ** 2
    box = lattice.fcc_box(nb, a)
    pos = lattice.fcc_positions(nb, a)
    half = np.rint(pos / (a / 2.0) - 0.5).astype(np.int64)
    sorte = np.all(half % 2 == 0, axis=1).astype(np.int32)
    rng = np.random.default_rng(seed)
    pos = pos + rng.normal(scale=0.04, size=pos.shape)
    masse = np.where(sorte == 1, 63.546, 26.9815)
    mom = lattice.maxwell_momenta(len(pos), 1.0, 0.03, seed=seed) * np.sqrt(masse)[:, None]
    dim = [g + 2 for g in global_cell_dim(box, cellsz)]
    return dict(box=box, pos=pos, mom=mom, masse=masse, sorte=sorte, dim=dim, cellsz=cellsz, tabs=(pair, rho, emb))


@pytest.mark.parametrize("dist", [None, 1.55, 1.2])
def test_two_types_record_kernels_and_global_gather_vs_oracle(dist):
    """Two atom types, tables beyond the LDS: the record kernels of the tile family (k_pass1_tile_rec / k_pass3_tile_rec; tables in the
    LDS and as records through L2, in every mix that fits) and the global-gather family against the oracle -- also with a pair of unlike atoms at 1.55 A (above pair_pot.begin of column 0, BELOW
    that of the mixed columns, 1.6 A: is_short through the per-column begin, the argument clamped per column, potaccess.h:331) and
    at 1.2 A (below every column)."""
    c = _two_type_case()
    if dist is not None:
        i = int(np.nonzero(c["sorte"] == 1)[0][3])
        j = int(np.nonzero(c["sorte"] == 0)[0][5])
        c["pos"][j] = c["pos"][i] + np.array([0.0, dist, 0.0])
    o = po.OracleSystem(c["dim"], 64, 2, c["box"], (1, 1, 1), c["cellsz"], 0.5, 0.1)
    for w, t in enumerate(c["tabs"]):
        o.set_table(w, t.as_dict())
    o.fill_from_positions(c["pos"] % np.diag(c["box"]), c["mom"], c["masse"], c["sorte"])
    o.calc_forces(0)
    idn = np.argsort(o.gather_inner("nummer"))
    fo, ro, eo = o.gather_inner("kraft")[idn], o.gather_inner("rho")[idn], o.gather_inner("pot_eng")[idn]
    assert o.s.is_short == (0 if dist is None else 1)
    for mode, lds in ((0, -1), (1, -1), (1, 0), (1, 2), (1, 7)):     # lds: which tables the record kernels keep in the LDS (what cannot fit is dropped)
        s = GpuSystem(c["box"], (1, 1, 1), c["dim"], 2, c["cellsz"], 0.5, 0.1)
        s.set_option("mode", mode)
        if lds >= 0:
            s.set_option("tile_rec_lds", lds)
        s.upload_tables(*c["tabs"])
        s.upload_atoms(c["pos"], c["mom"], c["masse"], c["sorte"])
        r = s.calc_forces(0)
        assert (s.get_info("list_mode"), s.get_info("pass1_form")) == ((1, 3) if mode else (0, 0))
        if mode and lds >= 0:
            assert s.get_info("rec_lds") & ~lds == 0
        d = s.download(fields=("kraft", "nummer", "eam_rho", "pot_eng"))
        s.close()
        k = np.argsort(d["nummer"])
        assert r["is_short"] == o.s.is_short, mode
        assert np.abs(d["kraft"][k] - fo).max() < 1e-9 * np.abs(fo).max(), mode
        assert np.abs(d["eam_rho"][k] - ro).max() < 1e-11 * np.abs(ro).max()
        assert np.abs(d["pot_eng"][k] - eo).max() < 1e-11 * np.abs(eo).max()
        assert abs(r["tot_pot_energy"] - o.s.tot_pot_energy) < 1e-11 * abs(o.s.tot_pot_energy)
        assert abs(r["virial"] - o.s.virial) < 1e-10 * abs(o.s.virial)


@pytest.mark.parametrize("ntypes,rows", [(3, 1.0), (3, 0.12), (4, 1.0), (4, 0.1)])
def test_three_and_four_types_vs_oracle(ntypes, rows):
    """Three and four atom types (the most the library takes: ntypes^2 <= 16 columns), every column with its own grid: list rows
    in three / four segments from one sweep of the list build per type (k_nbl_build_tile32<0>), the record kernels with full-size
    tables (through L2) and with tables small enough for the LDS, and the global-gather family, against the oracle."""
    import os
    import tempfile
    nb, a = (10, 10, 10), 4.05
    with tempfile.TemporaryDirectory() as d:
        potential.write_synthetic_eam_files_ntypes(d, ntypes, rows=rows)
        pair, c1 = potential.read_pot_table(os.path.join(d, "phi.pt"), ntypes * ntypes, True)
        rho, c2 = potential.read_pot_table(os.path.join(d, "rho.pt"), ntypes * ntypes, True)
        emb, _ = potential.read_pot_table(os.path.join(d, "F.pt"), ntypes, False)
    cellsz = (np.sqrt(max(c1, c2)) + 0.5) ** 2
    box = lattice.fcc_box(nb, a)
    pos = lattice.fcc_positions(nb, a)
    rng = np.random.default_rng(11 + ntypes)
    sorte = rng.choice(ntypes, size=len(pos), p=[0.5, 0.3, 0.2] if ntypes == 3 else [0.4, 0.3, 0.2, 0.1]).astype(np.int32)
    pos = pos + rng.normal(scale=0.04, size=pos.shape)
    masse = np.array([26.9815, 63.546, 58.69, 107.87])[sorte]
    mom = lattice.maxwell_momenta(len(pos), 1.0, 0.03, seed=3) * np.sqrt(masse)[:, None]
    dim = [g + 2 for g in global_cell_dim(box, cellsz)]
    o = po.OracleSystem(dim, 64, ntypes, box, (1, 1, 1), cellsz, 0.5, 0.1)
    for w, t in enumerate((pair, rho, emb)):
        o.set_table(w, t.as_dict())
    o.fill_from_positions(pos % np.diag(box), mom, masse, sorte)
    o.calc_forces(0)
    idn = np.argsort(o.gather_inner("nummer"))
    fo, ro, eo = o.gather_inner("kraft")[idn], o.gather_inner("rho")[idn], o.gather_inner("pot_eng")[idn]
    for mode, lds in ((0, -1), (1, -1), (1, 0)):
        s = GpuSystem(box, (1, 1, 1), dim, ntypes, cellsz, 0.5, 0.1)
        s.set_option("mode", mode)
        if lds >= 0:
            s.set_option("tile_rec_lds", lds)
        s.upload_tables(pair, rho, emb)
        s.upload_atoms(pos, mom, masse, sorte)
        r = s.calc_forces(0)
        assert (s.get_info("list_mode"), s.get_info("pass1_form")) == ((1, 3) if mode else (0, 0))
        if mode and lds < 0 and rows < 0.5:
            assert s.get_info("rec_lds") == 7                              # these tables all sit in the LDS
        d = s.download(fields=("kraft", "nummer", "eam_rho", "pot_eng"))
        s.close()
        k = np.argsort(d["nummer"])
        assert np.abs(d["kraft"][k] - fo).max() < 1e-9 * np.abs(fo).max(), (mode, lds)
        assert np.abs(d["eam_rho"][k] - ro).max() < 1e-11 * np.abs(ro).max()
        assert np.abs(d["pot_eng"][k] - eo).max() < 1e-11 * np.abs(eo).max()
        assert abs(r["tot_pot_energy"] - o.s.tot_pot_energy) < 1e-11 * abs(o.s.tot_pot_energy)
        assert abs(r["virial"] - o.s.virial) < 1e-10 * abs(o.s.virial)
    o.close()
