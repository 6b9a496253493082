"""GPU parity tests (run with -m gpu on an MI355X): the HIP force path, called
through the C ABI, against (a) golden vectors dumped from the real reference
binary and (b) the CPU oracle on the same inputs.

Tolerances: neighbour lists bit-exact; forces 1e-9 relative to max|F|
(BASELINE.json north_star); energies / rho / dF 1e-11 relative."""
import numpy as np
import pytest

from imd_amd.system import GpuSystem
from tests import golden_util as gu

pytestmark = pytest.mark.gpu

FTOL = 1e-9


def _gpu_from_golden(G, step, nbl=True, ttm=False):
    pre = "s%d_" % step
    dim = G["cell_dim"] if nbl else G["cell_dim"] + 2
    s = GpuSystem(G["box"], G["pbc_dirs"], dim, int(G["ntypes"]), float(G["cellsz"]),
                  float(G["nbl_margin"]) if nbl else 0.0, float(G["timestep"]), count_ttm_neighbours=ttm)
    s.upload_tables(gu.table(G, "pair_pot"), gu.table(G, "rho_h_tab"), gu.table(G, "embed_pot"))
    cell_n = np.zeros(int(np.prod(dim)), np.int32)
    if nbl:
        cell_n[G["cells"]] = G[pre + "cell_n"][G["cells"]]
    else:                            # cell-pair build: no ghost layer in the reference's cell array, embed it
        d0 = G["cell_dim"]
        i, j, k = np.meshgrid(np.arange(d0[0]), np.arange(d0[1]), np.arange(d0[2]), indexing="ij")
        padded = ((i + 1) * dim[1] + (j + 1)) * dim[2] + (k + 1)
        cell_n[padded.ravel()] = G[pre + "cell_n"][: padded.size]
    s.upload_cells(cell_n, G[pre + "ort"], G[pre + "impuls"], G[pre + "masse"], G[pre + "sorte"].astype(np.int32),
                   G[pre + "nummer"])
    return s


def _relmax(a, b, floor=1e-300):
    return np.abs(a - b).max() / max(np.abs(b).max(), floor)


@pytest.mark.parametrize("case,step,eam", [("eam_nbl_2048", 0, True), ("eam_nbl_2048", 40, True),
                                           ("eam_nbl_2048_dg", 0, True), ("eam_nbl_2048_dg", 40, True),
                                           ("lj_nbl_2048", 20, False), ("eam_nbl_slab", 10, True),
                                           ("eam2_l12_nbl", 0, True), ("eam2_l12_nbl", 20, True),
                                           ("eam2_l12_coarse", 0, True), ("eam2_l12_coarse", 20, True)])
def test_forces_vs_reference_golden(case, step, eam):
    G = gu.load(case)
    s = _gpu_from_golden(G, step)
    r = s.calc_forces(step)
    if case.endswith("_dg"):             # one type, phi / rho / F on three different grids: the distinct-grids pass 1 with the embedding
        assert (s.get_info("pass1_form"), s.get_info("fused_embedding")) == (1, 1)     # pass folded in, the coefficient form of the force pass
        if "IMDGPU_TILE" not in __import__("os").environ:      # a forced 3 x 4 x 4 tile leaves the coefficient table no room: three-row form
            assert s.get_info("p3_coef_rows") > 0 and s.get_info("p3_standin_passes") == 0
    elif case.startswith("eam_nbl"):
        assert s.get_info("pass1_form") == 2
    d = s.download()
    pre = "s%d_" % step
    assert np.array_equal(d["nummer"], G[pre + "nummer"])
    assert _relmax(d["kraft"], G[pre + "kraft"], floor=1e-3) < FTOL     # step 0: perfect lattice, |F|~1e-14
    assert _relmax(d["pot_eng"], G[pre + "pot_eng"]) < 1e-11
    if eam:
        assert _relmax(d["eam_rho"], G[pre + "eam_rho"]) < 1e-11
        assert _relmax(d["eam_dF"], G[pre + "eam_dF"]) < 1e-11
    assert abs(r["tot_pot_energy"] - float(G[pre + "tot_pot_energy"])) < 1e-11 * abs(float(G[pre + "tot_pot_energy"]))
    assert abs(r["virial"] - float(G[pre + "virial"])) < 1e-10 * abs(float(G[pre + "virial"]))
    assert r["is_short"] == 0
    s.close()


@pytest.mark.parametrize("step", [0, 10])
def test_cellpair_entry_vs_reference_golden(step):
    """the reference built WITHOUT -DNBL (calc_forces -> do_forces / do_embedding_energy /
    do_forces_eam2 over the cell-pair list, imd_main_risc_3d.c:31-182, imd_forces.c:33, imd_forces_eam2.c:202,239).
    Device equivalent: zero list margin, i.e. all pairs with r2 < cellsz of the current arrangement."""
    G = gu.load("eam_cellpair_2048")
    s = _gpu_from_golden(G, step, nbl=False)
    r = s.calc_forces(step)
    d = s.download()
    pre = "s%d_" % step
    assert np.array_equal(d["nummer"], G[pre + "nummer"])
    assert _relmax(d["kraft"], G[pre + "kraft"], floor=1e-3) < FTOL
    assert _relmax(d["pot_eng"], G[pre + "pot_eng"]) < 1e-11
    assert _relmax(d["eam_rho"], G[pre + "eam_rho"]) < 1e-11
    assert abs(r["tot_pot_energy"] - float(G[pre + "tot_pot_energy"])) < 1e-11 * abs(float(G[pre + "tot_pot_energy"]))
    assert abs(r["virial"] - float(G[pre + "virial"])) < 1e-10 * abs(float(G[pre + "virial"]))
    s.close()


@pytest.mark.parametrize("case", ["eam_nbl_2048", "eam_nbl_2048_dg", "lj_nbl_2048", "eam_nbl_slab", "eam2_l12_nbl"])
def test_neighbour_list_bit_exact(case):
    """make_nblist on the reference's step-0 arrangement, exported in the
    reference's own tl/tb/cl_off format, equals the reference's list exactly."""
    G = gu.load(case)
    s = _gpu_from_golden(G, 0)
    s.make_nblist()
    tl, tb, cl_off = s.export_nblist()
    assert np.array_equal(cl_off, G["s0_cl_off"])
    assert np.array_equal(tl, G["s0_tl"])
    assert np.array_equal(tb, G["s0_tb"])
    entries, mx = s.nblist_stats()
    assert entries == 2 * len(G["s0_tb"])          # device list = symmetrised reference list
    d = s.download(fields=("nbl_pos",))
    assert np.array_equal(d["nbl_pos"], G["s0_ort"])
    s.close()


@pytest.mark.parametrize("case", ["eam_nbl_2048", "eam_nbl_2048_dg"])
def test_forces_vs_oracle_same_inputs(case):
    G = gu.load(case)
    o = gu.oracle_from_golden(G, 20)
    o.s.have_valid_nbl = 0
    o.calc_forces(20)
    s = _gpu_from_golden(G, 20)
    r = s.calc_forces(20)
    d = s.download()
    assert _relmax(d["kraft"], o.gather_inner("kraft")) < FTOL
    assert _relmax(d["eam_rho"], o.gather_inner("rho")) < 1e-11
    assert abs(r["tot_pot_energy"] - o.s.tot_pot_energy) < 1e-11 * abs(o.s.tot_pot_energy)
    s.close()


def test_rebinning_unsorted_upload_matches_by_id():
    """upload_atoms (device fix_cells: wrap, bin, sort) gives the same per-atom
    results as the reference's arrangement, compared by atom number."""
    G = gu.load("eam_nbl_2048")
    rng = np.random.default_rng(1)
    n = len(G["s20_nummer"])
    perm = rng.permutation(n)
    ort = G["s20_ort"][perm] + np.array([G["box"][0, 0], -G["box"][1, 1], 0.0]) * (rng.random((n, 1)) < 0.3)
    s = GpuSystem(G["box"], G["pbc_dirs"], G["cell_dim"], 1, float(G["cellsz"]), float(G["nbl_margin"]),
                  float(G["timestep"]))
    s.upload_tables(gu.table(G, "pair_pot"), gu.table(G, "rho_h_tab"), gu.table(G, "embed_pot"))
    s.upload_atoms(ort, G["s20_impuls"][perm], G["s20_masse"][perm], None, G["s20_nummer"][perm])
    s.calc_forces(0)
    d = s.download()
    order = np.argsort(d["nummer"])
    gorder = np.argsort(G["s20_nummer"])
    assert np.array_equal(d["nummer"][order], G["s20_nummer"][gorder])
    assert _relmax(d["kraft"][order], G["s20_kraft"][gorder]) < FTOL
    assert _relmax(d["pot_eng"][order], G["s20_pot_eng"][gorder]) < 1e-11
    # atoms are inside the box and in the cell that contains them
    assert (d["ort"] >= 0).all() and (d["ort"] < np.diag(G["box"])).all()
    s.close()


@pytest.mark.parametrize("case,last", [("eam_nbl_2048", 40), ("eam_nbl_2048_dg", 40), ("eam_nbl_slab", 10), ("eam2_l12_nbl", 20)])
def test_trajectory_vs_reference(case, last):
    """Device-resident MD loop (forces, leap-frog, list check, re-binning) from
    the reference's step-0 state reproduces the reference's trajectory."""
    G = gu.load(case)
    s = _gpu_from_golden(G, 0)
    for st in range(last):
        s.md_step(st)
    r = s.calc_forces(last)
    d = s.download()
    pre = "s%d_" % last
    order = np.argsort(d["nummer"])
    gorder = np.argsort(G[pre + "nummer"])
    box = np.diag(G["box"])
    dx = d["ort"][order] - G[pre + "ort"][gorder]
    dx -= np.round(dx / box) * box * G["pbc_dirs"]
    assert np.abs(dx).max() < 1e-9
    assert _relmax(d["impuls"][order], G[pre + "impuls"][gorder]) < 1e-9
    assert _relmax(d["kraft"][order], G[pre + "kraft"][gorder]) < 1e-8
    assert abs(r["tot_pot_energy"] - float(G[pre + "tot_pot_energy"])) < 1e-9 * abs(float(G[pre + "tot_pot_energy"]))
    if case.startswith("eam_nbl_2048"):
        assert s.get_sums()["nbl_count"] == int(G[pre + "nbl_count"])
    s.close()


def test_shared_grid_tables_through_the_distinct_grids_kernel():
    """option no_merge: pair_pot and rho_h_tab of the shared-grid fixture kept apart, so the SAME inputs go through the
    distinct-grids form of pass 1 (two indices, six 8-byte row reads) -- equal to the interleaved form to rounding, and to the golden"""
    G = gu.load("eam_nbl_2048")
    out = []
    for nm in (0, 1):
        s = GpuSystem(G["box"], G["pbc_dirs"], G["cell_dim"], 1, float(G["cellsz"]), float(G["nbl_margin"]), float(G["timestep"]))
        s.set_option("no_merge", nm)
        s.upload_tables(gu.table(G, "pair_pot"), gu.table(G, "rho_h_tab"), gu.table(G, "embed_pot"))
        cell_n = np.zeros(int(np.prod(G["cell_dim"])), np.int32)
        cell_n[G["cells"]] = G["s40_cell_n"][G["cells"]]
        s.upload_cells(cell_n, G["s40_ort"], G["s40_impuls"], G["s40_masse"], G["s40_sorte"].astype(np.int32), G["s40_nummer"])
        r = s.calc_forces(40)
        assert s.get_info("pass1_form") == (1 if nm else 2)
        out.append((s.download(), r))
        s.close()
    (d0, r0), (d1, r1) = out
    assert _relmax(d1["kraft"], G["s40_kraft"]) < FTOL
    assert _relmax(d1["kraft"], d0["kraft"]) < 1e-12 and _relmax(d1["eam_rho"], d0["eam_rho"]) < 1e-13
    assert _relmax(d1["pot_eng"], d0["pot_eng"]) < 1e-13 and _relmax(d1["eam_dF"], d0["eam_dF"]) < 1e-12
    assert abs(r1["tot_pot_energy"] - r0["tot_pot_energy"]) < 1e-12 * abs(r0["tot_pot_energy"])
    assert abs(r1["virial"] - r0["virial"]) < 1e-11 * abs(r0["virial"])


@pytest.mark.parametrize("case,rec,lds", [("eam2_l12_nbl", 0, -1), ("eam2_l12_coarse", 0, -1), ("eam2_l12_nbl", 1, 0), ("eam2_l12_coarse", 1, 0),
                                          ("eam2_l12_coarse", 1, 1), ("eam2_l12_coarse", 1, 2), ("eam2_l12_coarse", 1, 4), ("eam2_l12_coarse", 1, 7)])
def test_two_types_other_kernel_choice(case, rec, lds):
    """Several atom types, every form the library has, same goldens.  Option tile_rec = 0: the older kernels -- tables beyond the LDS
    select the global-gather family (round 4), tables that fit the tile kernels with the full tables and a type per slot in the LDS.
    tile_rec = 1 (default): the record kernels on type-segmented list rows; option tile_rec_lds forces which tables they keep in the
    LDS (1 pair_pot and 2 rho_h_tab in pass 1, 4 rho_h_tab in pass 3), the others are coefficient records through L2."""
    G = gu.load(case)
    for step in (0, 20):
        s = _gpu_from_golden(G, step)
        s.set_option("tile_rec", rec)
        if lds >= 0:
            s.set_option("tile_rec_lds", lds)
        r = s.calc_forces(step)
        if rec == 0:
            assert (s.get_info("list_mode"), s.get_info("pass1_form")) == ((0, 0) if case == "eam2_l12_nbl" else (1, 0))
        else:
            assert (s.get_info("list_mode"), s.get_info("pass1_form"), s.get_info("rec_lds")) == (1, 3, lds)
        d = s.download()
        pre = "s%d_" % step
        assert _relmax(d["kraft"], G[pre + "kraft"], floor=1e-3) < FTOL
        assert _relmax(d["pot_eng"], G[pre + "pot_eng"]) < 1e-11
        assert _relmax(d["eam_rho"], G[pre + "eam_rho"]) < 1e-11 and _relmax(d["eam_dF"], G[pre + "eam_dF"]) < 1e-11
        assert abs(r["tot_pot_energy"] - float(G[pre + "tot_pot_energy"])) < 1e-11 * abs(float(G[pre + "tot_pot_energy"]))
        assert abs(r["virial"] - float(G[pre + "virial"])) < 1e-10 * abs(float(G[pre + "virial"]))
        s.close()


def test_check_nblist_threshold():
    G = gu.load("eam_nbl_2048")
    s = _gpu_from_golden(G, 0)
    s.calc_forces(0)
    assert s.check_nblist() is True
    for st in range(40):
        s.md_step(st)
    assert s.get_sums()["nbl_count"] >= 2
    s.close()


@pytest.mark.parametrize("mode,case", [(0, "eam_nbl_2048"), (1, "eam_nbl_2048"), (0, "eam_nbl_2048_dg"), (1, "eam_nbl_2048_dg"),
                                       (0, "eam2_l12_nbl"), (1, "eam2_l12_nbl"),
                                       (0, "eam2_l12_coarse"), (1, "eam2_l12_coarse")])
def test_both_kernel_modes_match_reference(mode, case):
    G = gu.load(case)
    s = _gpu_from_golden(G, 20)
    s.set_option("mode", mode)
    r = s.calc_forces(20)
    # no silent fall-back to the other kernel family: several atom types run on the record kernels of the tile family (pass1_form 3),
    # with the tables that fit in the LDS (the coarse ones: all; 2 x 4 columns x 4002 rows: pair_pot in pass 1) and the others through L2
    assert s.get_info("list_mode") == mode
    if mode == 1:
        assert s.get_info("pass1_form") == {"eam_nbl_2048": 2, "eam_nbl_2048_dg": 1, "eam2_l12_nbl": 3, "eam2_l12_coarse": 3}[case]
        assert s.get_info("rec_lds") == {"eam_nbl_2048": 0, "eam_nbl_2048_dg": 0, "eam2_l12_nbl": 1, "eam2_l12_coarse": 7}[case]
    d = s.download()
    assert _relmax(d["kraft"], G["s20_kraft"]) < FTOL
    assert _relmax(d["eam_rho"], G["s20_eam_rho"]) < 1e-11
    assert abs(r["tot_pot_energy"] - float(G["s20_tot_pot_energy"])) < 1e-11 * abs(float(G["s20_tot_pot_energy"]))
    s.close()


def test_large_system_properties():
    """Config-2 size (1M atoms, EAM): size-independent properties instead of an oracle run --
    the two independent kernel families agree, total force vanishes (Newton 3), the
    symmetric list has an even number of entries, energy is conserved over a short run."""
    from imd_amd import lattice, potential
    from imd_amd.system import global_cell_dim
    nb = (64, 64, 64)
    box = lattice.fcc_box(nb)
    pos = lattice.fcc_positions(nb)
    rng = np.random.default_rng(3)
    pos = pos + rng.normal(scale=0.05, size=pos.shape)
    mom = lattice.maxwell_momenta(len(pos), 26.9815, 0.02)
    pair, rho, emb, r2cut = potential.synthetic_eam_tables()
    cellsz = (np.sqrt(r2cut) + 0.5) ** 2
    dim = [g + 2 for g in global_cell_dim(box, cellsz)]
    res = {}
    for mode in (0, 1):
        s = GpuSystem(box, (1, 1, 1), dim, 1, cellsz, 0.5, 0.1)
        s.set_option("mode", mode)
        s.upload_tables(pair, rho, emb)
        s.upload_atoms(pos, mom, 26.9815)
        r = s.calc_forces(0)
        assert s.get_info("list_mode") == mode
        d = s.download(fields=("kraft", "nummer", "pot_eng", "eam_rho"))
        o = np.argsort(d["nummer"])
        res[mode] = (d["kraft"][o], d["pot_eng"][o], r)
        ent, mx = s.nblist_stats()
        assert ent % 2 == 0 and mx < 120
        if mode == 1:
            etot = []
            for st in range(30):
                s.md_step(st)
                if st in (5, 29):
                    g = s.get_sums()       # Epot of this step's forces + leap-frog mean kinetic energy
                    etot.append((g["tot_pot_energy"] + g["tot_kin_energy"], g["tot_kin_energy"]))
            assert abs(etot[1][0] - etot[0][0]) < 2e-3 * etot[0][1]      # NVE: drift << kinetic energy
        s.close()
    f0, e0, r0 = res[0]; f1, e1, r1 = res[1]
    assert np.abs(f0 - f1).max() < 1e-11 * np.abs(f0).max()
    assert np.abs(e0 - e1).max() < 1e-12 * np.abs(e0).max()
    assert abs(r0["tot_pot_energy"] - r1["tot_pot_energy"]) < 1e-12 * abs(r0["tot_pot_energy"])
    assert np.abs(f1.sum(axis=0)).max() < 1e-7            # sum of all forces ~ 0 (1M atoms, |F|~0.5)


def test_bench_size_properties_with_ttm():
    """BASELINE config 3 at full size (136^3 unit cells = 10 061 824 atoms, EAM + 3D TTM fd_ext 2 2 2, 100 sub-steps):
    the sequence bench.py times, do_ADV included -- properties that need no oracle run: the two kernel families agree on every force, the total force vanishes, the
    symmetric list is even; after device-resident MD+TTM steps every atom sits in exactly one FD cell (sum natoms = N),
    the lattice temperature of the FD cells averages to 2/3 of the kinetic energy per atom (in the frame of each
    cell's centre of mass, hence a little below), electron temperatures stay within the initial thermal spread,
    and the coupling xi is finite everywhere."""
    from imd_amd import lattice, potential, ttm as ttm_mod
    from imd_amd.system import global_cell_dim
    nb = (136, 136, 136)
    box = lattice.fcc_box(nb)
    pos = lattice.fcc_positions(nb)
    n = len(pos)
    assert n == 10061824
    mom = lattice.maxwell_momenta(n, 26.9815, 0.0258, seed=12345)
    pair, rho, emb, r2cut = potential.synthetic_eam_tables()
    cellsz = (np.sqrt(r2cut) + 0.5) ** 2
    gd = global_cell_dim(box, cellsz, (2, 2, 2))
    dim = [g + 2 for g in gd]
    res = {}
    for mode in (0, 1):
        s = GpuSystem(box, (1, 1, 1), dim, 1, cellsz, 0.5, 0.1, count_ttm_neighbours=True)
        s.set_option("mode", mode)
        s.upload_tables(pair, rho, emb)
        s.upload_atoms(pos, mom, 26.9815)
        s.calc_forces(0)
        d = s.download(fields=("kraft", "nummer", "numneighs"))
        o = np.argsort(d["nummer"])
        res[mode] = (d["kraft"][o], d["numneighs"][o])
        ent, mx = s.nblist_stats()
        assert ent % 2 == 0
        if mode == 1:
            s.set_option("ttm_adv", 1)             # the benchmarked sequence: calc_ttm with do_ADV(1.0) after update_fd (bench.py's default)
            ttm_mod.attach_default(s, fd_ext=(2, 2, 2), diff_substeps=100, fd_min_atoms=5)
            for st in range(4):
                s.md_step(st)
            lat = s.ttm_download(fields=("natoms", "natoms_old", "md_temp", "temp", "xi", "U", "flux"))
            inner = (slice(1, -1),) * 3
            assert int(lat["natoms"][inner].sum()) == n
            # do_ADV's inputs and outputs: every atom that changed its FD cell was counted once as an arrival (so the arrivals
            # cover every cell's net gain), atoms are conserved between the two counts, the transported energy stays finite
            # and positive where there are atoms
            assert int(lat["natoms_old"][inner].sum()) == n
            gain = np.maximum(lat["natoms"][inner] - lat["natoms_old"][inner], 0)
            assert (lat["flux"][inner] >= 0).all() and (lat["flux"][inner].sum(axis=-1) >= gain).all()
            assert np.isfinite(lat["U"][inner]).all() and (lat["U"][inner][lat["natoms"][inner] >= 5] > 0).all()
            ekin = s.get_sums()["tot_kin_energy"] / n
            tl = lat["md_temp"][inner]
            assert 0.90 * (2.0 / 3.0) * ekin < tl.mean() < 1.02 * (2.0 / 3.0) * ekin
            te = lat["temp"][inner]
            assert te.min() > 0.5 * tl.mean() and te.max() < 1.5 * tl.mean()
            assert np.isfinite(lat["xi"]).all()
        s.close()
    f0, f1 = res[0][0], res[1][0]
    assert np.abs(f0 - f1).max() < 1e-11 * max(np.abs(f0).max(), 1e-3)
    assert np.array_equal(res[0][1], res[1][1])
    assert np.abs(f1.sum(axis=0)).max() < 1e-6


@pytest.mark.parametrize("tile", ["1,1,2", "3,4,4"])
def test_parity_cases_under_other_tile_shapes(tile):
    """the force / list / trajectory parity cases again with the tile shape forced (IMDGPU_TILE): tiny tiles (many tiles per
    workgroup, every wave stages ahead) and tiles with more own atoms than threads (a thread takes a second atom: the work a
    force pass defers to the next tile's prologue -- embedding look-up, force read-modify-write -- is completed in between)"""
    import os
    import subprocess
    import sys
    env = dict(os.environ, IMDGPU_TILE=tile)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k",
                        "forces_vs_reference_golden or neighbour_list_bit_exact or trajectory_vs_reference or forces_vs_oracle or distinct_grids_kernel"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


def test_experiment_switches_do_not_exist_in_the_product_build():
    """IMDGPU_DBG=1 used to make the force kernels skip their neighbour loops (wrong physics, status 0).  The
    default build does not read experiment names from the environment at all (exp_env, imdgpu_internal.h): with the whole set
    in the environment the golden forces, the list and a TTM trajectory still come out, and get_info("experiments") is 0."""
    import os
    import subprocess
    import sys
    G = gu.load("eam_nbl_2048")
    s = _gpu_from_golden(G, 0)
    assert s.get_info("experiments") == 0
    s.close()
    env = dict(os.environ, IMDGPU_DBG="1", IMDGPU_TTM_PERSIST="0", IMDGPU_NBL32="0", IMDGPU_P3_COEF="0", IMDGPU_FUSE_EMB="0",
               IMDGPU_FAST_CAP="0", IMDGPU_TILE_GRID="3", IMDGPU_TTM_REMOTE_COARSE="1", IMDGPU_DYN_EXP="2", IMDGPU_NO_SPECULATION="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_forces.py"), os.path.join(root, "tests", "test_gpu_ttm.py"),
                        "-x", "-q", "-m", "gpu", "-k",
                        "(forces_vs_reference_golden and eam_nbl_2048) or (neighbour_list_bit_exact and eam_nbl_2048) or (md_trajectory and ttm_2048)"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
