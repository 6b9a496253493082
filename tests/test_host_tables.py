"""Host-side table construction against the reference's own memory (CPU tests).
Row a2 of SURVEY 8: pot_table_t as left by read_pot_table / create_pot_table / init_threepoint
(imd_potential.c:161-466, 476-705, 1260-1276); the goldens hold the tables the reference binary built."""
import os
import tempfile

import numpy as np
import pytest

from imd_amd import potential
from tests import golden_util as gu


def test_lj_table_creation_bit_exact():
    """BASELINE config 1 (LJ, epsilon 0.392, sigma 2.62, r_cut 6.0, r_begin 1.5, pot_res 2000): the table
    create_lj_table builds equals the reference's pair_pot after create_pot_table bit for bit."""
    t = gu.table(gu.load("lj_nbl_2048"), "pair_pot")
    pt, r2_end = potential.create_lj_table(0.392, 2.62, 6.0, 1.5, 2000)
    assert r2_end == float(t["end"][0])
    for f in ("begin", "end", "step", "invstep"):
        assert np.array_equal(getattr(pt, f), t[f]), f
    assert pt.maxsteps == int(t["maxsteps"]) and pt.ncols == int(t["ncols"])
    assert np.array_equal(pt.table, t["table"])


@pytest.mark.parametrize("case,grids", [("eam_nbl_2048", "shared"), ("eam_nbl_2048_dg", "distinct")])
def test_eam_files_read_back_like_the_reference(case, grids):
    """write_synthetic_eam_files -> read_pot_table (format 2, tabulated in r^2 / rho, cut-off shift, two extrapolated
    rows) gives the tables the reference holds after reading the same files -- also when phi, rho and F come with three
    different headers (begin, end, step, row count), and synthetic_eam_tables (what bench.py and the size tests hand the
    device) holds the same bits."""
    G = gu.load(case)
    direct = potential.synthetic_eam_tables(grids=grids)
    for pt, key in zip(direct[:3], ("pair_pot", "rho_h_tab", "embed_pot")):
        t = gu.table(G, key)
        for f in ("begin", "end", "invstep"):
            assert np.array_equal(getattr(pt, f), t[f]), (key, f)
        assert np.array_equal(pt.table, t["table"]), key
    assert direct[3] == float(max(G["pair_pot_end"][0], G["rho_h_tab_end"][0]))
    with tempfile.TemporaryDirectory() as d:
        potential.write_synthetic_eam_files(d, grids=grids)
        for fn, key, radial in (("phi.pt", "pair_pot", True), ("rho.pt", "rho_h_tab", True), ("F.pt", "embed_pot", False)):
            pt, _ = potential.read_pot_table(os.path.join(d, fn), 1, radial)
            t = gu.table(G, key)
            for f in ("begin", "end", "step", "invstep"):
                assert np.array_equal(getattr(pt, f), t[f]), (key, f)
            assert np.array_equal(pt.table, t["table"]), key


def test_format1_pair_file_reads_back_like_the_reference():
    """read_pot_table1 (imd_potential.c:297-375): `r2 value` lines, end/len from the last non-zero value, cut-off shift,
    allocation in PSTEP blocks, then init_threepoint's two extrapolated rows.  The golden holds pair_pot as the
    reference built it from the same file (731 samples of the LJ table, written by write_pot_table1)."""
    G = gu.load("pair_fmt1")
    t = gu.table(G, "pair_pot")
    with tempfile.TemporaryDirectory() as d:
        lj, _ = potential.create_lj_table(0.392, 2.62, 6.0, 1.5, 731)
        r2 = lj.begin[0] + lj.step[0] * np.arange(731)
        potential.write_pot_table1(os.path.join(d, "pair1.pt"), r2, [lj.table[:731]])
        pt, cellsz = potential.read_pot_table(os.path.join(d, "pair1.pt"), 1, True)
    for f in ("begin", "end", "step", "invstep", "len"):
        assert np.array_equal(getattr(pt, f), t[f]), f
    assert pt.maxsteps == int(t["maxsteps"]) == 750 and pt.ncols == 1
    n = int(t["len"][0]) + 2                       # rows beyond len+2 are uninitialised heap in the reference
    assert np.array_equal(pt.table[:n], t["table"][:n])
    assert cellsz == float(t["end"][0])


def test_laser_setup_like_init_laser():
    """init_laser / get_surface (imd_laser.c:283-327, :117-282) from the inputs of the laser fixture: the derived
    globals equal the ones the reference held (probe record `laserpar` of tests/golden/ttm_laser.npz)."""
    from imd_amd import ttm as ttm_mod
    G = gu.load("ttm_laser")
    off = ttm_mod.get_surface(G["s0_ort"][:, 0], float(G["box"][0, 0]))
    assert off == float(G["laser_offset"])
    p = ttm_mod.init_laser(mu=0.05, sigma_e=40.0, sigma_t=0.15, t_0=0.3, direction=(1, 0, 0), offset=off)
    assert p["p_peak"] == float(G["laser_p_peak"]) and p["sigma_t_squared"] == float(G["laser_sigma_t_squared"])
    assert p["mu"] == float(G["laser_mu"]) and p["t_0"] == float(G["laser_t_0"]) and list(p["direction"]) == list(G["laser_dir"])


@pytest.mark.parametrize("ntypes", [3, 4])
def test_ntype_tables_round_trip(tmp_path, ntypes):
    """write_synthetic_eam_files_ntypes -> read_pot_table (format 2, imd_potential.c:394-466): ntypes^2 columns, every column with its
    own begin / end / step / length, the two pad rows from the three-point extrapolation; pair columns symmetric in (i, j)"""
    potential.write_synthetic_eam_files_ntypes(str(tmp_path), ntypes, rows=0.1)
    nc = ntypes * ntypes
    pair, c1 = potential.read_pot_table(str(tmp_path / "phi.pt"), nc, True)
    rho, c2 = potential.read_pot_table(str(tmp_path / "rho.pt"), nc, True)
    emb, _ = potential.read_pot_table(str(tmp_path / "F.pt"), ntypes, False)
    assert pair.ncols == rho.ncols == nc and emb.ncols == ntypes
    assert len(set(pair.len.tolist())) > 1 and len(set(rho.len.tolist())) > 1            # lengths differ between columns
    assert pair.maxsteps == int(pair.len.max()) and c1 == float(pair.end.max()) and c2 == float(rho.end.max())
    tp = pair.table.reshape(-1, nc)
    for i in range(ntypes):
        for j in range(ntypes):
            a, b = i * ntypes + j, j * ntypes + i
            assert pair.len[a] == pair.len[b] and pair.begin[a] == pair.begin[b] and pair.end[a] == pair.end[b]
            assert np.array_equal(tp[: pair.len[a] + 2, a], tp[: pair.len[b] + 2, b])
    for t in (pair, rho, emb):
        tt = t.table.reshape(-1, t.ncols)
        for c in range(t.ncols):
            n = int(t.len[c])
            assert abs(t.begin[c] + (n - 1) * t.step[c] - t.end[c]) < 1e-9 * abs(t.end[c])
            assert tt[n, c] == 3 * tt[n - 1, c] - 3 * tt[n - 2, c] + tt[n - 3, c]               # init_threepoint (imd_potential.c:1272)
            assert np.all(tt[n + 2:, c] == 0.0)
