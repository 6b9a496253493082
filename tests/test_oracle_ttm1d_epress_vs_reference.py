"""-DELECPRESS of the reference's 1D model (do_electronic_pressure, imd_ttm.c:553-623; the extra force in move_atoms_ttm,
imd_integrate.c:657-743) as restated in oracle/ttm1d_oracle.c, against a two-rank run of the reference built with the flag
(oracle/_ref/imd_mpi_eam_ttm1d_tmm_ep_nbl, fixture tests/golden/ttm1d_ep_slab.npz: laser on, a planar pressure table).  CPU only."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests import golden_util as gu
from tests.test_oracle_ttm1d_tmm_vs_reference import tmm_oracle
from tests.test_oracle_ttm1d_vs_reference import atoms, check_stage, load_stage


def ep_oracle(G):
    o = tmm_oracle(G)
    o.set_epress(dict(x=G["eos_r"], y=G["eos_t"], z=G["epress_z"]))
    return o


def ref_deriv(G, s):
    """the row as the two ranks hold it: each rank fills its own share (cells 0..12 / 13..25), inf where a cell has no atoms"""
    d0, d1 = G["e%d_epress_deriv_rank0" % s], G["e%d_epress_deriv_rank1" % s]
    half = len(d0) // 2
    return np.concatenate([d0[:half], d1[half:]])


@pytest.mark.parametrize("s", [1, 2, 3, 4, 8])
def test_pressure_and_its_gradient_equal_the_reference(s):
    G = gu.load("ttm1d_ep_slab")
    o = ep_oracle(G)
    load_stage(o, G, s, "substeps_done")
    # substeps_done is dumped after the EOS update of U, which does not touch what the pressure reads (dens, temp, natoms)
    o.elec_pressure(s)
    # the reference interpolates linearly on nn-c's triangles, the oracle in the grid cell: equal on a planar table up to rounding
    refp = G["e%d_epress_global" % s]
    assert np.array_equal(o.epress == 0, refp == 0) and np.abs(o.epress - refp).max() <= 1e-13 * refp.max()
    ref = ref_deriv(G, s)
    fin = np.isfinite(ref)
    assert np.array_equal(np.isinf(o.epress_deriv), np.isinf(ref))
    # a difference of neighbouring pressures: their rounding relative to the gradient of the deepest cells
    assert np.abs(o.epress_deriv[fin] - ref[fin]).max() <= 1e-12 * max(np.abs(ref[fin]).max(), 1e-300) + 1e-15
    if s > 1:
        assert np.isinf(ref).sum() == 1 and np.abs(ref[np.isfinite(ref)]).max() > 0.03      # the vacuum cell in front of the surface; the surface cells
    else:
        assert not ref.any()                                                                # nothing before step 2 (:614)
    o.close()


def test_trajectory_with_the_pressure_force():
    """calc_ttm + move_atoms_ttm of steps 0..3 from the reference's atoms of step 0 (forces of the reference at each step):
    the atoms of steps 1..4 -- the pressure force included from step 2 on, for the atoms whose owner rank also owns their slab"""
    G = gu.load("ttm1d_ep_slab")
    o = ep_oracle(G)
    box = np.diag(G["box"])
    edge = box[0] / 2
    worst = 0.0
    for s in (0, 1, 2, 3):
        ort, sorte, imp, mas, nn = atoms(G, s)
        o.calc(s, ort, sorte, imp, mas, nn)
        check_stage(o, G, s, "substeps_done", fields=("natoms", "temp", "U", "xi", "md_temp", "dens"), tol=1e-9)
        pos, mom, _ = o.move_atoms(ort, sorte, imp, G["s%d_kraft" % s], mas)
        a, b = np.argsort(G["s%d_nummer" % s]), np.argsort(G["s%d_nummer" % (s + 1)])
        dp = np.abs(mom[a] - G["s%d_impuls" % (s + 1)][b]).max(axis=1)
        # ownership is taken from the position; the reference's owner is the rank of the last re-binning: atoms that sit within
        # thermal reach of the rank boundary may be the other rank's
        far = np.abs(ort[a][:, 0] - edge) > 0.6
        assert dp[far].max() < 1e-11 * np.abs(G["s%d_impuls" % (s + 1)]).max(), (s, dp[far].max())
        worst = max(worst, dp.max())
        dx = pos[a] - G["s%d_ort" % (s + 1)][b]; dx[:, 1:] -= np.round(dx[:, 1:] / box[1:]) * box[1:]
        assert np.abs(dx[far]).max() < 1e-11
    # the force is there: without it step 3's momenta would be off by dt * 0.07 eV/A at the front surface
    o2 = tmm_oracle(G)
    ort, sorte, imp, mas, nn = atoms(G, 3)
    load_stage(o2, G, 3, "substeps_done")
    o2.t.cur = 0
    o2.xiarr[:] = G["t3_substeps_done_xiarr_global"]
    _, mom2, _ = o2.move_atoms(ort, sorte, imp, G["s3_kraft"], mas)
    a, b = np.argsort(G["s3_nummer"]), np.argsort(G["s4_nummer"])
    assert np.abs(mom2[a] - G["s4_impuls"][b]).max() > 1e-3
    o.close(); o2.close()
