"""Pin the CPU oracle (oracle/imd_oracle.c) against the REAL reference:
the fixtures in tests/golden/ are dumps of the reference binary's memory
(tests/golden/make_golden.py).  Everything here is bit-exact."""
import numpy as np
import pytest

from tests import golden_util as gu


def _check_outputs(s, G, step, eam=True):
    pre = "s%d_" % step
    assert np.array_equal(s.gather_inner("kraft"), G[pre + "kraft"])
    assert np.array_equal(s.gather_inner("pot_eng"), G[pre + "pot_eng"])
    if eam:
        assert np.array_equal(s.gather_inner("rho"), G[pre + "eam_rho"])
        assert np.array_equal(s.gather_inner("dF"), G[pre + "eam_dF"])
    assert s.s.tot_pot_energy == float(G[pre + "tot_pot_energy"])
    assert s.s.virial == float(G[pre + "virial"])


@pytest.mark.parametrize("case,eam", [("eam_nbl_2048", True), ("eam_nbl_2048_dg", True), ("lj_nbl_2048", False), ("eam_nbl_slab", True),
                                      ("eam2_l12_nbl", True), ("eam2_l12_coarse", True)])
def test_list_and_forces_at_step0(case, eam):
    G = gu.load(case)
    s = gu.oracle_from_golden(G, 0)
    s.calc_forces(0)                       # builds ghost layer + Verlet list itself
    assert np.array_equal(s.cl_off[:-1], G["s0_cl_off"])
    assert np.array_equal(s.tl, G["s0_tl"])
    assert np.array_equal(s.tb, G["s0_tb"])         # neighbour indices bit-exact
    assert np.array_equal(s.n, G["s0_cell_n"])      # ghost occupancy (AR: x=0 wall empty)
    _check_outputs(s, G, 0, eam)


def test_cell_stencil_matches_reference():
    G = gu.load("eam_nbl_slab")            # open x boundary: -1 slots present
    s = gu.oracle_from_golden(G, 0)
    assert np.array_equal(s.inner, G["cells"])
    assert np.array_equal(s.nq, G["cnbrs"][:, 1:])
    assert (G["cnbrs"][:, 1:] < 0).any()


@pytest.mark.parametrize("case,eam,last", [("eam_nbl_2048", True, 40), ("eam_nbl_2048_dg", True, 40), ("lj_nbl_2048", False, 20),
                                           ("eam_nbl_slab", True, 10), ("eam2_l12_nbl", True, 20)])
def test_trajectory_bit_exact(case, eam, last):
    """calc_forces -> move_atoms_nve -> check_nblist for `last` steps, incl. list
    rebuilds with atoms changing cells (fix_cells slot order)."""
    G = gu.load(case)
    s = gu.oracle_from_golden(G, 0)
    steps = [int(x) for x in G["steps"]]
    for st in range(last + 1):
        s.calc_forces(st)
        if st in steps:
            pre = "s%d_" % st
            assert s.s.nbl_count == int(G[pre + "nbl_count"])
            assert np.array_equal(s.gather_inner("nummer"), G[pre + "nummer"])   # same slots
            assert np.array_equal(s.gather_inner("ort"), G[pre + "ort"])
            assert np.array_equal(s.gather_inner("impuls"), G[pre + "impuls"])
            _check_outputs(s, G, st, eam)
        s.move_atoms_nve()
        s.check_nblist()
    if case == "eam_nbl_2048":
        assert int(G["s40_nbl_count"]) >= 2          # the fixture does exercise a rebuild


def test_rebuilt_list_bit_exact():
    G = gu.load("eam_nbl_2048")
    s = gu.oracle_from_golden(G, 40)
    # list at step 40 is older than step 40; rebuild from the stored reference positions
    for c in s.inner:
        s.ort[c, : s.n[c]] = s.nbl_pos[c, : s.n[c]]
    s.send_cells(0, 0)
    s.make_nblist()
    assert np.array_equal(s.tl, G["s40_tl"])
    assert np.array_equal(s.tb, G["s40_tb"])
