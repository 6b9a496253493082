"""The transfer-matrix laser source of the 1D two-temperature model (imd_tmm.c: tmm_init, do_tmm, tmm_get_epsilon, Runge5)
as restated in oracle/ttm1d_oracle.c against the reference's OWN functions: fixture tests/golden/ttm1d_tmm_slab.npz, a two-rank
run of oracle/_ref/imd_mpi_eam_ttm1d_tmm_nbl (the configuration the reference's Makefile target imd_mpi_eam_ttm_tmm_nbl names)
with do_tmm at the top of every sub-step as in calc_ttm, on synthetic input tables (imd_amd/tmm.py).  CPU only."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests import golden_util as gu
from tests.test_oracle_ttm1d_vs_reference import atoms, check_stage, load_stage, oracle_from

TMM_STEPS = (2, 3, 8, 12)


def tmm_oracle(G):
    o = oracle_from(G)
    o.tmm_init(float(G["tmm_I0"]), float(G["tmm_lam"]), float(G["tmm_sigma_t"]), float(G["tmm_t0"]), G["tmm_eps_bb"], G["tmm_K12"],
               dict(x=G["eos_r"], y=G["eos_t"], z=G["tmm_phase_z"]), abs_threshold=float(G["tmm_abs_threshold"]))
    return o


def _rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-300)


def test_tmm_init_constants():
    G = gu.load("ttm1d_tmm_slab")
    o = tmm_oracle(G)
    assert o.t.k0 == float(G["tmm_k0"]) and o.t.omega_laser == float(G["tmm_omega_laser"])
    o.close()


@pytest.mark.parametrize("s", TMM_STEPS)
def test_do_tmm_equals_the_reference(s):
    """the first do_tmm of an MD step from the reference's lattice in front of it (stage `cfl`): permittivity of every cell
    (wide-range model with the interband term switched off where the phase table says molten), active cells, reflectivity /
    transmission / absorption of the stack, the absorbed power per cell down to the absorption depth, node.source of both ranks"""
    G = gu.load("ttm1d_tmm_slab")
    o = tmm_oracle(G)
    load_stage(o, G, s, "cfl")
    o.t.tmm_time = float(G["m%d_tmm_time" % s])
    o.do_tmm(s)
    q = "m%d_sub0_" % s
    sc = G[q + "scalars"]
    assert o.t.laser_active == int(sc[5]) == 1
    assert abs(o.t.I_t - sc[0]) <= 1e-15 * sc[0]
    act = G[q + "active"].astype(int)
    assert np.array_equal(o.tmm_active, act) and act.sum() > 15
    # eps_real of INACTIVE cells: 1 on rank 0's share, 0 beyond (the correction loop runs over the solving rank's own cells, :264-273)
    assert np.array_equal(o.eps_re, G[q + "eps_re"]) and np.array_equal(o.eps_im, G[q + "eps_im"])
    assert (o.t.refl, o.t.trans, o.t.absorbed) == (sc[2], sc[3], sc[4])
    ecut = int(o.t.ecut)
    assert 13 < ecut < 22                                   # the absorption depth lies on the second rank's share of the row
    assert np.array_equal(o.qabs[:ecut], G[q + "qabs"][:ecut]) and (o.qabs[:ecut][act[:ecut] == 1] > 0).all()
    assert np.array_equal(o.source[1:-1][:ecut], G[q + "source"][:ecut])
    o.close()


def test_interband_term_follows_the_phase_table():
    """cells whose ion temperature puts them beyond phase 2 lose the interband part of the permittivity (imd_tmm.c:541-544):
    the fixture has both kinds of cells"""
    G = gu.load("ttm1d_tmm_slab")
    q, t = "m2_sub0_", "t2_cfl_"
    act = G[q + "active"] == 1
    Ti = G[t + "md_temp"] * 11604.52500617
    molten = act & (Ti > 1160.0)
    solid = act & (Ti < 1000.0)
    assert molten.sum() >= 1 and solid.sum() >= 3
    assert G[q + "eps_im"][molten].max() < 0.8 * G[q + "eps_im"][solid].min()


def test_trajectory_with_the_laser_from_step_0():
    """calc_ttm of steps 0..3 on the reference's atoms of those steps, the laser switching on at step 2 (do_tmm does nothing
    before): lattice after the loop, absorbed energy, tmm_time and the source the LAST do_tmm of each step left"""
    G = gu.load("ttm1d_tmm_slab")
    o = tmm_oracle(G)
    for s in (0, 1, 2, 3):
        assert abs(o.t.tmm_time - float(G["m%d_tmm_time" % s])) <= 1e-12 * max(float(G["m%d_tmm_time" % s]), 1e-30)
        nsub = o.calc(s, *atoms(G, s))
        assert nsub == int(G["m%d_nsub" % s])
        check_stage(o, G, s, "substeps_done", fields=("natoms", "temp", "temp2", "U", "U2", "xi", "md_temp", "dens", "fd_k", "fd_g", "Ce"), tol=1e-9)
        q = "m%d_sub%d_" % (s, nsub - 1)
        ref_src = G["t%d_substeps_done_source" % s]
        assert _rel(o.source[1:-1], ref_src) <= 1e-9 if ref_src.max() > 0 else not o.source.any()
        # Eabs_local is dumped right after the last do_tmm: the last do_DIFF's share is still to come
        tau = o.timestep / nsub
        last = (ref_src * o.t.vol * tau)[: int(o.t.last_active) + 1].sum() if s >= 2 else 0.0
        assert abs(o.t.eabs - (float(G[q + "eabs"]) + last)) <= 1e-9 * max(o.t.eabs, 1e-30), (s, o.t.eabs, float(G[q + "eabs"]), last)
    assert o.t.eabs > 1.0 and o.temp[1:-1].max() > 1.05 * 0.09                 # the electrons at the surface did heat up
    o.close()
