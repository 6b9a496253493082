"""The 1D-TTM oracle (oracle/ttm1d_oracle.c: imd_ttm.c's own variant -- position-binned slabs, global slab arrays, RHOMIN,
the static virtual lattice, sub-step count from the stability limit) against the reference's own functions: fixture
tests/golden/ttm1d_slab.npz, a two-rank run of oracle/_ref/imd_mpi_eam_ttm1d_nbl with do_ADV on, dumped stage by stage by
oracle/ref_probe_ttm1d.c.  CPU only."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests import golden_util as gu

STAGE_FIELDS = ("natoms", "natoms_old", "temp", "xi", "md_temp", "dens", "fd_k", "fd_g", "Z", "Ce", "ne", "U", "temp2", "U2")


def oracle_from(G):
    return po.OracleTTM1D(int(G["ttm_global_fd_dim"]), G["box"], int(G["ref_ranks"]), int(G["ttm_vlatdim"]), int(G["ttm_vlatbuffer"]),
                          float(G["ttm_vlatdens"]), int(G["ttm_fd_min_atoms"]), int(G["ttm_fd_n_timesteps"]), float(G["ttm_atomic_weight"]),
                          float(G["ttm_atomic_charge"]), float(G["pair_pot_end"][0]), float(G["timestep"]),
                          dict(x=G["eos_r"], y=G["eos_t"], z=G["eos_cve"]), dict(x=G["eos_r"], y=G["eos_ts"], z=G["eos_ee"]))


def load_stage(o, G, s, tag):
    """put the reference's lattice of stage `tag` into the oracle (global row -> index g + 1)"""
    q = "t%d_%s_" % (s, tag)
    sl = slice(1, -1)
    o.natoms[sl] = G[q + "natoms"].astype(np.int32); o.natoms_old[sl] = G[q + "natoms_old"].astype(np.int32)
    o.t.cur = 0; o.t.vcur = 0
    o._temp[0][sl] = G[q + "temp"]; o._temp[1][sl] = G[q + "temp2"]; o._U[0][sl] = G[q + "U"]; o._U[1][sl] = G[q + "U2"]
    for f in ("xi", "md_temp", "source", "dens", "fd_k", "fd_g", "Z", "Ce", "ne"):
        getattr(o, f)[sl] = G[q + f]
    o.fluxl[:] = G[q + "fluxfromleft"].astype(np.int32); o.fluxr[:] = G[q + "fluxfromright"].astype(np.int32)
    o.natomsglobal[:] = G[q + "natomsglobal"].astype(np.int32)
    for d, k in enumerate(("vcomxglobal", "vcomyglobal", "vcomzglobal")):
        o.vcom[d][:] = G[q + k]
    o._vt[0][:] = G[q + "vlat_temp"]; o._vm[0][:] = G[q + "vlat_md_temp"]
    for f, k in (("vk", "fd_k"), ("vg", "fd_g"), ("vCe", "Ce"), ("vZ", "Z"), ("vne", "ne")):
        getattr(o, f)[:] = G[q + "vlat_" + k]
    sc = G[q + "scalars"]
    o.t.last_active, o.t.vlat_proc, o.t.diff_substeps, o.t.max_dt_ttm = int(sc[0]), int(sc[1]), int(sc[2]), float(sc[3])


def check_stage(o, G, s, tag, fields=STAGE_FIELDS, tol=1e-12, active_only=()):
    q = "t%d_%s_" % (s, tag)
    sl = slice(1, -1)
    got = dict(natoms=o.natoms[sl], natoms_old=o.natoms_old[sl], temp=o.temp[sl], temp2=o.temp2[sl], U=o.U[sl], U2=o.U2[sl])
    for f in ("xi", "md_temp", "dens", "fd_k", "fd_g", "Z", "Ce", "ne"):
        got[f] = getattr(o, f)[sl]
    for f in fields:
        ref = G[q + f]
        if f in ("natoms", "natoms_old"):
            assert np.array_equal(got[f], ref.astype(int)), (s, tag, f, got[f], ref)
        else:
            scale = max(np.abs(ref).max(), 1e-300)
            assert np.abs(got[f] - ref).max() <= tol * scale, (s, tag, f, np.abs(got[f] - ref).max() / scale)


def atoms(G, s):
    pre = "s%d_" % s
    return G[pre + "ort"], G[pre + "sorte"].astype(np.int32), G[pre + "impuls"], G[pre + "masse"], G[pre + "numneighs"]


def test_update_fd_equals_the_reference_slab_sums():
    """atoms of the dumped steps (positions, momenta, neighbour counts as the reference held them) -> natoms per slab, the
    global v_com arrays, md_temp, density, the deactivation behind the last active cell (-1), the flux counters"""
    G = gu.load("ttm1d_slab")
    o = oracle_from(G)
    steps = [int(v) for v in G["steps"]]
    # the flux counters compare the slab of an atom with its slab at the PREVIOUS step: only consecutive dumped steps qualify
    for s in steps:
        if s > 0 and (s - 1) in steps:
            o.update_fd(s - 1, *atoms(G, s - 1))            # sets fdi from the positions of step s - 1
        elif s > 0:
            o.fdi = None
        load_stage(o, G, s, "substeps_done") if False else None
        # state before update_fd of step s = the end of step s - 1; natoms_old / last_active come from there: take them from the golden
        if s > 0:
            prev = steps[steps.index(s) - 1]
            load_stage(o, G, prev, "substeps_done")
            if prev != s - 1:                                # natoms of step s - 1 is not in the fixture: natoms_old cannot be checked
                pass
        o.update_fd(s, *atoms(G, s))
        q = "t%d_update_fd_" % s
        assert np.array_equal(o.natomsglobal, G[q + "natomsglobal"].astype(int))
        assert np.array_equal(o.natoms[1:-1], G[q + "natoms"].astype(int))
        for d, k in enumerate(("vcomxglobal", "vcomyglobal", "vcomzglobal")):
            assert np.abs(o.vcom[d] - G[q + k]).max() <= 1e-13 * max(np.abs(G[q + k]).max(), 1e-300)
        for f in ("md_temp", "dens"):
            assert np.abs(getattr(o, f)[1:-1] - G[q + f]).max() <= 1e-12 * np.abs(G[q + f]).max(), f
        assert int(o.t.last_active) == int(G[q + "scalars"][0]) and int(o.t.vlat_proc) == int(G[q + "scalars"][1])
        if s == 0 or (s - 1) in steps:
            assert np.array_equal(o.fluxl, G[q + "fluxfromleft"].astype(int)) and np.array_equal(o.fluxr, G[q + "fluxfromright"].astype(int))
    assert int(G["t9_update_fd_fluxfromright"].sum() + G["t9_update_fd_fluxfromleft"].sum()) > 0        # not vacuous: step 9 has a crossing, step 8 is in the fixture
    o.close()


@pytest.mark.parametrize("s", [0, 1, 2, 8, 9, 20])
def test_stages_of_calc_ttm_equal_the_reference(s):
    """every stage from the reference's lattice of the stage before it: do_ADV, activation + FILLMESH, CFL_maxdt and the
    sub-step count, the first do_DIFF (virtual lattice included), FILLMESH after it, and the whole loop + the EOS update of U"""
    G = gu.load("ttm1d_slab")
    o = oracle_from(G)
    load_stage(o, G, s, "update_fd")
    o.adv(1.0, s)
    check_stage(o, G, s, "adv", fields=("natoms", "temp", "U", "temp2", "U2"), tol=2e-8)       # inverse EOS: the minimiser's tolerance
    load_stage(o, G, s, "adv")
    o.cell_activation(); o.fillmesh(s); o.fill_ghosts()
    check_stage(o, G, s, "fillmesh0")
    load_stage(o, G, s, "fillmesh0")
    o.cfl(s)
    nsub = max(int(G["ttm_fd_n_timesteps"]), int(o.timestep / o.t.max_dt_ttm))
    assert nsub == int(G["t%d_cfl_scalars" % s][2])
    assert abs(o.t.max_dt_ttm - G["t%d_cfl_scalars" % s][3]) <= 1e-12 * G["t%d_cfl_scalars" % s][3]
    load_stage(o, G, s, "cfl")
    tau = o.timestep / nsub
    o.diff(tau)
    check_stage(o, G, s, "diff_only1", fields=("temp", "temp2", "U", "U2", "xi"))
    for f, k in (("vtemp", "vlat_temp"), ("vmd", "vlat_md_temp")):
        ref = G["t%d_diff_only1_%s" % (s, k)]
        assert np.abs(getattr(o, f) - ref).max() <= 1e-13 * np.abs(ref).max(), f
    o.fillmesh(s); o.fill_ghosts()
    check_stage(o, G, s, "substep1")
    for i in range(1, nsub):
        o.diff(tau); o.fillmesh(s); o.fill_ghosts()
    # the EOS update of U after the loop is part of orc_ttm1d_calc: do it by hand through a full calc below; here the loop's fields
    check_stage(o, G, s, "substeps_done", fields=("temp", "temp2", "xi", "fd_k", "fd_g", "Ce", "Z", "ne"), tol=1e-10)
    ref = G["t%d_substeps_done_vlat_temp" % s]
    assert np.abs(o.vtemp - ref).max() <= 1e-11 * np.abs(ref).max()
    o.close()


def test_whole_calc_ttm_and_move_atoms_from_the_atoms_of_step_0():
    """orc_ttm1d_calc on the reference's atoms of step 0 (fresh lattice): the lattice after the loop incl. the EOS update of U
    and xiarr_global, the sub-step count (111 at step 0: the stability limit of the initial max_dt_ttm), then
    move_atoms_ttm's 1D branch reproduces the reference's atoms of step 1"""
    G = gu.load("ttm1d_slab")
    o = oracle_from(G)
    ort, sorte, imp, mas, nn = atoms(G, 0)
    nsub = o.calc(0, ort, sorte, imp, mas, nn)
    assert nsub == int(G["t0_substeps_done_scalars"][2]) == 111
    check_stage(o, G, 0, "substeps_done", fields=("natoms", "temp", "temp2", "U", "U2", "xi", "md_temp", "dens", "fd_k", "fd_g", "Ce"), tol=1e-10)
    ref = G["t0_substeps_done_xiarr_global"]
    assert np.abs(o.xiarr - ref).max() <= 1e-10 * np.abs(ref).max()
    pos, mom, _ = o.move_atoms(ort, sorte, imp, G["s0_kraft"], mas)
    a, b = np.argsort(G["s0_nummer"]), np.argsort(G["s1_nummer"])
    box = np.diag(G["box"])
    dx = pos[a] - G["s1_ort"][b]; dx[:, 1:] -= np.round(dx[:, 1:] / box[1:]) * box[1:]
    assert np.abs(dx).max() < 1e-12
    assert np.abs(mom[a] - G["s1_impuls"][b]).max() < 1e-12 * np.abs(G["s1_impuls"]).max()
    o.close()
