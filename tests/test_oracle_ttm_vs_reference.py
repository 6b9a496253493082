"""Pin the TTM oracle (oracle/ttm_oracle.c) against the reference's own compiled
imd_ttm_3D.c functions (fixtures tests/golden/ttm_*.npz, probe
oracle/ref_probe_ttm.c).  Everything except the EOS-interpolated quantities
(different linear-interpolation rule on a planar table) and what depends on
them is bit-exact; the rest agrees to 1e-12 relative."""
import numpy as np
import pytest

from tests import golden_util as gu

# update_fd outputs: bit-exact.  Everything downstream of the EOS look-up (Ce -> Te -> k, g, Z, xi ...):
# 1e-12 relative (the restatement interpolates bilinearly, the reference on nn-c's triangulation; on the
# planar fixture tables the two agree to rounding, observed <= 4e-16).
EXACT = ("natoms", "natoms_old", "md_temp", "source", "vcom", "dens")


def _cmp(t, st, tag):
    for f, ref in st.items():
        if f == "max_dt_ttm":
            continue
        if f == "Z" and tag in ("diff_only1", "entry", "update_fd"):
            continue      # Z is FILLMESH scratch kept only in l1: stale in the reference after the l1<->l2 swap
        a = getattr(t, f)
        # only these three are defined in the ghost shell, and only after ttm_fill_ghost_layers
        # (right after do_DIFF's l1<->l2 swap the reference's ghost shell holds the other buffer's stale copy)
        if f not in ("natoms", "temp", "fd_k") or tag == "diff_only1":
            a, ref = a[1:-1, 1:-1, 1:-1], ref[1:-1, 1:-1, 1:-1]
        if f in EXACT:
            assert np.array_equal(a, ref), "%s: %s differs (max %g)" % (tag, f, np.abs(a - ref).max())
        else:
            assert np.abs(a - ref).max() <= 1e-12 * max(np.abs(ref).max(), 1e-300), "%s: %s" % (tag, f)


@pytest.mark.parametrize("case", ["ttm_2048", "ttm_2048_dg", "ttm_6912_ext2", "ttm_slab", "ttm_2types"])
def test_ttm_trajectory_against_reference(case):
    G = gu.load(case)
    s = gu.oracle_from_golden(G, 0)
    s.s.count_ttm_neigh = 1
    t = gu.oracle_ttm_from_golden(G, s)
    n = int(G["ttm_diff_substeps"])
    tau = float(G["ttm_tau"])
    steps = [int(x) for x in G["steps"]]
    for st in range(steps[-1] + 1):
        s.calc_forces(st)
        pre = "s%d_" % st
        assert np.array_equal(s.gather_inner("numneighs"), G[pre + "numneighs"])      # TTM neighbour count
        assert np.abs(s.gather_inner("kraft") - G[pre + "kraft"]).max() < 1e-12
        assert np.abs(s.gather_inner("impuls") - G[pre + "impuls"]).max() < 1e-12
        # stage by stage, like the probe
        _cmp(t, gu.ttm_stage(G, st, "entry"), "entry")
        t.update_fd(st);                                  _cmp(t, gu.ttm_stage(G, st, "update_fd"), "update_fd")
        t.cell_activation(); t.fillmesh(st); t.fill_ghosts(); _cmp(t, gu.ttm_stage(G, st, "fillmesh0"), "fillmesh0")
        for i in range(n):
            t.diff(tau, n)
            if i == 0:
                _cmp(t, gu.ttm_stage(G, st, "diff_only1"), "diff_only1")
            t.fillmesh(st); t.fill_ghosts()
            if i == 0:
                _cmp(t, gu.ttm_stage(G, st, "substep1"), "substep1")
            if i == 1:
                _cmp(t, gu.ttm_stage(G, st, "substep2"), "substep2")
        _cmp(t, gu.ttm_stage(G, st, "substeps_done"), "substeps_done")
        t.cfl(st, float(G["timestep"]), int(G["ttm_fd_n_timesteps"]))
        assert abs(t.t.max_dt_ttm - float(G["t%d_cfl_max_dt_ttm" % st])) <= 1e-12 * t.t.max_dt_ttm
        t.move_atoms_ttm()
        s.check_nblist()
    if case == "ttm_slab":
        nat = G["t0_update_fd_natoms"][1:-1, 1:-1, 1:-1]
        m = int(G["ttm_fd_min_atoms"])
        assert (nat < m).any() and (nat >= m).any()          # vacuum / partially filled cells are covered


def test_calc_ttm_sequence_equals_stages():
    G = gu.load("ttm_2048")
    s = gu.oracle_from_golden(G, 0); s.s.count_ttm_neigh = 1
    t = gu.oracle_ttm_from_golden(G, s)
    s.calc_forces(0)
    t.calc_ttm(0, int(G["ttm_diff_substeps"]))
    _cmp(t, gu.ttm_stage(G, 0, "cfl"), "cfl")


def test_inverse_eos_vs_reference():
    """EOS_te_from_r_ee (Brent minimiser, imd_ttm_3D.c:2786-2811 + fminbnd3.c) and EOS_ee_from_r_te: the oracle's
    restatement against values computed by the reference's own functions (tests/golden/eos_inverse.npz, 126 points
    on and off the ee(T) curve).  The iteration is deterministic, but the table look-up rules differ in the last
    ulp (triangle-linear vs bilinear), which can flip one late comparison: 90 of the 126 points agree to 1e-9 K, the
    rest to 1.6e-6 K.  The bar is the minimiser's own x tolerance, 1e-5 K (fminbnd3.c:8)."""
    G = gu.load("eos_inverse")
    G0 = gu.load("ttm_2048")
    s = gu.oracle_from_golden(G0, 0)
    t = gu.oracle_ttm_from_golden(G0, s)
    nr, nt = int(G["eos_nr"]), int(G["eos_nt"])
    t.set_eos(1, nr, nt, G["eos_r"][0], G["eos_r"][-1], G["eos_ts"][0], G["eos_ts"][-1], G["eos_ee"])
    on_curve = np.arange(len(G["r"])) % 3 == 0
    ee = np.array([t.eos_ee(r, tk) for r, tk in zip(G["r"][on_curve], G["t_in"][on_curve])])
    assert np.abs(ee / G["e"][on_curve] - 1).max() < 1e-13
    te = np.array([t.eos_te_from_r_ee(r, e) for r, e in zip(G["r"], G["e"])])
    assert np.abs(te - G["t_out"]).max() < 1e-5
    assert (np.abs(te - G["t_out"]) < 1e-9).sum() >= 80
    t.close()


def test_flux_counters_of_update_fd_over_a_trajectory():
    """update_fd's atom-flux counters (imd_ttm_3D.c:335-370; the input of do_ADV): MD + TTM from the reference's
    step-0 arrangement for 43 steps; at the three list rebuilds where atoms changed FD cells (steps 17, 30, 43) the
    eight counters of every cell must equal the reference's exactly, and the arrangement must still be the reference's."""
    G = gu.load("ttm_flux")
    s = gu.oracle_from_golden(G, 0)
    s.s.count_ttm_neigh = 1
    t = gu.oracle_ttm_from_golden(G, s)
    nsub = int(G["ttm_diff_substeps"])
    inner = (slice(1, -1),) * 3
    for st in range(44):
        s.calc_forces(st)
        if st in (0, 1, 17, 30, 43):
            t.update_fd(st)
            assert np.array_equal(t.flux[inner], G["t%d_flux" % st][inner]), st
            assert np.array_equal(t.natoms[inner], G["t%d_update_fd_natoms" % st][inner])
            # run the rest of calc_ttm from here (update_fd again is idempotent for everything but the counters)
            t.cell_activation(); t.fillmesh(st); t.fill_ghosts()
            for _ in range(nsub):
                t.diff(float(G["ttm_tau"]), nsub); t.fillmesh(st); t.fill_ghosts()
        else:
            t.calc_ttm(st, nsub)
        if st in (17, 30, 43):
            assert np.array_equal(s.gather_inner("nummer"), G["s%d_nummer" % st])
        t.move_atoms_ttm()
        s.check_nblist()
    assert int(G["t17_flux"].sum()) > 0 and int(G["t30_flux"].sum()) > 0
    t.close()


def _load_lattice(t, G, pre, flux):
    t.t.cur = 0
    for f in ("natoms", "natoms_old", "xi", "md_temp", "source", "dens", "fd_k", "fd_g", "Z", "Ce", "ne"):
        getattr(t, f)[...] = G[pre + f]
    t.vcom[...] = G[pre + "vcom"]
    t._temp[0][...] = G[pre + "temp"]; t._temp[1][...] = G[pre + "temp2"]
    t._U[0][...] = G[pre + "U"]; t._U[1][...] = G[pre + "U2"]
    t.flux[...] = flux


def test_do_adv_single_stage_against_the_two_rank_reference():
    """do_ADV + do_COMMFLUX (imd_ttm_3D.c:1321-1458, :667-995).  The fixture is the reference run on two ranks along x
    (the step cannot run on one), assembled into the global lattice.  The oracle's lattice is loaded with the
    reference's state after update_fd (both lattices + flux counters), orc_ttm_adv runs once, and the result is
    compared with the reference's state after do_ADV: U (both lattices = the swap) and the untouched temp2 bit-exact,
    the new temp -- it comes out of the Brent minimiser of EOS_te_from_r_ee -- within that minimiser's own x
    tolerance, 1e-5 K (see test_inverse_eos...).  Steps 41 and 48 are list rebuilds where atoms changed FD cells."""
    G = gu.load("ttm_adv")
    s = gu.oracle_from_golden(G, 0)
    t = gu.oracle_ttm_from_golden(G, s)
    inner = (slice(1, -1),) * 3
    assert {41, 48} <= set(int(v) for v in G["flux_steps"])
    for st in (int(v) for v in G["steps"]):
        _load_lattice(t, G, "t%d_update_fd_" % st, G["t%d_flux" % st])
        t.adv(1.0, st)
        for f in ("U", "U2", "temp2"):
            assert np.array_equal(getattr(t, f)[inner], G["t%d_adv_%s" % (st, f)][inner]), (st, f)
        assert np.abs(t.temp[inner] - G["t%d_adv_temp" % st][inner]).max() * 11604.5 < 1e-5, st
    # the advected energy really moved at the rebuild steps
    assert np.abs(G["t41_adv_U"][inner] - G["t41_update_fd_U"][inner]).max() > 1e-4
    t.close()


def test_diffusion_after_the_adv_swap():
    """From the reference's lattice right after do_ADV: cell activation, FILLMESH, ghost fill and the 100 sub-steps
    must land on the reference's lattice (1e-12): pins which lattice is current after the extra swap."""
    G = gu.load("ttm_adv")
    s = gu.oracle_from_golden(G, 0)
    t = gu.oracle_ttm_from_golden(G, s)
    nsub = int(G["ttm_diff_substeps"])
    inner = (slice(1, -1),) * 3
    for st in (2, 41):
        _load_lattice(t, G, "t%d_adv_" % st, G["t%d_flux" % st])
        t.cell_activation(); t.fillmesh(st); t.fill_ghosts()
        for f in ("temp", "U", "temp2", "U2"):
            assert np.array_equal(getattr(t, f)[inner], G["t%d_fillmesh0_%s" % (st, f)][inner]), (st, f)
        for _ in range(nsub):
            t.diff(float(G["ttm_tau"]), nsub); t.fillmesh(st); t.fill_ghosts()
        for f in ("temp", "U", "temp2", "U2", "xi", "fd_k", "fd_g", "Ce"):
            a, b = getattr(t, f)[inner], G["t%d_substeps_done_%s" % (st, f)][inner]
            assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max(), (st, f)
    t.close()


def test_md_ttm_trajectory_with_advection_against_the_two_rank_reference():
    """MD + TTM with do_ADV every step from the reference's (global) step-0 arrangement for 50 steps: flux counters
    exact at every kept step (41 and 48 are rebuilds with atoms changing FD cells), lattice fields within 2e-8
    relative (each step's new temperatures carry the inverse-EOS minimiser's 1e-5 K tolerance, ~4e-9 relative)."""
    G = gu.load("ttm_adv")
    s = gu.oracle_from_golden(G, 0)
    s.s.count_ttm_neigh = 1
    t = gu.oracle_ttm_from_golden(G, s)
    t.t.adv = 1
    nsub = int(G["ttm_diff_substeps"])
    inner = (slice(1, -1),) * 3
    for st in range(50):
        s.calc_forces(st)
        t.calc_ttm(st, nsub)
        if st in G["steps"]:
            assert np.array_equal(t.flux[inner], G["t%d_flux" % st][inner]), st
            assert np.array_equal(t.natoms[inner], G["t%d_substeps_done_natoms" % st][inner])
            for f in ("temp", "U", "temp2", "U2", "xi", "md_temp", "dens", "vcom"):
                a, b = getattr(t, f)[inner], G["t%d_substeps_done_%s" % (st, f)][inner]
                assert np.abs(a - b).max() <= 2e-8 * np.abs(b).max(), (st, f)
        t.move_atoms_ttm()
        s.check_nblist()
    t.close()


def _laser(G):
    return (G["laser_dir"], float(G["laser_offset"]), float(G["laser_mu"]), float(G["laser_p_peak"]), float(G["laser_t_0"]),
            float(G["laser_sigma_t_squared"]))


def test_laser_source_and_diffusion_with_source_against_reference():
    """laser_rescale_ttm (Beer-Lambert source, imd_laser.c:715-788; -DLASER build of the reference) and do_DIFF with a
    non-zero source term (imd_ttm_3D.c:1718): slab fixture, pulse centred on step 3.
    (1) from the reference's lattice after update_fd the source of every cell equals the reference's bit for bit;
    (2) from the lattice with the source in, activation + FILLMESH + 100 sub-steps land on the reference's lattice
        (1e-11; temperatures rise from 0.03 to 3 eV within the pulse)."""
    G = gu.load("ttm_laser")
    s = gu.oracle_from_golden(G, 0)
    t = gu.oracle_ttm_from_golden(G, s)
    nsub = int(G["ttm_diff_substeps"])
    inner = (slice(1, -1),) * 3
    dt = float(G["timestep"])
    for st in (0, 2, 3, 5):
        _load_lattice(t, G, "t%d_update_fd_" % st, 0)
        t.laser_rescale(*_laser(G), time=dt * st)
        assert np.array_equal(t.source[inner], G["t%d_laser_source" % st][inner]), st
        t.cell_activation(); t.fillmesh(st); t.fill_ghosts()
        for q in range(nsub):
            t.diff(float(G["ttm_tau"]), nsub); t.fillmesh(st); t.fill_ghosts()
            if q == 0:
                for f in ("temp", "U", "xi"):
                    a, b = getattr(t, f)[inner], G["t%d_substep1_%s" % (st, f)][inner]
                    assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max(), (st, f)
        for f in ("temp", "U", "xi", "fd_k", "fd_g", "Ce", "Z"):
            a, b = getattr(t, f)[inner], G["t%d_substeps_done_%s" % (st, f)][inner]
            assert np.abs(a - b).max() <= 1e-11 * np.abs(b).max(), (st, f)
    assert G["t3_laser_source"].max() > 5.0 and G["t5_substeps_done_temp"].max() > 3.0
    t.close()


def test_md_ttm_trajectory_with_laser_against_reference():
    """MD + TTM with the Beer-Lambert source for six steps from the reference's step-0 arrangement (calc_ttm sequence
    with laser_rescale_ttm after update_fd): lattice after the sub-steps of steps 2, 3, 5 within 1e-9, and do_DIFF's
    running total of the absorbed energy Eabs_local (imd_ttm_3D.c:1647-1651) within 1e-12."""
    G = gu.load("ttm_laser")
    s = gu.oracle_from_golden(G, 0)
    s.s.count_ttm_neigh = 1
    t = gu.oracle_ttm_from_golden(G, s)
    nsub = int(G["ttm_diff_substeps"])
    inner = (slice(1, -1),) * 3
    dt = float(G["timestep"])
    for st in range(6):
        s.calc_forces(st)
        t.update_fd(st)
        t.laser_rescale(*_laser(G), time=dt * st)
        t.cell_activation(); t.fillmesh(st); t.fill_ghosts()
        for _ in range(nsub):
            t.diff(float(G["ttm_tau"]), nsub); t.fillmesh(st); t.fill_ghosts()
        if st in (0, 2, 3, 5):
            assert t.t.laser_active == int(G["t%d_laser_active" % st])
            assert abs(t.t.eabs - float(G["t%d_eabs" % st])) <= 1e-12 * float(G["t%d_eabs" % st]), st
            for f in ("temp", "U", "xi", "md_temp"):
                a, b = getattr(t, f)[inner], G["t%d_substeps_done_%s" % (st, f)][inner]
                assert np.abs(a - b).max() <= 1e-9 * np.abs(b).max(), (st, f)
        t.move_atoms_ttm()
        s.check_nblist()
    t.close()
