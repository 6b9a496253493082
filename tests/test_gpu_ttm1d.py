"""The reference's own 1D TTM (imd_ttm.c, -DTTM1D) on the device (run with -m gpu): imdgpu_ttm1d_* against
tests/golden/ttm1d_slab.npz -- the reference's functions driven stage by stage on two ranks (oracle/ref_probe_ttm1d.c) -- and
against the CPU oracle of the same variant.  Function level (every stage from the reference's lattice of the stage before)
and as a whole (20 MD steps through imdgpu_md_step: forces, calc_ttm of the 1D file beside them, its move_atoms_ttm)."""
import numpy as np
import pytest

from imd_amd.system import GpuSystem
from tests import golden_util as gu

pytestmark = pytest.mark.gpu
S_UFD, S_ADV, S_ACT, S_CFL, S_DIFF1, S_FILL, S_LOOP, S_UEOS = 1, 2, 4, 8, 16, 32, 64, 128


def _gpu(G, s, with_lattice=True):
    pre = "s%d_" % s
    g = GpuSystem(G["box"], G["pbc_dirs"], G["cell_dim"], int(G["ntypes"]), float(G["cellsz"]), float(G["nbl_margin"]), float(G["timestep"]),
                  count_ttm_neighbours=True)
    g.upload_tables(gu.table(G, "pair_pot"), gu.table(G, "rho_h_tab"), gu.table(G, "embed_pot"))
    g.upload_atoms(G[pre + "ort"], G[pre + "impuls"], G[pre + "masse"], None, G[pre + "nummer"])
    if with_lattice:
        g.init_ttm1d(int(G["ttm_global_fd_dim"]), int(G["ttm_fd_min_atoms"]), int(G["ttm_fd_n_timesteps"]), int(G["ttm_vlatdim"]),
                     int(G["ttm_vlatbuffer"]), float(G["ttm_vlatdens"]), float(G["ttm_atomic_weight"]), float(G["ttm_atomic_charge"]),
                     dict(x=G["eos_r"], y=G["eos_t"], z=G["eos_cve"]), dict(x=G["eos_r"], y=G["eos_ts"], z=G["eos_ee"]),
                     ref_ranks=int(G["ref_ranks"]), adv=True)
    return g


def _pad(a):
    return np.concatenate([[0], a, [0]])


def _load(g, G, s, tag):
    q = "t%d_%s_" % (s, tag)
    f = {k: _pad(G[q + k]) for k in ("natoms", "natoms_old", "temp", "temp2", "U", "U2", "xi", "md_temp", "source", "dens", "fd_k", "fd_g", "Z", "Ce", "ne")}
    f.update(natomsglobal=G[q + "natomsglobal"], fluxfromleft=G[q + "fluxfromleft"], fluxfromright=G[q + "fluxfromright"],
             vcomglobal=np.stack([G[q + "vcomxglobal"], G[q + "vcomyglobal"], G[q + "vcomzglobal"]]), xiarr_global=G[q + "xiarr_global"],
             vlat_temp=G[q + "vlat_temp"], vlat_md_temp=G[q + "vlat_md_temp"], vlat_fd_k=G[q + "vlat_fd_k"], vlat_fd_g=G[q + "vlat_fd_g"],
             vlat_Ce=G[q + "vlat_Ce"], scalars=G[q + "scalars"])
    g.ttm1d_upload(**f)


def _check(d, G, s, tag, fields, tol):
    q = "t%d_%s_" % (s, tag)
    for f in fields:
        ref, got = G[q + f], d[f][1:-1]
        if f in ("natoms", "natoms_old"):
            assert np.array_equal(got, ref.astype(int)), (s, tag, f)
        else:
            assert np.abs(got - ref).max() <= tol * max(np.abs(ref).max(), 1e-300), (s, tag, f, np.abs(got - ref).max())


@pytest.mark.parametrize("s", [0, 1, 2, 8, 9, 20])
def test_stages_of_the_1d_calc_ttm_equal_the_reference(s):
    G = gu.load("ttm1d_slab")
    g = _gpu(G, s)
    _load(g, G, s, "update_fd"); g.ttm1d_stage(s, S_ADV)
    _check(g.ttm1d_download(), G, s, "adv", ("natoms", "temp", "U", "temp2", "U2"), 2e-8)
    _load(g, G, s, "adv"); g.ttm1d_stage(s, S_ACT)
    _check(g.ttm1d_download(), G, s, "fillmesh0", ("natoms", "temp", "temp2", "U", "U2", "xi", "fd_k", "fd_g", "Ce", "Z", "ne"), 1e-10)
    _load(g, G, s, "fillmesh0"); g.ttm1d_stage(s, S_CFL)
    d = g.ttm1d_download()
    assert int(d["scalars"][2]) == int(G["t%d_cfl_scalars" % s][2])
    assert abs(d["scalars"][3] - G["t%d_cfl_scalars" % s][3]) <= 1e-10 * G["t%d_cfl_scalars" % s][3]
    _load(g, G, s, "cfl"); g.ttm1d_stage(s, S_DIFF1)
    d = g.ttm1d_download()
    _check(d, G, s, "diff_only1", ("temp", "temp2", "U", "U2", "xi"), 1e-11)
    for f in ("vlat_temp", "vlat_md_temp"):
        ref = G["t%d_diff_only1_%s" % (s, f)]
        assert np.abs(d[f] - ref).max() <= 1e-12 * np.abs(ref).max(), f
    _load(g, G, s, "diff_only1"); g.ttm1d_stage(s, S_FILL)
    _check(g.ttm1d_download(), G, s, "substep1", ("temp", "temp2", "fd_k", "fd_g", "Ce", "Z", "ne"), 1e-10)
    _load(g, G, s, "cfl"); g.ttm1d_stage(s, S_LOOP | S_UEOS)
    d = g.ttm1d_download()
    _check(d, G, s, "substeps_done", ("temp", "temp2", "U", "U2", "xi", "fd_k", "fd_g", "Ce"), 1e-9)
    ref = G["t%d_substeps_done_vlat_temp" % s]
    assert np.abs(d["vlat_temp"] - ref).max() <= 1e-10 * np.abs(ref).max()
    ref = G["t%d_substeps_done_xiarr_global" % s]
    assert np.abs(d["xiarr_global"] - ref).max() <= 1e-9 * np.abs(ref).max()
    g.close()


@pytest.mark.parametrize("s", [0, 1, 9])
def test_update_fd_slab_sums_from_the_atoms(s):
    """the atoms of step s on the device (forces first: the neighbour counts), slab sums by position: natoms, v_com, md_temp,
    density, the cells behind the last active one switched off; at step 9 (atoms of step 8 seen first) the flux counters"""
    G = gu.load("ttm1d_slab")
    g = _gpu(G, s)
    if s == 9:                                   # the slab every atom was in at step 8: one update_fd on those positions
        g8 = None
        g.upload_atoms(G["s8_ort"], G["s8_impuls"], G["s8_masse"], None, G["s8_nummer"])
        g.calc_forces(8); g.ttm1d_stage(8, S_UFD)
        # the same atoms (by number) at their positions of step 9: the list of step 8 still serves (no re-binning in between)
        d8 = g.download(fields=("nummer",))["nummer"]
        o9 = np.argsort(G["s9_nummer"])
        idx = o9[np.searchsorted(G["s9_nummer"][o9], d8)]
        g.update_positions(G["s9_ort"][idx]); g.update_momenta(G["s9_impuls"][idx])
    if s > 0:
        prev = {1: 0, 9: 8}[s]
        _load(g, G, prev, "substeps_done")       # natoms_old, last_active_cell come from the step before
    g.calc_forces(s)
    g.ttm1d_stage(s, S_UFD)
    d = g.ttm1d_download()
    q = "t%d_update_fd_" % s
    assert np.array_equal(d["natomsglobal"], G[q + "natomsglobal"].astype(int))
    assert np.array_equal(d["natoms"][1:-1], G[q + "natoms"].astype(int))
    for k, f in enumerate(("vcomxglobal", "vcomyglobal", "vcomzglobal")):
        assert np.abs(d["vcomglobal"][k] - G[q + f]).max() <= 1e-12 * max(np.abs(G[q + f]).max(), 1e-300)
    for f in ("md_temp", "dens"):
        assert np.abs(d[f][1:-1] - G[q + f]).max() <= 1e-12 * np.abs(G[q + f]).max(), f
    assert int(d["scalars"][0]) == int(G[q + "scalars"][0]) and int(d["scalars"][1]) == int(G[q + "scalars"][1])
    if s == 9:
        assert G[q + "fluxfromright"].sum() + G[q + "fluxfromleft"].sum() > 0
        assert np.array_equal(d["fluxfromleft"], G[q + "fluxfromleft"].astype(int)) and np.array_equal(d["fluxfromright"], G[q + "fluxfromright"].astype(int))
    g.close()


def test_twenty_md_steps_with_the_1d_ttm_follow_the_reference():
    """imdgpu_md_step with a 1D lattice attached: forces, the 1D calc_ttm forked beside them, move_atoms_ttm's TTM1D branch.
    20 steps from the reference's atoms of step 0 (two list rebuilds, atoms changing slabs, do_ADV active): positions and
    momenta of step 20 and the lattice after step 20's calc_ttm"""
    G = gu.load("ttm1d_slab")
    g = _gpu(G, 0)
    for st in range(20):
        g.md_step(st)
    d = g.download(fields=("ort", "impuls", "nummer"))
    a, b = np.argsort(d["nummer"]), np.argsort(G["s20_nummer"])
    box = np.diag(G["box"])
    dx = d["ort"][a] - G["s20_ort"][b]; dx[:, 1:] -= np.round(dx[:, 1:] / box[1:]) * box[1:]
    assert np.abs(dx).max() < 1e-9
    assert np.abs(d["impuls"][a] - G["s20_impuls"][b]).max() < 1e-9 * np.abs(G["s20_impuls"]).max()
    g.calc_forces(20); g.ttm1d_calc(20)
    lat = g.ttm1d_download()
    _check(lat, G, 20, "substeps_done", ("natoms", "temp", "U", "md_temp", "dens"), 1e-8)
    # xi accumulates (Te - Tl): a relative 1e-11 on md_temp (the device sums the slabs in another order than the reference, and
    # twenty MD steps amplify the last bit) is 1e-8 on xi
    _check(lat, G, 20, "substeps_done", ("xi",), 1e-6)
    assert int(lat["scalars"][2]) == int(G["t20_substeps_done_scalars"][2])
    g.close()
