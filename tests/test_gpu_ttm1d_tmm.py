"""The transfer-matrix laser source of the reference's 1D two-temperature model (imd_tmm.c) on the device (run with -m gpu):
imdgpu_ttm1d_tmm_* against tests/golden/ttm1d_tmm_slab.npz -- the reference's OWN tmm_init / do_tmm run on two ranks with
synthetic input tables (oracle/_ref/imd_mpi_eam_ttm1d_tmm_nbl, oracle/ref_probe_ttm1d.c).  Function level (one do_tmm from
the reference's lattice in front of it) and as a whole (MD steps through imdgpu_md_step with the laser coming on at step 2).
Tolerances: the device's exp / sin / cos / log / pow differ from glibc's in the last bits; the recursions over the layers
run in the reference's order, so nothing is re-associated."""
import numpy as np
import pytest

from tests import golden_util as gu
from tests.test_gpu_ttm1d import _check, _gpu, _load

pytestmark = pytest.mark.gpu
S_TMM = 256


def _tmm(g, G):
    g.init_ttm1d_tmm(float(G["tmm_I0"]), float(G["tmm_lam"]), float(G["tmm_sigma_t"]), float(G["tmm_t0"]), G["tmm_eps_bb"], G["tmm_K12"],
                     dict(x=G["eos_r"], y=G["eos_t"], z=G["tmm_phase_z"]), tmm_threshold=float(G["tmm_abs_threshold"]))


def _rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("s", [2, 3, 8, 12])
def test_one_do_tmm_equals_the_reference(s):
    G = gu.load("ttm1d_tmm_slab")
    g = _gpu(G, s)
    _tmm(g, G)
    _load(g, G, s, "cfl")
    g.ttm1d_tmm_set_time(float(G["m%d_tmm_time" % s]))
    g.ttm1d_stage(s, S_TMM)
    st = g.ttm1d_tmm_state()
    q = "m%d_sub0_" % s
    sc = G[q + "scalars"]
    act = G[q + "active"].astype(int)
    assert st["too_short"] == 0 and st["laser_active"] == 1 and np.array_equal(st["active"], act)
    assert abs(st["I_t"] - sc[0]) <= 1e-13 * sc[0]
    on = act == 1
    assert _rel(st["eps_re"][on], G[q + "eps_re"][on]) < 1e-11 and _rel(st["eps_im"][on], G[q + "eps_im"][on]) < 1e-11
    assert abs(st["refl"] - sc[2]) < 1e-10 and abs(st["trans"] - sc[3]) < 1e-10 * sc[3] + 1e-14 and abs(st["absorbed"] - sc[4]) < 1e-10
    ecut = st["ecut"]
    ref_q = G[q + "qabs"]
    assert 13 < ecut < 22 and (ref_q[ecut:] == 0).all() if s == 2 else ecut > 13
    assert _rel(st["qabs"][:ecut], ref_q[:ecut]) < 1e-9
    src = g.ttm1d_download()["source"][1:-1]
    assert _rel(src[:ecut], G[q + "source"][:ecut]) < 1e-9
    # cells whose phase says molten lost the interband term: visible in the fixture, reproduced here
    if s == 2:
        Ti = G["t2_cfl_md_temp"] * 11604.52500617
        assert st["eps_im"][on & (Ti > 1160.0)].max() < 0.8 * st["eps_im"][on & (Ti < 1000.0)].min()
    g.close()


def test_md_steps_with_the_laser_coming_on():
    """imdgpu_md_step with the 1D lattice and its laser source from the reference's atoms of step 0: the trajectory reaches the
    reference's atoms of steps 3 and 8; the lattice after step 3's calc_ttm, the absorbed energy and the clock of the source"""
    G = gu.load("ttm1d_tmm_slab")
    g = _gpu(G, 0)
    _tmm(g, G)
    box = np.diag(G["box"])

    def atoms_equal(s, tol):
        d = g.download(fields=("ort", "impuls", "nummer"))
        a, b = np.argsort(d["nummer"]), np.argsort(G["s%d_nummer" % s])
        dx = d["ort"][a] - G["s%d_ort" % s][b]; dx[:, 1:] -= np.round(dx[:, 1:] / box[1:]) * box[1:]
        assert np.abs(dx).max() < tol, (s, np.abs(dx).max())
        assert np.abs(d["impuls"][a] - G["s%d_impuls" % s][b]).max() < tol * np.abs(G["s%d_impuls" % s]).max()

    for st in range(3):
        g.md_step(st)
    atoms_equal(3, 1e-9)
    st3 = g.ttm1d_tmm_state()
    assert abs(st3["tmm_time"] - float(G["m3_tmm_time"])) <= 1e-12 * float(G["m3_tmm_time"])
    # Eabs_local right after the last do_tmm of step 2 + that sub-step's share = what the device holds after step 2
    tau = float(G["timestep"]) / int(G["m2_nsub"])
    last = float(G["m2_sub%d_eabs" % (int(G["m2_nsub"]) - 1)])
    vol = float(G["ttm_fd_vol"])
    last += (G["t2_substeps_done_source"] * vol * tau)[: int(G["t2_substeps_done_scalars"][0]) + 1].sum()
    assert abs(st3["eabs"] - last) <= 1e-8 * last and last > 1.0
    for st in range(3, 8):
        g.md_step(st)
    atoms_equal(8, 1e-8)
    g.calc_forces(8); g.ttm1d_calc(8)
    lat = g.ttm1d_download()
    _check(lat, G, 8, "substeps_done", ("natoms", "temp", "U", "xi", "md_temp", "dens", "source"), 1e-7)
    st8 = g.ttm1d_tmm_state()
    q = "m8_sub%d_" % (int(G["m8_nsub"]) - 1)
    assert st8["laser_active"] == 1 and abs(st8["refl"] - G[q + "scalars"][2]) < 1e-8
    g.close()


def test_without_the_tables_nothing_changes_and_stage_needs_init():
    from imd_amd.lib import ImdGpuError
    G = gu.load("ttm1d_tmm_slab")
    g = _gpu(G, 2)
    with pytest.raises(ImdGpuError):
        g.ttm1d_stage(2, S_TMM)
    with pytest.raises(ImdGpuError):
        g.ttm1d_tmm_state()
    g.close()


def test_a_sample_too_short_for_the_threshold_is_an_error_like_in_the_reference():
    """imd_tmm.c:355-363: when the transmitted intensity never falls below exp(-tmm_threshold) inside the sample the reference
    aborts ("Sample is too short for TMM"); the MD step fails with the same words"""
    from imd_amd.lib import ImdGpuError
    G = gu.load("ttm1d_tmm_slab")
    g = _gpu(G, 0)
    g.init_ttm1d_tmm(float(G["tmm_I0"]), float(G["tmm_lam"]), float(G["tmm_sigma_t"]), float(G["tmm_t0"]), G["tmm_eps_bb"], G["tmm_K12"],
                     dict(x=G["eos_r"], y=G["eos_t"], z=G["tmm_phase_z"]), tmm_threshold=20.0)
    g.md_step(0); g.md_step(1)                      # nothing happens before step 2
    with pytest.raises(ImdGpuError, match="too short for TMM"):
        g.md_step(2)
    g.close()
