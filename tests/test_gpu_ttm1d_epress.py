"""-DELECPRESS of the reference's 1D model on the device (run with -m gpu): do_electronic_pressure at the end of calc_ttm and
the pressure force in move_atoms_ttm (imdgpu_ttm1d_set_epress) against tests/golden/ttm1d_ep_slab.npz, a two-rank run of the
reference built with the flag (laser on, planar pressure table)."""
import numpy as np
import pytest

from tests import golden_util as gu
from tests.test_gpu_ttm1d import _check, _gpu, _load
from tests.test_gpu_ttm1d_tmm import _tmm
from tests.test_oracle_ttm1d_epress_vs_reference import ref_deriv

pytestmark = pytest.mark.gpu
S_UEOS = 128


def _ep(g, G):
    _tmm(g, G)
    g.ttm1d_set_epress(dict(x=G["eos_r"], y=G["eos_t"], z=G["epress_z"]))


@pytest.mark.parametrize("s", [1, 2, 8])
def test_pressure_and_gradient(s):
    G = gu.load("ttm1d_ep_slab")
    g = _gpu(G, s)
    _ep(g, G)
    _load(g, G, s, "substeps_done")
    g.ttm1d_stage(s, S_UEOS)                         # the end of calc_ttm: do_electronic_pressure, then U from the EOS
    P, dP = g.ttm1d_epress()
    refp, ref = G["e%d_epress_global" % s], ref_deriv(G, s)
    assert np.array_equal(P == 0, refp == 0) and np.abs(P - refp).max() <= 1e-12 * refp.max()
    fin = np.isfinite(ref)
    assert np.array_equal(np.isinf(dP), np.isinf(ref))
    assert np.abs(dP[fin] - ref[fin]).max() <= 1e-10 * max(np.abs(ref[fin]).max(), 1e-300) + 1e-15
    g.close()


def test_md_steps_with_the_pressure_force():
    """imdgpu_md_step from the reference's atoms of step 0: the atoms of step 4 and step 8 with the force (away from the rank
    boundary of the reference run, where ownership by position and by the last re-binning may differ), and visibly not without"""
    G = gu.load("ttm1d_ep_slab")
    box = np.diag(G["box"])
    res = {}
    for with_force in (True, False):
        g = _gpu(G, 0)
        _ep(g, G) if with_force else _tmm(g, G)
        for st in range(8):
            g.md_step(st)
            if st == 3:
                res[(with_force, 4)] = g.download(fields=("ort", "impuls", "nummer"))
        res[(with_force, 8)] = g.download(fields=("ort", "impuls", "nummer"))
        if with_force:
            g.calc_forces(8); g.ttm1d_calc(8)
            _check(g.ttm1d_download(), G, 8, "substeps_done", ("natoms", "temp", "U", "md_temp", "dens"), 1e-6)
        g.close()
    for s, tol in ((4, 1e-9), (8, 1e-7)):
        d = res[(True, s)]
        a, b = np.argsort(d["nummer"]), np.argsort(G["s%d_nummer" % s])
        far = np.abs(G["s%d_ort" % s][b][:, 0] - box[0] / 2) > 1.0
        dp = np.abs(d["impuls"][a] - G["s%d_impuls" % s][b]).max(axis=1)
        assert dp[far].max() < tol * np.abs(G["s%d_impuls" % s]).max(), (s, dp[far].max())
        d0 = res[(False, s)]
        a0 = np.argsort(d0["nummer"])
        assert np.abs(d0["impuls"][a0] - G["s%d_impuls" % s][b]).max() > 1e-3
