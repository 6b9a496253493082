"""GPU parity for fitDL, the per-cell Drude-Lorentz parameters of the Maxwell solver from tricubic tables (run with
-m gpu): the device's look-up (tricub_lookup in imdgpu_ttm.hip) against the answers of the reference's OWN
read_tricub_interp / do_tricubinterp (tests/golden/tricub.npz, oracle/ref_probe_tricub.c) -- bit for bit, the
out-of-table -1 included -- and its fitDL against the oracle's (imd_ttm_3D.c:2919-2968).  The look-up inside the
fused solver loop is covered by test_gpu_fdtd.py::test_solver_loop_with_diffusion_vs_oracle[tables]."""
import os

import numpy as np
import pytest

from imd_amd import tricub
from oracle import pyoracle as po
from tests.test_gpu_fdtd import _row_system

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "tricub.npz"))
NAMES = ("centred", "end_to_end", "uneven", "narrow")


def _table(name, scale=1.0):
    return tricub.TricubTable(G[name + "_range"], G[name + "_x"], G[name + "_y"], G[name + "_z"], G[name + "_f"] * scale)


def _system():
    s, _ = _row_system(40, 7.0)
    s.init_fdtd(1e15, 800e-9, 1e-15, 0.5e-15, 15 * 7e-10, bw=8)
    return s


def test_lookup_equals_the_reference_bit_for_bit():
    s = _system()
    s.fdtd_set_dl_tables([_table(n) for n in NAMES] + [_table("centred", 2.0)])
    for w, name in enumerate(NAMES):
        got = s.fdtd_dl_lookup(w, G[name + "_queries"])
        ref = G[name + "_ref"]
        assert (got == -1).sum() == 6
        assert np.array_equal(got, ref), (name, np.abs(got - ref).max())
    s.close()


@pytest.mark.parametrize("grids", ["one_grid", "mixed"])
def test_fitdl_equals_the_oracle(grids):
    """one_grid: five quantities on the same nodes (one cell search on the device); mixed: tables of different nodes"""
    s = _system()
    tabs = [_table("centred" if grids == "one_grid" or q % 2 == 0 else ("end_to_end", "uneven")[q // 2], sc)
            for q, sc in enumerate((1.0, 0.7, 1.3, 2.1, 0.2))]
    s.fdtd_set_dl_tables(tabs)
    otabs = [po.OracleTricub(t.range, t.x, t.y, t.z, t.f) for t in tabs]
    rng = np.random.default_rng(8)
    n = 4000
    temp = 10 ** rng.uniform(-2.5, 2.0, n)            # below the 0.0259 eV floor up to beyond the table (clamped to ymax)
    ti = 10 ** rng.uniform(-2.5, 1.0, n)
    dens = rng.uniform(1700.0, 3100.0, n)             # some outside the table: -1 (and -1 / hbar) like the reference
    ne = 10 ** rng.uniform(27, 29.5, n)
    DL, st = po.oracle_fitdl(otabs, temp, ti, dens, ne)
    got = s.fdtd_fitdl(temp, ti, dens, ne)
    assert (st == -1).sum() > 50 and np.array_equal(got[:, 0] == -1, st == -1)
    # log10 on the device may differ from glibc's in the last bit, which moves a look-up only if it falls on a cell border
    assert (got[:, :5] == DL[:, :5]).all(axis=1).mean() > 0.998
    assert np.abs(got[:, 5] / DL[:, 5] - 1).max() < 1e-15
    for o in otabs:
        o.close()
    s.close()


def test_tables_need_the_solver_and_switch_off_again():
    from imd_amd.lib import ImdGpuError
    s, _ = _row_system(20, 7.0)
    with pytest.raises(ImdGpuError):
        s.fdtd_set_dl_tables([_table("centred")] * 5)
    s.init_fdtd(1e15, 800e-9, 1e-15, 0.5e-15, 15 * 7e-10, bw=5)
    with pytest.raises(ImdGpuError):
        s.fdtd_dl_lookup(0, np.zeros((1, 3)))
    s.fdtd_set_dl_tables([_table("centred")] * 5)
    assert s.fdtd_dl_lookup(4, G["centred_queries"][:5]).shape == (5,)
    s.fdtd_set_dl_tables(None)
    with pytest.raises(ImdGpuError):
        s.fdtd_dl_lookup(0, np.zeros((1, 3)))
    s.close()
