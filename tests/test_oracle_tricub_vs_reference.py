"""The oracle's restatement of the tricubic look-up behind fitDL (oracle/tricub_oracle.c) against the reference's own
read_tricub_interp / do_tricubinterp (tests/golden/tricub.npz, made by oracle/ref_probe_tricub.c): bit for bit, the
out-of-table answers (-1) included; the host-side table reader against what the reference kept after reading."""
import os

import numpy as np
import pytest

from imd_amd import tricub
from oracle import pyoracle as po

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "tricub.npz"))
NAMES = ("centred", "end_to_end", "uneven", "narrow")


def _table(name):
    return tricub.TricubTable(G[name + "_range"], G[name + "_x"], G[name + "_y"], G[name + "_z"], G[name + "_f"])


@pytest.mark.parametrize("name", NAMES)
def test_oracle_lookup_equals_the_reference_bit_for_bit(name):
    t = _table(name)
    o = po.OracleTricub(t.range, t.x, t.y, t.z, t.f)
    got = o.eval(G[name + "_queries"])
    o.close()
    ref = G[name + "_ref"]
    assert (ref == -1).sum() == 6
    assert np.array_equal(got, ref), np.abs(got - ref).max()


def test_the_goldens_exercise_all_64_coefficients():
    """tricub_eval's powers pass through int: inside (-1, 1) only a[0] = the node below survives.  The tables with
    nodes off the cell centres give scaled coordinates beyond 1, where every coefficient carries weight"""
    t = _table("centred")
    ref = G["centred_ref"]
    inside = ref != -1
    assert np.isin(ref[inside], t.f).mean() > 0.7           # mostly the node below; the upper ends of the axes blend
    t = _table("uneven")
    ref = G["uneven_ref"]
    assert (~np.isin(ref[ref != -1], t.f)).sum() > 300


@pytest.mark.parametrize("name", NAMES)
def test_table_file_round_trip_keeps_what_the_reference_read(name, tmp_path):
    t = _table(name)
    p = str(tmp_path / "DL.txt")
    tricub.write_table(p, t)
    r = tricub.read_table(p)
    hd = G[name + "_ref_header"]
    assert r.dims == tuple(int(v) for v in hd[:3])
    assert np.array_equal(r.range, hd[3:9])
    assert np.array_equal(r.x, t.x) and np.array_equal(r.y, t.y) and np.array_equal(r.z, t.z) and np.array_equal(r.f, t.f)


def test_fitdl_clamps_and_units():
    """fitDL's wrapper (imd_ttm_3D.c:2939-2966): Te, Ti floored at 0.0259 eV, lg floored at -1.587504 and capped at the
    FIRST table's ymax / zmax; DL[1..4] divided by hbar in eV s; DL[5] the plasma frequency of ne"""
    t = _table("centred")
    tabs = [po.OracleTricub(t.range, t.x, t.y, t.z, t.f * s) for s in (1.0, 2.0, 3.0, 4.0, 5.0)]
    temp = np.array([0.001, 0.05, 3.0, 1e3]); ti = np.array([0.0259, 0.001, 0.2, 50.0])
    dens = np.array([2000.0, 2500.0, 2999.0, 2700.0]); ne = np.array([1e28, 2e28, 1.8e29, 5e28])
    DL, st = po.oracle_fitdl(tabs, temp, ti, dens, ne)
    assert (st == 0).all()
    lgTe = np.minimum(np.maximum(np.log10(np.maximum(temp, 0.0259)), -1.587504), t.range[3])
    lgTi = np.minimum(np.maximum(np.log10(np.maximum(ti, 0.0259)), -1.587504), t.range[5])
    base = tabs[0].eval(np.stack([dens, lgTe, lgTi], axis=1))
    assert np.array_equal(DL[:, 0], base)
    for q in range(1, 5):
        assert np.array_equal(DL[:, q], tabs[q].eval(np.stack([dens, lgTe, lgTi], axis=1)) / 6.582119487171653e-16)
    assert np.allclose(DL[:, 5], np.sqrt(ne * 1.60217662e-19 ** 2 / 8.841941282883075e-12 / 9.10938356e-31), rtol=1e-15)
    DL, st = po.oracle_fitdl(tabs, temp[:1], ti[:1], np.array([100.0]), ne[:1])          # density below the table
    assert st[0] == -1 and DL[0, 0] == -1
    for o in tabs:
        o.close()
