"""Which diagonal the reference's triangulation puts into every cell of an EOS table grid, WITHOUT the reference: the library's
restatement of the algorithm behind nn-c's delaunay_build (Shewchuk's Triangle: divide-and-conquer with alternating cuts, exact
predicates, its tie rules) -- imdgpu_eos_triangulation, host only -- against
  * tests/golden/eos_triangulation.npz: the answers of nn_interpol/libnn.a itself for 96 rectilinear grids (oracle/ref_probe_tri.c),
    uniform, geometric / quadratic and random node spacings;
  * tests/golden/eos_grid.npz: the diagonals read off the reference PROGRAM's interpolators (lpi_interpolate_point at the cell
    centres of curved tables).
CPU only (the function needs no GPU)."""
import numpy as np
import pytest

from imd_amd import lib
from tests import golden_util as gu


def native(nx, ny):
    L = lib.load()
    d = np.zeros((nx - 1, ny - 1), np.uint8)
    assert L.imdgpu_eos_triangulation(int(nx), int(ny), d.ctypes.data) == 0
    return d


def test_equal_to_the_reference_triangulator_on_96_grids():
    G = gu.load("eos_triangulation")
    n = int(G["n"])
    assert n == 96
    kinds = set()
    for q in range(n):
        ref = G["g%d_diag" % q]
        nx, ny = len(G["g%d_x" % q]), len(G["g%d_y" % q])
        assert ref.shape == (nx - 1, ny - 1)
        assert np.array_equal(native(nx, ny), ref), (q, nx, ny)
        x = G["g%d_x" % q]
        kinds.add("uniform" if np.allclose(np.diff(x), np.diff(x)[0]) else "uneven")
    assert kinds == {"uniform", "uneven"}


def test_the_map_depends_on_the_grid_size_only():
    """the fixture holds grids of the same size with different node coordinates: the reference's own answers agree"""
    G = gu.load("eos_triangulation")
    by_size = {}
    for q in range(int(G["n"])):
        by_size.setdefault(G["g%d_diag" % q].shape, []).append(G["g%d_diag" % q])
    several = [v for v in by_size.values() if len(v) > 1]
    assert len(several) >= 3
    for v in several:
        assert all(np.array_equal(v[0], w) for w in v[1:])


@pytest.mark.parametrize("c", ["a", "b"])
def test_equal_to_the_diagonals_of_the_reference_programs_interpolators(c):
    from tests.test_eos_grid import triangulation_from_golden
    G = gu.load("eos_grid")
    for which in (0, 1):
        ref = triangulation_from_golden(G, c, which)
        assert np.array_equal(native(ref.shape[0] + 1, ref.shape[1] + 1), ref)
    # not a trivial pattern: both diagonals occur
    assert (ref == 1).sum() > 10 and (ref == 2).sum() > 10


def test_rejects_degenerate_grids():
    L = lib.load()
    d = np.zeros(4, np.uint8)
    assert L.imdgpu_eos_triangulation(1, 5, d.ctypes.data) != 0
    assert L.imdgpu_eos_triangulation(3, 3, None) != 0
