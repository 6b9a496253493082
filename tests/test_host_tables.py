"""Host-side table construction against the reference's own memory (CPU tests).
Row a2 of SURVEY 8: pot_table_t as left by read_pot_table / create_pot_table / init_threepoint
(imd_potential.c:161-466, 476-705, 1260-1276); the goldens hold the tables the reference binary built."""
import os
import tempfile

import numpy as np

from imd_amd import potential
from tests import golden_util as gu


def test_lj_table_creation_bit_exact():
    """BASELINE config 1 (LJ, epsilon 0.392, sigma 2.62, r_cut 6.0, r_begin 1.5, pot_res 2000): the table
    create_lj_table builds equals the reference's pair_pot after create_pot_table bit for bit."""
    t = gu.table(gu.load("lj_nbl_2048"), "pair_pot")
    pt, r2_end = potential.create_lj_table(0.392, 2.62, 6.0, 1.5, 2000)
    assert r2_end == float(t["end"][0])
    for f in ("begin", "end", "step", "invstep"):
        assert np.array_equal(getattr(pt, f), t[f]), f
    assert pt.maxsteps == int(t["maxsteps"]) and pt.ncols == int(t["ncols"])
    assert np.array_equal(pt.table, t["table"])


def test_eam_files_read_back_like_the_reference():
    """write_synthetic_eam_files -> read_pot_table (format 2, tabulated in r^2 / rho, cut-off shift, two extrapolated
    rows) gives the tables the reference holds after reading the same files."""
    G = gu.load("eam_nbl_2048")
    with tempfile.TemporaryDirectory() as d:
        potential.write_synthetic_eam_files(d)
        for fn, key, radial in (("phi.pt", "pair_pot", True), ("rho.pt", "rho_h_tab", True), ("F.pt", "embed_pot", False)):
            pt, _ = potential.read_pot_table(os.path.join(d, fn), 1, radial)
            t = gu.table(G, key)
            for f in ("begin", "end", "step", "invstep"):
                assert np.array_equal(getattr(pt, f), t[f]), (key, f)
            assert np.array_equal(pt.table, t["table"]), key
