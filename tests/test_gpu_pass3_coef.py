"""EAM force pass in its two forms (one atom type; pair_pot and rho_h_tab on one r^2 grid or -- case *_dg -- each on its own: the
coefficients are rho_h_tab's and the rows pass 1 reports are rows of rho_h_tab): k_pass3_tile_coef reads
one (A, B) coefficient pair per neighbour from the rows [klo, ncoef) it holds in LDS; k_pass3_tile reads three
table rows and stands in whenever a pair of the step falls below row klo (pass 1 reports the lowest row).  Both are
DERIV_FUNC2 (potaccess.h:591-621) and agree to rounding (1e-11 of max|F|); the stand-in IS the three-row kernel, so a step it takes
over equals `p3_coef = 0` bit for bit."""
import numpy as np
import pytest

from tests import golden_util as gu
from tests.test_gpu_forces import FTOL, _gpu_from_golden, _relmax

pytestmark = pytest.mark.gpu

CASES = [("eam_nbl_2048", 40), ("eam_nbl_slab", 10), ("eam_nbl_2048", 0), ("eam_nbl_2048_dg", 40)]


def _forces(G, step, **opts):
    s = _gpu_from_golden(G, step)
    for k, v in opts.items():
        s.set_option(k, v)
    r = s.calc_forces(step)
    d = s.download()
    info = (s.get_info("p3_coef_rows"), s.get_info("p3_coef_klo"))
    s.close()
    return d, r, info


@pytest.mark.parametrize("case,step", CASES)
def test_coefficient_form_against_three_row_form_and_golden(case, step):
    G = gu.load(case)
    pre = "s%d_" % step
    dc, rc, ic = _forces(G, step)
    do, ro, io = _forces(G, step, p3_coef=0)
    assert ic[0] > 0 and ic[1] >= 0, "the coefficient form is the default for one type on a shared grid"
    assert io == (0, -1)
    for d, r in ((dc, rc), (do, ro)):
        assert _relmax(d["kraft"], G[pre + "kraft"], floor=1e-3) < FTOL
        assert abs(r["virial"] - float(G[pre + "virial"])) < 1e-10 * abs(float(G[pre + "virial"]))
    # the coefficient form takes r2 with fused multiply-adds (an ulp of r2 moves chi by ~1e-12 and the gradient by ~1e-15 of
    # itself); A = dv - d2v/2 is rounded once more: 1e-11 of max|F| (floor: the perfect lattice's forces vanish) is three orders
    # inside the parity bar
    assert _relmax(dc["kraft"], do["kraft"], floor=1e-3) < 1e-11
    assert abs(rc["virial"] - ro["virial"]) < 1e-11 * abs(ro["virial"])
    # everything pass 1 writes is untouched by the choice
    for k in ("pot_eng", "eam_rho", "eam_dF"):
        assert np.array_equal(dc[k], do[k])
    assert rc["is_short"] == ro["is_short"] == 0


@pytest.mark.parametrize("case,step", CASES[:2] + CASES[3:])
def test_pairs_below_the_rows_in_lds_fall_back_to_the_three_row_kernel(case, step):
    G = gu.load(case)
    do, ro, _ = _forces(G, step, p3_coef=0)
    ncoef = int(G["rho_h_tab_maxsteps"]) + 1
    # keep (nearly) every row out of the LDS: every pair inside the cut-off lies below klo
    dh, rh, ih = _forces(G, step, p3_klo=1 << 20)
    assert ih == (1, ncoef - 1)
    assert np.array_equal(dh["kraft"], do["kraft"]) and rh["virial"] == ro["virial"]
    # klo just above the lowest row in use: still the stand-in; at the lowest row in use: the coefficient form
    t = gu.table(G, "rho_h_tab")
    ort = G["s%d_ort" % step].reshape(-1, 3)
    box = np.asarray(G["box"]).reshape(3, 3)
    L = np.diag(box)
    dmin2 = np.inf
    for i in range(0, len(ort), max(1, len(ort) // 256)):               # a sample of atoms is enough for the minimum image search
        d = ort - ort[i]
        d -= np.round(d / L) * L * np.asarray(G["pbc_dirs"], float)
        r2 = (d * d).sum(1)
        r2[i] = np.inf
        dmin2 = min(dmin2, r2.min())
    kmin_sample = int((dmin2 - float(t["begin"][0])) * float(t["invstep"][0]))
    dl, rl, il = _forces(G, step, p3_klo=max(kmin_sample - 40, 1))
    assert il[1] == max(kmin_sample - 40, 1)
    assert _relmax(dl["kraft"], do["kraft"], floor=1e-3) < 1e-11
    dhh, rhh, _ = _forces(G, step, p3_klo=kmin_sample + 1)
    assert np.array_equal(dhh["kraft"], do["kraft"]) and rhh["virial"] == ro["virial"]


def test_md_steps_with_either_form_stay_together():
    G = gu.load("eam_nbl_2048")
    out = []
    for coef in (1, 0):
        s = _gpu_from_golden(G, 0)
        s.set_option("p3_coef", coef)
        for n in range(30):
            s.md_step(n)
        out.append(s.download())
        s.close()
    assert np.array_equal(out[0]["nummer"], out[1]["nummer"])
    assert np.abs(out[0]["ort"] - out[1]["ort"]).max() < 1e-11
    assert np.abs(out[0]["impuls"] - out[1]["impuls"]).max() < 1e-11 * max(np.abs(out[1]["impuls"]).max(), 1.0)
