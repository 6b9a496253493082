"""BASELINE config 4 at FULL size on one GPU (run with -m gpu): 272^3 FCC unit cells = 80 494 592 atoms as a slab (open x,
vacuum in front of and behind the atoms: shiftx_front / shiftx_rear of the reference's generator, pbc_dirs 0 1 1), EAM +
3D TTM with fd_ext 2 2 2 and 100 sub-steps -- what the 2x2x2 grid of eight GPUs holds together, here resident in one
MI355X's 288 GB (about 60 GB).  No oracle run at this size: size-independent properties, as in
test_bench_size_properties_with_ttm.  What it pins down beyond the 10 M-atom case: every index of the list, the staging
plan and the re-binning survives 8x the atoms (3.9e9 list slots, 1.2e5 tiles, 4.7e5 FD cells), vacuum tiles and the
cost-cut block ranges at scale, the lattice's inactive cells and open boundary."""
import numpy as np
import pytest

from imd_amd.system import GpuSystem

pytestmark = pytest.mark.gpu


def test_config4_80M_atom_slab_with_ttm_on_one_gpu():
    from imd_amd import lattice, potential, ttm as ttm_mod
    from imd_amd.system import global_cell_dim
    nb = (272, 272, 272)
    a = 4.05
    pair, rho, emb, r2cut = potential.synthetic_eam_tables()
    cellsz = (np.sqrt(r2cut) + 0.5) ** 2
    box = lattice.fcc_box(nb, a)
    gd0 = global_cell_dim(box, cellsz, (2, 2, 2))
    hx = box[0, 0] / gd0[0]
    nvac = 12
    box = box.copy(); box[0, 0] += 2 * nvac * hx
    gdim = (gd0[0] + 2 * nvac, gd0[1], gd0[2])
    pos = lattice.fcc_positions(nb, a)
    pos[:, 0] += nvac * hx
    n = len(pos)
    assert n == 80494592
    mom = lattice.maxwell_momenta(n, 26.9815, 0.0258, seed=4)
    dim = [g + 2 for g in gdim]
    s = GpuSystem(box, (0, 1, 1), dim, 1, cellsz, 0.5, 0.1, count_ttm_neighbours=True, global_cell_dim=gdim)
    s.upload_tables(pair, rho, emb)
    s.upload_atoms(pos, mom, 26.9815)
    del pos, mom
    s.set_option("ttm_adv", 1)                 # calc_ttm as bench.py --slab runs it: do_ADV(1.0) after update_fd
    ttm_mod.attach_default(s, fd_ext=(2, 2, 2), diff_substeps=100, fd_min_atoms=5)
    s.calc_forces(0)
    assert s.get_info("list_mode") == 1                                   # the LDS-tile kernels
    ent, mx = s.nblist_stats()
    assert ent % 2 == 0 and ent > 2 ** 32                                 # symmetric list; more slots than 32 bits count
    assert 70 < ent / n < 90 and mx <= s.get_info("W")
    d = s.download(fields=("kraft", "numneighs"))
    f = d["kraft"]
    assert np.isfinite(f).all()
    assert np.abs(f.sum(axis=0)).max() < 1e-5                             # Newton's third law over 8e7 atoms, |F| ~ 0.5
    nn = d["numneighs"]
    bulk = np.bincount(nn).argmax()
    assert 50 <= bulk <= 90 and (np.abs(nn - bulk) <= 3).mean() > 0.9 and nn.min() >= bulk // 3     # surfaces have fewer, nobody has none
    del d, f, nn
    for st in range(4):
        s.md_step(st)
    assert s.num_atoms()[0] == n
    lat = s.ttm_download(fields=("natoms", "natoms_old", "md_temp", "temp", "xi", "U", "flux"))
    inner = (slice(1, -1),) * 3
    nat = lat["natoms"][inner]
    assert int(nat.sum()) == n                                            # every atom sits in exactly one FD cell
    assert int(lat["natoms_old"][inner].sum()) == n                       # do_ADV's two counts: atoms conserved, arrivals cover the gains
    assert (lat["flux"][inner] >= 0).all() and (lat["flux"][inner].sum(axis=-1) >= np.maximum(nat - lat["natoms_old"][inner], 0)).all()
    assert np.isfinite(lat["U"][inner]).all() and (lat["U"][inner][nat >= 5] > 0).all()
    act = nat >= 5
    assert act.sum() == (gd0[0] // 2) * (gd0[1] // 2) * (gd0[2] // 2)     # the slab's cells are active, the vacuum's are not
    assert not act[: nvac // 2].any() and not act[-(nvac // 2):].any()
    ekin = s.get_sums()["tot_kin_energy"] / n
    tl = lat["md_temp"][inner][act]
    assert 0.90 * (2.0 / 3.0) * ekin < tl.mean() < 1.02 * (2.0 / 3.0) * ekin
    te = lat["temp"][inner][act]
    assert te.min() > 0.4 * tl.mean() and te.max() < 1.6 * tl.mean()
    assert np.isfinite(lat["xi"][inner][act]).all()
    ids = s.download(fields=("nummer",))["nummer"]
    assert len(ids) == n and ids.min() == 0 and ids.max() == n - 1 and len(np.unique(ids)) == n
    s.close()
