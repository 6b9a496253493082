"""GPU test (run with -m gpu): the multi-rank host layer (imd_amd.parallel: halo plan, device gather/scatter,
migration, TTM fork/join) driven on ONE rank with the library's external_halo option must reproduce the native
single-rank step (imdgpu_md_step), whose ghosts are periodic images made inside the library."""
import numpy as np
import pytest

from imd_amd import lattice, potential
from imd_amd.system import GpuSystem, global_cell_dim

pytestmark = pytest.mark.gpu


def _system(external):
    a, nb = 4.05, (12, 12, 12)
    pair, rho, emb, r2cut = potential.synthetic_eam_tables()
    cellsz = (np.sqrt(r2cut) + 0.5) ** 2
    box = lattice.fcc_box(nb, a)
    gdim = global_cell_dim(box, cellsz, (2, 2, 2))
    s = GpuSystem(box, (1, 1, 1), tuple(g + 2 for g in gdim), 1, cellsz, 0.5, 0.1, global_cell_dim=gdim,
                  count_ttm_neighbours=True, external_halo=external)
    s.upload_tables(pair, rho, emb)
    from imd_amd import ttm as ttm_mod
    ttm_mod.attach_default(s, fd_ext=(2, 2, 2), diff_substeps=20, fd_min_atoms=5)
    pos = lattice.fcc_positions(nb, a)
    mom = lattice.maxwell_momenta(len(pos), 26.9815, 0.06, seed=7)
    s.upload_atoms(pos, mom, 26.9815, None, np.arange(len(pos), dtype=np.int32))
    return s, gdim, box


@pytest.mark.parametrize("self_comm", [False, True])
def test_distributed_layer_on_one_rank_equals_native_step(self_comm, monkeypatch):
    """self_comm: the 26 self-neighbour messages, the flag all-reduce and the TTM all-gather really go through RCCL
    (ncclSend/ncclRecv to the own rank in one group, collectives on a one-rank communicator) instead of the local
    short cuts -- the only way to execute the RCCL calls of imdgpu_dist.hip on a one-GPU box."""
    import torch
    from imd_amd.parallel import Decomposition, DistributedMD, GpuOps
    if self_comm:
        monkeypatch.setenv("IMDGPU_DIST_SELF_COMM", "1")
    ref, gdim, box = _system(False)
    s, _, _ = _system(True)
    ops = GpuOps(s, 0)
    md = DistributedMD(Decomposition(0, (1, 1, 1), gdim, box, (1, 1, 1)), ops, ttm=True)
    assert ops.transport == "rccl"
    for step in range(25):                       # two list rebuilds at this temperature
        ref.md_step(step)
        md.md_step(step)
    ref.calc_forces(25); md.calc_forces()        # (the native step leaves the NEXT step's pass 1 in the force arrays)
    s.synchronize(); torch.cuda.synchronize()
    a, b = ref.download(), s.download()
    oa, ob = np.argsort(a["nummer"]), np.argsort(b["nummer"])
    assert ref.get_sums()["nbl_count"] == s.get_sums()["nbl_count"] >= 2
    assert np.abs(a["ort"][oa] - b["ort"][ob]).max() < 1e-10
    assert np.abs(a["kraft"][oa] - b["kraft"][ob]).max() < 1e-9 * np.abs(a["kraft"]).max()
    assert np.abs(a["impuls"][oa] - b["impuls"][ob]).max() < 1e-10 * np.abs(a["impuls"]).max()
    ta, tb = ref.ttm_download(), s.ttm_download()
    inner = (slice(1, -1),) * 3
    assert np.allclose(ta["temp"][inner], tb["temp"][inner], rtol=1e-10, atol=0)
    assert np.array_equal(ta["natoms"][inner], tb["natoms"][inner])
    ref.close(); s.close()
