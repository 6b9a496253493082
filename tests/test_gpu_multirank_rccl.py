"""N > 1 ranks on the RCCL transport, ONE PROCESS PER GPU (run with -m gpu): the parity cases of tests/test_gpu_multirank.py and
tests/test_gpu_multirank_local.py with torch.distributed on nccl, rank r on cuda:r, the library's own communicators
(imdgpu_dist_init_rccl) carrying halos, flags, the lattice all-gather and do_COMMFLUX's columns over xGMI, the decomposed
lattice's ghost records in peer memory through hipIpc.  Every case is SKIPPED unless the box shows at least as many devices as
the case has ranks -- on a one-GPU box only the one-rank dry runs at the end execute (self messages and one-rank collectives
through RCCL, IMDGPU_DIST_SELF_COMM) -- and switches itself on at the first multi-GPU lease, before `bench.py --gpus N` is
believed.  Expected values: the reference's own runs (tests/golden/*, several of them under mpiexec -n 8 / -n 2).
Reference: imd_comm_force_3d.c:222-396, imd_fix_cells_3d.c:436-542, imd_ttm_3D.c:667-995,3314-3504."""
import numpy as np
import pytest

from tests import golden_util as gu
from tests.test_gpu_multirank import _run

pytestmark = pytest.mark.gpu


def _ndev():
    try:
        import torch
        return int(torch.cuda.device_count())          # counting devices does not initialise the GPU
    except Exception:
        return 0


NDEV = _ndev()


def needs(world):
    return pytest.mark.skipif(NDEV < world, reason="RCCL transport with one process per GPU: %d ranks need %d devices, this box shows %d" % (world, world, NDEV))


def _atoms(res, G, step, forces=True):
    ids = np.concatenate([r[1]["nummer"] for r in res]); pos = np.concatenate([r[1]["ort"] for r in res])
    pre = "s%d_" % step
    assert len(ids) == len(G[pre + "nummer"]) == len(np.unique(ids))
    o, go = np.argsort(ids), np.argsort(G[pre + "nummer"])
    box = np.diag(G["box"])
    dx = pos[o] - G[pre + "ort"][go]; dx -= np.round(dx / box) * box
    assert np.abs(dx).max() < 1e-9
    if forces:
        frc = np.concatenate([r[1]["kraft"] for r in res])
        assert np.abs(frc[o] - G[pre + "kraft"][go]).max() < 1e-9 * np.abs(G[pre + "kraft"]).max()
        assert abs(res[0][2]["tot_pot_energy"] - float(G[pre + "tot_pot_energy"])) < 1e-9 * abs(float(G[pre + "tot_pot_energy"]))
    assert all(r[7] == "rccl" for r in res), [r[7] for r in res]
    return o, go


def _lattice(res, G, step, decomposed, fields=("temp", "xi"), tol=1e-10):
    inner = (slice(1, -1),) * 3
    pre = "t%d_substeps_done_" % step
    for r in res:
        lat, c = r[4], r[5]
        if decomposed:
            n = np.array(lat["temp"].shape) - 2
            sl = tuple(slice(1 + c[q] * n[q], 1 + (c[q] + 1) * n[q]) for q in range(3))
        else:
            sl = inner
        assert np.array_equal(lat["natoms"][inner], G[pre + "natoms"][sl])
        for f in fields:
            ref = G[pre + f]
            assert np.abs(lat[f][inner] - ref[sl]).max() <= tol * np.abs(ref[inner]).max(), (r[0], f)
    if decomposed:            # faces through ghost records in the PEER GPU's memory (hipIpc, fine-grained), not the host mailbox
        assert all(r[6][0] >= 1 and r[6][1] == 2 for r in res), [r[6] for r in res]


@pytest.mark.parametrize("cpu_dim", [pytest.param((2, 1, 1), marks=needs(2)), pytest.param((1, 1, 2), marks=needs(2)),
                                     pytest.param((2, 2, 1), marks=needs(4)), pytest.param((2, 2, 2), marks=needs(8))])
def test_nve_single_domain_fixture(cpu_dim):
    G = gu.load("eam_nbl_2048")
    res = _run("eam_nbl_2048", cpu_dim, 20, False, backend="nccl")
    _atoms(res, G, 20)
    assert all(r[3] >= 2 for r in res)


@needs(8)
def test_nve_2x2x2_against_the_reference_on_8_mpi_ranks():
    G = gu.load("eam_nbl_16384_mpi8")
    res = _run("eam_nbl_16384_mpi8", (2, 2, 2), 20, False, backend="nccl")
    _atoms(res, G, 20)


@pytest.mark.parametrize("decomposed", [False, True])
@pytest.mark.parametrize("cpu_dim", [pytest.param((2, 1, 1), marks=needs(2)), pytest.param((2, 2, 1), marks=needs(4))])
def test_ttm_single_domain_fixture(cpu_dim, decomposed):
    G = gu.load("ttm_2048")
    res = _run("ttm_2048", cpu_dim, 3, True, decomposed=decomposed, backend="nccl")
    _atoms(res, G, 3, forces=False)
    _lattice(res, G, 2, decomposed, fields=("temp",))


@needs(8)
@pytest.mark.parametrize("decomposed", [False, True])
def test_ttm_2x2x2_against_the_reference_on_8_mpi_ranks(decomposed):
    G = gu.load("ttm_16384_mpi8")
    res = _run("ttm_16384_mpi8", (2, 2, 2), 3, True, decomposed=decomposed, backend="nccl")
    _atoms(res, G, 3, forces=False)
    _lattice(res, G, 2, decomposed)


@needs(2)
@pytest.mark.parametrize("decomposed", [False, True])
def test_ttm_with_advection_two_ranks_along_x(decomposed):
    """do_ADV with do_COMMFLUX inside the native step (decomposed) / flux counters in the all-gathered blocks (replicated)"""
    G = gu.load("ttm_adv")
    res = _run("ttm_adv", (2, 1, 1), 49, True, decomposed=decomposed, adv=True, backend="nccl")
    nx = (int(G["ttm_fd_dim"][0]) - 2) // 2
    inner = (slice(1, -1),) * 3
    for r in res:
        sl = (slice(1 + r[5][0] * nx, 1 + (r[5][0] + 1) * nx), slice(1, -1), slice(1, -1)) if decomposed else inner
        assert np.array_equal(r[4]["flux"][inner], G["t48_flux"][sl])
        for f in ("temp", "U", "temp2", "U2", "xi"):
            a, b = r[4][f][inner], G["t48_substeps_done_" + f][sl]
            assert np.abs(a - b).max() <= 2e-8 * np.abs(b).max(), f
    assert all(r[7] == "rccl" for r in res)


@needs(8)
@pytest.mark.parametrize("decomposed", [False, True])
def test_ttm_with_advection_2x2x2_against_the_reference_on_8_mpi_ranks(decomposed):
    """the headline's calc_ttm sequence on config 4's layout (open-x slab, 2 x 2 x 2), 27 steps with three list rebuilds"""
    G = gu.load("ttm_adv_mpi8")
    last = int(G["steps"][-1])
    res = _run("ttm_adv_mpi8", (2, 2, 2), last, True, decomposed=decomposed, adv=True, backend="nccl")
    _atoms(res, G, last, forces=False)
    _lattice(res, G, last - 1, decomposed, fields=("temp", "U", "temp2", "U2", "xi"), tol=2e-8)
    inner = (slice(1, -1),) * 3
    for r in res:
        n = np.array(r[4]["temp"].shape) - 2
        sl = tuple(slice(1 + r[5][q] * n[q], 1 + (r[5][q] + 1) * n[q]) for q in range(3)) if decomposed else inner
        assert np.array_equal(r[4]["flux"][inner], G["t%d_flux" % (last - 1)][sl])


@needs(2)
def test_slab_on_two_ranks_with_five_layers_each():
    G = gu.load("ttm_slab_mpi2")
    res = _run("ttm_slab_mpi2", (2, 1, 1), 3, True, decomposed=True, backend="nccl")
    _atoms(res, G, 3, forces=False)
    assert all(r[6][0] >= 1 and r[6][1] == 2 for r in res), [r[6] for r in res]
    ref, nat = G["t2_substeps_done_temp"], G["t2_substeps_done_natoms"]
    for r in res:
        lat = r[4]["temp"]
        n = lat.shape[0] - 2
        sl = slice(r[5][0] * n, r[5][0] * n + n + 2)
        act = nat[sl][1:-1, 1:-1, 1:-1] >= 5
        assert np.abs(lat[1:-1, 1:-1, 1:-1] - ref[sl][1:-1, 1:-1, 1:-1])[act].max() < 1e-10 * np.abs(ref).max()


# ---- one rank: the same worker and the same RCCL calls on a one-rank communicator (runs on every box) -------------------------
@pytest.mark.parametrize("case,ttm", [("eam_nbl_2048", False), ("ttm_2048", True)])
def test_dry_run_on_one_rank_with_self_messages_through_rccl(case, ttm, monkeypatch):
    monkeypatch.setenv("IMDGPU_DIST_SELF_COMM", "1")
    G = gu.load(case)
    nsteps = 3 if ttm else 20
    res = _run(case, (1, 1, 1), nsteps, ttm, backend="nccl")
    _atoms(res, G, nsteps, forces=not ttm)
    if ttm:
        _lattice(res, G, 2, False, fields=("temp",))
