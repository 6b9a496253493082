"""BASELINE config 4's rank grid on the device path (run with -m gpu): 2x2x2 and a z split (1x1x2), NVE and TTM with the
replicated and the decomposed lattice, every rank with its own libimdgpu context on cuda:0 and REAL device operations
-- native rebuild (imdgpu_dist_rebuild: migration, counts, ghost layout, send lists, list) and native step
(imdgpu_dist_md_step) over the library's in-process transport, the ranks being threads of one child process (a GPU box
admits at most six processes on its card).  Expected values: the reference ITSELF on 2x2x2 MPI ranks
(tests/golden/eam_nbl_16384_mpi8.npz, ttm_16384_mpi8.npz: mpiexec -n 8, every rank's memory dumped and assembled) and
the single-domain fixtures for the z split.
Reference: setup_mpi_topology imd_geom_mpi_3d.c:32-90, send_cells imd_comm_force_3d.c:222-396, send_atoms
imd_fix_cells_3d.c:436-542, ttm_fill_ghost_layers imd_ttm_3D.c:3314-3504."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from tests import golden_util as gu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(case, cpu_dim, nsteps, ttm=False, decomposed=False, adv=False, env=None):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "res.npz")
        env = dict(os.environ, GPU_MAX_HW_QUEUES="32", PYTHONPATH=ROOT, **(env or {}))     # 8 ranks x 2 streams: polling lattice kernels of different ranks must not share a queue
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "local_ranks_worker.py"), case] + [str(v) for v in cpu_dim] +
                           [str(nsteps), str(3 if adv else int(ttm)), str(int(decomposed)), out], env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=900)
        assert r.returncode == 0, "local ranks %s failed:\n%s" % (str(cpu_dim), r.stdout[-4000:])
        R = dict(np.load(out))
    world = int(np.prod(cpu_dim))
    assert "in-process" in str(R["transport"])
    return R, world


def _cat(R, world, k):
    return np.concatenate([R["r%d_%s" % (r, k)] for r in range(world)])


def _check_atoms(R, world, G, step, forces=True):
    ids, pos = _cat(R, world, "nummer"), _cat(R, world, "ort")
    pre = "s%d_" % step
    assert len(ids) == len(G[pre + "nummer"]) == len(np.unique(ids))
    o, go = np.argsort(ids), np.argsort(G[pre + "nummer"])
    box = np.diag(G["box"])
    dx = pos[o] - G[pre + "ort"][go]; dx -= np.round(dx / box) * box
    if not np.abs(dx).max() < 1e-9:                   # say where: which ranks hold the deviating atoms, and how the lattice launches went
        owner = np.concatenate([np.full(len(R["r%d_nummer" % r]), r) for r in range(world)])[o]
        bad = np.abs(dx).max(axis=1) >= 1e-9
        info = {k: [int(R["r%d_%s" % (r, k)]) for r in range(world)] for k in ("nrebuild", "persist", "remote", "persist_runs", "persist_redone")}
        raise AssertionError("positions differ by up to %.3e A: %d atoms, by rank %s; %s" %
                             (np.abs(dx).max(), int(bad.sum()), np.bincount(owner[bad], minlength=world).tolist(), info))
    if forces:
        frc = _cat(R, world, "kraft")
        assert np.abs(frc[o] - G[pre + "kraft"][go]).max() < 1e-9 * np.abs(G[pre + "kraft"]).max()
        assert abs(R["r0_tot_pot_energy"] - float(G[pre + "tot_pot_energy"])) < 1e-9 * abs(float(G[pre + "tot_pot_energy"]))
    return o, go


def test_2x2x2_nve_against_the_reference_on_8_mpi_ranks():
    """16 384 atoms on 2x2x2 ranks, 20 steps with a list rebuild and migration on the way: positions, forces, total energy
    of the reference's own 8-rank run"""
    G = gu.load("eam_nbl_16384_mpi8")
    R, world = _run("eam_nbl_16384_mpi8", (2, 2, 2), 20)
    _check_atoms(R, world, G, 20)
    assert all(int(R["r%d_nrebuild" % r]) >= 2 for r in range(world))
    assert all(int(R["r%d_list_mode" % r]) == 1 for r in range(world))          # the LDS-tile kernels, not the fallback
    assert int(R["r0_natoms"]) == 16384


@pytest.mark.parametrize("decomposed", [False, True])
def test_2x2x2_ttm_against_the_reference_on_8_mpi_ranks(decomposed):
    """the same grid with the 3D TTM (fd_ext 1 1 1: an 8^3 lattice, 4^3 per rank): atoms after 3 steps and the electron
    lattice after the sub-steps of step 2 equal the reference's 8-rank run -- replicated lattice (all-gathered update_fd
    blocks) and decomposed lattice (one block per rank, ghost faces exchanged per sub-step)"""
    G = gu.load("ttm_16384_mpi8")
    R, world = _run("ttm_16384_mpi8", (2, 2, 2), 3, ttm=True, decomposed=decomposed)
    o, go = _check_atoms(R, world, G, 3, forces=False)
    nn = _cat(R, world, "numneighs")
    assert np.array_equal(nn[o], G["s3_numneighs"][go])
    ref = G["t2_substeps_done_temp"][1:-1, 1:-1, 1:-1]
    refx = G["t2_substeps_done_xi"][1:-1, 1:-1, 1:-1]
    if not decomposed:
        for r in range(world):
            assert np.array_equal(R["r%d_lat_natoms" % r][1:-1, 1:-1, 1:-1], G["t2_substeps_done_natoms"][1:-1, 1:-1, 1:-1])
            assert np.abs(R["r%d_lat_temp" % r][1:-1, 1:-1, 1:-1] - ref).max() < 1e-10 * np.abs(ref).max()
            assert np.abs(R["r%d_lat_xi" % r][1:-1, 1:-1, 1:-1] - refx).max() < 1e-9 * np.abs(refx).max()
    else:
        # not the host mailbox: every rank's persistent kernel exchanged its faces through ghost records in its neighbours'
        # device memory (steps 1 and 2; step 0 runs un-captured through the mailbox)
        assert all(int(R["r%d_remote" % r]) >= 1 and int(R["r%d_persist" % r]) == 2 and int(R["r%d_persist_runs" % r]) >= 1 for r in range(world)), \
            [(int(R["r%d_remote" % r]), int(R["r%d_persist" % r]), int(R["r%d_persist_runs" % r])) for r in range(world)]
        full = np.zeros_like(ref)
        for r in range(world):
            blk = R["r%d_lat_temp" % r][1:-1, 1:-1, 1:-1]
            c, n = R["r%d_coord" % r], blk.shape
            full[c[0] * n[0]:(c[0] + 1) * n[0], c[1] * n[1]:(c[1] + 1) * n[1], c[2] * n[2]:(c[2] + 1) * n[2]] = blk
        assert np.abs(full - ref).max() < 1e-10 * np.abs(ref).max()
        # the ghost shell every rank holds after the last sub-step = its neighbours' boundary cells (ttm_fill_ghost_layers)
        refg = G["t2_substeps_done_temp"]
        for r in range(world):
            lat, c = R["r%d_lat_temp" % r], R["r%d_coord" % r]
            n = np.array(lat.shape) - 2
            lo = np.array(c) * n                                              # my first inner cell in the global inner lattice
            gi = (np.arange(-1, n[0] + 1)[:, None, None] + lo[0]) % ref.shape[0]
            gj = (np.arange(-1, n[1] + 1)[None, :, None] + lo[1]) % ref.shape[1]
            gk = (np.arange(-1, n[2] + 1)[None, None, :] + lo[2]) % ref.shape[2]
            want = ref[gi, gj, gk]
            for face in (np.s_[0, 1:-1, 1:-1], np.s_[-1, 1:-1, 1:-1], np.s_[1:-1, 0, 1:-1], np.s_[1:-1, -1, 1:-1],
                         np.s_[1:-1, 1:-1, 0], np.s_[1:-1, 1:-1, -1]):
                assert np.abs(lat[face] - want[face]).max() < 1e-10 * np.abs(ref).max()


@pytest.mark.parametrize("decomposed", [False, True])
def test_2x2x2_ttm_with_advection_against_the_reference_on_8_mpi_ranks(decomposed):
    """The headline's calc_ttm sequence (update_fd, do_ADV with do_COMMFLUX, activation, FILLMESH, 100 sub-steps) on config 4's
    layout: an open-x slab on 2 x 2 x 2 ranks, 27 steps through the NATIVE step (imdgpu_dist_md_step; nothing of the step is
    driven from Python) with three list rebuilds at which atoms change FD cells -- also across the rank boundary along x, in
    both directions -- against the reference's own `mpiexec -n 8` run with IMD_PROBE_ADV (tests/golden/ttm_adv_mpi8.npz).
    Replicated lattice: every rank advects the plane k = 1 of every rank block, like the eight ranks of the reference do
    between them; decomposed lattice: do_COMMFLUX's columns travel as one point-to-point batch inside the step
    (imdgpu_dist.hip: dist_commflux; imd_ttm_3D.c:667-995,1321-1458)."""
    G = gu.load("ttm_adv_mpi8")
    last = int(G["steps"][-1])
    assert last == 27 and int(G["t25_flux"].sum()) > 0 and int(G["t17_flux"][8:10].sum()) > 0     # flux next to the rank boundary (cells 8 | 9)
    R, world = _run("ttm_adv_mpi8", (2, 2, 2), last, ttm=True, decomposed=decomposed, adv=True)
    o, go = _check_atoms(R, world, G, last, forces=False)
    nn = _cat(R, world, "numneighs")
    assert np.array_equal(nn[o], G["s%d_numneighs" % last][go])
    assert all(int(R["r%d_nrebuild" % r]) >= 4 for r in range(world))
    inner = (slice(1, -1),) * 3
    pre = "t%d_" % (last - 1)
    for r in range(world):
        lat = {k: R["r%d_lat_%s" % (r, k)] for k in ("temp", "xi", "natoms", "U", "temp2", "U2", "flux")}
        if decomposed:
            c, n = R["r%d_coord" % r], np.array(lat["temp"].shape) - 2
            sl = tuple(slice(1 + c[q] * n[q], 1 + (c[q] + 1) * n[q]) for q in range(3))
        else:
            sl = inner
        assert np.array_equal(lat["flux"][inner], G[pre + "flux"][sl]), r
        assert np.array_equal(lat["natoms"][inner], G[pre + "substeps_done_natoms"][sl]), r
        for f in ("temp", "U", "temp2", "U2", "xi"):
            a, b = lat[f][inner], G[pre + "substeps_done_" + f][sl]
            assert np.abs(a - b).max() <= 2e-8 * np.abs(G[pre + "substeps_done_" + f][inner]).max(), (r, f)   # the inverse EOS's tolerance (test_gpu_ttm)
    if decomposed:        # the faces went through ghost records in the neighbours' device memory, not through the host
        assert all(int(R["r%d_remote" % r]) >= 1 and int(R["r%d_persist" % r]) == 2 for r in range(world)), \
            [(int(R["r%d_remote" % r]), int(R["r%d_persist" % r])) for r in range(world)]


def test_slab_on_two_ranks_with_five_layers_each_takes_the_remote_records():
    """what a rank of BASELINE config 4 holds, at fixture size: a slab with vacuum on 2 x 1 x 1 ranks, FIVE FD layers per rank
    (an extent of 4 q + 1, the outer layer empty).  The persistent kernel anchors its blocks on the filled layers and hands
    the face over through ghost records in the neighbour's device memory; lattice of step 2 and atoms of step 3 against the
    reference's own two-rank run (tests/golden/ttm_slab_mpi2.npz); ttm_fill_ghost_layers imd_ttm_3D.c:3314-3504"""
    G = gu.load("ttm_slab_mpi2")
    R, world = _run("ttm_slab_mpi2", (2, 1, 1), 3, ttm=True, decomposed=True)
    _check_atoms(R, world, G, 3, forces=False)
    assert all(int(R["r%d_remote" % r]) >= 1 and int(R["r%d_persist" % r]) == 2 and int(R["r%d_persist_runs" % r]) >= 1 for r in range(world)), \
        [(int(R["r%d_remote" % r]), int(R["r%d_persist" % r]), int(R["r%d_persist_runs" % r])) for r in range(world)]
    ref = G["t2_substeps_done_temp"]
    nat = G["t2_substeps_done_natoms"]
    for r in range(world):
        lat, latn = R["r%d_lat_temp" % r], R["r%d_lat_natoms" % r]
        n = lat.shape[0] - 2
        assert n == 5
        sl = slice(r * n, r * n + n + 2)                  # my block with its ghost columns in the global lattice (x only is split)
        act = nat[sl][1:-1, 1:-1, 1:-1] >= 5
        assert np.array_equal(latn[1:-1, 1:-1, 1:-1], nat[sl][1:-1, 1:-1, 1:-1])
        assert np.abs(lat[1:-1, 1:-1, 1:-1] - ref[sl][1:-1, 1:-1, 1:-1])[act].max() < 1e-10 * np.abs(ref).max()
        face = (-1 if r == 0 else 0, slice(1, -1), slice(1, -1))      # the ghost column towards the other rank
        assert np.array_equal(latn[face], nat[sl][face])
        assert np.abs(lat[face] - ref[sl][face]).max() < 1e-10 * np.abs(ref).max()


def test_a_missed_census_on_one_of_eight_ranks_is_repeated_by_all():
    """fail soft on several ranks (replicated lattice): rank 3's persistent launch of step 4 steps aside; its status rides in the
    step's flag all-reduce, that rank issues its sub-steps through the launch chain and integrates again, every rank repeats
    halo, flags and the pass enqueued ahead -- the atoms of all ranks equal the undisturbed run's, bit for bit"""
    A, world = _run("ttm_16384_mpi8", (2, 2, 2), 7, ttm=True)
    B, _ = _run("ttm_16384_mpi8", (2, 2, 2), 7, ttm=True, env=dict(IMD_TEST_PERSIST_MISS="3:4"))
    assert int(B["r3_persist_redone"]) == 1 and all(int(B["r%d_persist_redone" % r]) == 0 for r in range(world) if r != 3)
    assert all(int(B["r%d_persist" % r]) == 2 for r in range(world))
    for r in range(world):
        for k in ("nummer", "ort", "impuls", "lat_temp", "lat_xi"):
            assert np.array_equal(A["r%d_%s" % (r, k)], B["r%d_%s" % (r, k)]), (r, k)


@pytest.mark.parametrize("cpu_dim", [(1, 1, 2), (2, 2, 2)])
def test_z_split_and_cube_on_the_single_domain_fixture(cpu_dim):
    """2 048 atoms (4x4x4 MD cells: two cells per rank and direction on 2x2x2 -- every cell of a rank is a boundary cell and
    most peers are reached in several directions) against the reference's single-domain trajectory"""
    G = gu.load("eam_nbl_2048")
    R, world = _run("eam_nbl_2048", cpu_dim, 20)
    _check_atoms(R, world, G, 20)


@pytest.mark.parametrize("cpu_dim", [(2, 1, 1), (2, 2, 2)])
def test_pass1_under_the_position_halo(cpu_dim):
    """option halo_overlap: the position halo of the next step travels on its own stream while pass 1 -- tiles from a work queue,
    two launches -- starts on the tiles that need no atom of another rank.  Same trajectory as the reference's 8-rank run; on
    2x1x1 the middle tile layers along x are interior, on 2x2x2 every tile waits for the halo (the second launch does it all)"""
    G = gu.load("eam_nbl_16384_mpi8")
    env = dict(IMD_TEST_HALO_OVERLAP="8")
    if cpu_dim == (2, 1, 1):
        env["IMDGPU_TILE"] = "1,2,2"            # four cells per rank along x: four tile layers, the middle two interior
    R, world = _run("eam_nbl_16384_mpi8", cpu_dim, 20, env=env)
    _check_atoms(R, world, G, 20)
    assert all(int(R["r%d_overlapped" % r]) >= 15 for r in range(world)), [int(R["r%d_overlapped" % r]) for r in range(world)]
    inner = [int(R["r%d_interior_tiles" % r]) for r in range(world)]
    assert all(v >= 0 for v in inner)
    if cpu_dim == (2, 1, 1):
        assert all(0 < v < int(R["r%d_ntiles" % r]) for r, v in enumerate(inner)), inner


def test_pass1_under_the_position_halo_with_the_lattice():
    """the same with the TTM lattice chain forked beside the force pass (decomposed lattice, 2x2x2)"""
    G = gu.load("ttm_16384_mpi8")
    R, world = _run("ttm_16384_mpi8", (2, 2, 2), 3, ttm=True, decomposed=True, env=dict(IMD_TEST_HALO_OVERLAP="8"))
    o, go = _check_atoms(R, world, G, 3, forces=False)
    assert np.array_equal(_cat(R, world, "numneighs")[o], G["s3_numneighs"][go])
    assert all(int(R["r%d_overlapped" % r]) >= 2 for r in range(world))
    ref = G["t2_substeps_done_temp"][1:-1, 1:-1, 1:-1]
    full = np.zeros_like(ref)
    for r in range(world):
        blk = R["r%d_lat_temp" % r][1:-1, 1:-1, 1:-1]
        c, n = R["r%d_coord" % r], blk.shape
        full[c[0] * n[0]:(c[0] + 1) * n[0], c[1] * n[1]:(c[1] + 1) * n[1], c[2] * n[2]:(c[2] + 1) * n[2]] = blk
    assert np.abs(full - ref).max() < 1e-10 * np.abs(ref).max()


@pytest.mark.parametrize("decomposed", [False, True])
def test_z_split_ttm(decomposed):
    G = gu.load("ttm_2048")
    R, world = _run("ttm_2048", (1, 1, 2), 3, ttm=True, decomposed=decomposed)
    _check_atoms(R, world, G, 3, forces=False)
    ref = G["t2_substeps_done_temp"][1:-1, 1:-1, 1:-1]
    full = np.zeros_like(ref)
    for r in range(world):
        blk = R["r%d_lat_temp" % r][1:-1, 1:-1, 1:-1]
        if not decomposed:
            assert np.abs(blk - ref).max() < 1e-10 * np.abs(ref).max()
            continue
        c, n = R["r%d_coord" % r], blk.shape
        full[c[0] * n[0]:(c[0] + 1) * n[0], c[1] * n[1]:(c[1] + 1) * n[1], c[2] * n[2]:(c[2] + 1) * n[2]] = blk
    if decomposed:
        assert np.abs(full - ref).max() < 1e-10 * np.abs(ref).max()


@pytest.mark.parametrize("args", [("2", "1", "1", "25", "0"), ("2", "2", "2", "25", "0"), ("1", "2", "2", "6", "1")])
def test_c_host_without_python_drives_the_ranks(args):
    """imd_amd/host/dist_host.c: a plain C program (threads as ranks, imdgpu_dist_init_local + imdgpu_dist_step) against its
    own single-rank imdgpu_md_step run of the same atoms: the whole multi-rank loop -- rebuilds included -- needs no host
    layer but the C ABI"""
    src = os.path.join(ROOT, "imd_amd", "host", "dist_host.c")
    lib = os.path.join(ROOT, "imd_amd", "csrc")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "dist_host")
        b = subprocess.run(["gcc", "-O2", "-o", exe, src, "-I" + os.path.join(ROOT, "include"), "-L" + lib, "-limdgpu", "-lpthread",
                            "-lm", "-Wl,-rpath," + lib], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert b.returncode == 0, b.stdout
        r = subprocess.run([exe] + list(args), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600,
                           env=dict(os.environ, GPU_MAX_HW_QUEUES="32"))
        assert r.returncode == 0, r.stdout[-3000:]
        assert "max |dx|" in r.stdout


@pytest.mark.parametrize("cpu_dim", [(2, 1, 1), (4, 1, 1)])
def test_1d_model_on_several_ranks_against_the_reference_on_two(cpu_dim):
    """imd_ttm.c's 1D model (transfer-matrix laser source, electron pressure) on several ranks: every rank bins ITS atoms, the
    slab sums are added over the ranks (the reference's MPI_Allreduce of natomsglobal / vcom*global / ..., imd_ttm.c:325-347,
    404-407), every rank runs the same lattice launch on the whole row and integrates its atoms.  Expected values: the
    reference ITSELF on two ranks (tests/golden/ttm1d_ep_slab.npz) -- for 2x1x1 the very decomposition of that run."""
    G = gu.load("ttm1d_ep_slab")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "res.npz")
        env = dict(os.environ, GPU_MAX_HW_QUEUES="32", PYTHONPATH=ROOT)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "local_ranks_worker.py"), "ttm1d_ep_slab"] + [str(v) for v in cpu_dim] +
                           ["8", "2", "0", out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        assert r.returncode == 0, "local ranks %s failed:\n%s" % (str(cpu_dim), r.stdout[-4000:])
        R = dict(np.load(out))
    world = int(np.prod(cpu_dim))
    ids, pos, mom = _cat(R, world, "nummer"), _cat(R, world, "ort"), _cat(R, world, "impuls")
    assert len(ids) == len(G["s8_nummer"]) == len(np.unique(ids))
    o, go = np.argsort(ids), np.argsort(G["s8_nummer"])
    box = np.diag(G["box"])
    far = np.abs(G["s8_ort"][go][:, 0] - box[0] / 2) > 1.0          # the pressure force near the reference's rank boundary: owner by last re-binning
    dx = pos[o] - G["s8_ort"][go]; dx[:, 1:] -= np.round(dx[:, 1:] / box[1:]) * box[1:]
    assert np.abs(dx[far]).max() < 1e-7
    assert np.abs(mom[o] - G["s8_impuls"][go])[far].max() < 1e-7 * np.abs(G["s8_impuls"]).max()
    # every rank holds the same row, the reference's after step 8's calc_ttm
    for rk in range(world):
        assert np.array_equal(R["r%d_lat_natomsglobal" % rk], G["t8_substeps_done_natomsglobal"].astype(int))
        for f in ("temp", "md_temp", "dens", "U", "source"):
            ref = G["t8_substeps_done_" + f]
            assert np.abs(R["r%d_lat_%s" % (rk, f)][1:-1] - ref).max() <= 1e-6 * np.abs(ref).max(), (rk, f)
        if rk:
            for f in ("temp", "xi", "md_temp"):
                assert np.abs(R["r%d_lat_%s" % (rk, f)] - R["r0_lat_" + f]).max() <= 1e-12 * max(np.abs(R["r0_lat_" + f]).max(), 1e-300), (rk, f)
