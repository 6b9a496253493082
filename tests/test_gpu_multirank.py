"""GPU multi-rank path on ONE GPU (run with -m gpu): two processes, each with its own libimdgpu context on
cuda:0, exchange halos over gloo (host-staged; on the 8-GPU node the same code runs over nccl = RCCL).
The distributed trajectory must reproduce the reference's single-domain trajectory."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from tests import golden_util as gu

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, case, cpu_dim, nsteps, ttm, q, decomposed=False, adv=False, backend="gloo"):
    try:
        _worker_body(rank, world, port, case, cpu_dim, nsteps, ttm, q, decomposed, adv, backend)
    except BaseException:
        import traceback
        q.put((rank, "ERROR", traceback.format_exc(), 0, None))


def _worker_body(rank, world, port, case, cpu_dim, nsteps, ttm, q, decomposed=False, adv=False, backend="gloo"):
    """backend gloo: all ranks on cuda:0, host-staged transport; nccl: rank r on cuda:r, the library's RCCL transport"""
    import torch
    import torch.distributed as dist
    from imd_amd.parallel import Decomposition, DistributedMD, GpuOps
    from imd_amd.system import GpuSystem
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dev = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    G = gu.load(case)
    dec = Decomposition(rank, cpu_dim, G["global_cell_dim"], G["box"], G["pbc_dirs"])
    pos, mom, mass, ident = G["s0_ort"], G["s0_impuls"], G["s0_masse"], G["s0_nummer"]
    t = np.linalg.inv(G["box"]).T
    c = np.clip((np.array(dec.gdim) * (pos @ t.T)).astype(int), 0, np.array(dec.gdim) - 1)
    mine = np.all(c // (np.array(dec.dim) - 2) == np.array(dec.coord), axis=1)
    ntypes = int(G["ntypes"])
    s = GpuSystem(G["box"], G["pbc_dirs"], dec.dim, ntypes, float(G["cellsz"]), float(G["nbl_margin"]), float(G["timestep"]),
                  device=dev, global_cell_dim=dec.gdim, cpu_dim=cpu_dim, my_coord=dec.coord, ttm_decomposed=decomposed,
                  external_halo=(world == 1))
    s.upload_tables(gu.table(G, "pair_pot"), gu.table(G, "rho_h_tab"), gu.table(G, "embed_pot"))
    if adv:
        s.set_option("ttm_adv", 1)
    if ttm:
        s.init_ttm(G["ttm_fd_ext"], int(G["ttm_fd_min_atoms"]), int(G["ttm_diff_substeps"]), float(G["ttm_atomic_weight"]),
                   float(G["ttm_atomic_charge"]), dict(x=G["eos_r"], y=G["eos_t"], z=G["eos_cve"]),
                   dict(x=G["eos_r"], y=G["eos_ts"], z=G["eos_ee"]), float(G["ttm_omega_laser"]))
    s.upload_atoms(pos[mine], mom[mine], mass[mine], G["s0_sorte"][mine].astype(np.int32) if ntypes > 1 else None, ident[mine])
    ops = GpuOps(s, dev)
    md = DistributedMD(dec, ops, ttm=ttm)
    if backend == "nccl":
        assert ops.transport == "rccl", ops.transport
    for st in range(nsteps):
        md.md_step(st)
    md.calc_forces()
    sums = md.global_sums()
    d = s.download(fields=("ort", "impuls", "kraft", "nummer", "numneighs"))
    lat = s.ttm_download(fields=("temp", "xi", "natoms", "U", "temp2", "U2", "flux")) if ttm else None
    info = (s.get_info("ttm_remote"), s.get_info("ttm_persist_mode"), s.get_info("ttm_persist_runs")) if ttm else (0, 0, 0)
    q.put((rank, d, sums, md.nrebuild, lat, tuple(dec.coord), info, ops.transport))
    s.close()
    dist.destroy_process_group()


def _run(case, cpu_dim, nsteps, ttm, decomposed=False, adv=False, backend="gloo"):
    world = int(np.prod(cpu_dim))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, cpu_dim, nsteps, ttm, q, decomposed, adv, backend)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in range(world):
            r = q.get(timeout=300)
            assert r[1] != "ERROR", "rank %d failed:\n%s" % (r[0], r[2])
            res.append(r)
    finally:
        for p in procs:
            p.join(timeout=20)
            if p.is_alive():
                p.kill()
    return sorted(res, key=lambda r: r[0])


@pytest.mark.parametrize("cpu_dim", [(2, 1, 1), (1, 2, 1)])
def test_two_ranks_nve(cpu_dim):
    res = _run("eam_nbl_2048", cpu_dim, 20, False)
    G = gu.load("eam_nbl_2048")
    ids = np.concatenate([r[1]["nummer"] for r in res]); pos = np.concatenate([r[1]["ort"] for r in res])
    frc = np.concatenate([r[1]["kraft"] for r in res])
    assert len(np.unique(ids)) == len(G["s20_nummer"]) == len(ids)
    o, go = np.argsort(ids), np.argsort(G["s20_nummer"])
    box = np.diag(G["box"])
    dx = pos[o] - G["s20_ort"][go]; dx -= np.round(dx / box) * box
    assert np.abs(dx).max() < 1e-9
    assert np.abs(frc[o] - G["s20_kraft"][go]).max() < 1e-9 * np.abs(G["s20_kraft"]).max()
    assert abs(res[0][2]["tot_pot_energy"] - float(G["s20_tot_pot_energy"])) < 1e-9 * abs(float(G["s20_tot_pot_energy"]))
    assert all(r[3] >= 2 for r in res)


def test_two_ranks_two_atom_types():
    """Cu3Au lattice, two atom types (4-column tables with per-column ranges): the halo also carries the types of the ghost
    atoms (what = 2) at every rebuild; 20 steps on 2 ranks against the reference's single-domain trajectory"""
    res = _run("eam2_l12_coarse", (2, 1, 1), 20, False)
    G = gu.load("eam2_l12_coarse")
    ids = np.concatenate([r[1]["nummer"] for r in res]); pos = np.concatenate([r[1]["ort"] for r in res])
    frc = np.concatenate([r[1]["kraft"] for r in res])
    o, go = np.argsort(ids), np.argsort(G["s20_nummer"])
    box = np.diag(G["box"])
    dx = pos[o] - G["s20_ort"][go]; dx -= np.round(dx / box) * box
    assert len(ids) == len(G["s20_nummer"]) and np.abs(dx).max() < 1e-9
    assert np.abs(frc[o] - G["s20_kraft"][go]).max() < 1e-9 * np.abs(G["s20_kraft"]).max()
    assert abs(res[0][2]["tot_pot_energy"] - float(G["s20_tot_pot_energy"])) < 1e-9 * abs(float(G["s20_tot_pot_energy"]))


@pytest.mark.parametrize("cpu_dim", [(2, 1, 1), (2, 2, 1)])
def test_ranks_ttm(cpu_dim):
    """2 and 4 ranks (the 2x2x1 layout of bench.py --gpus 4) sharing the one GPU: the replicated lattice, fed by
    all-gathered update_fd blocks, must equal the reference's after the sub-steps."""
    res = _run("ttm_2048", cpu_dim, 3, True)
    G = gu.load("ttm_2048")
    ids = np.concatenate([r[1]["nummer"] for r in res]); pos = np.concatenate([r[1]["ort"] for r in res])
    nn = np.concatenate([r[1]["numneighs"] for r in res])
    o, go = np.argsort(ids), np.argsort(G["s3_nummer"])
    assert np.abs(pos[o] - G["s3_ort"][go]).max() < 1e-9
    assert np.array_equal(nn[o], G["s3_numneighs"][go])
    # replicated lattice: every rank holds the reference's lattice after the sub-steps of step 2
    ref = G["t2_substeps_done_temp"][1:-1, 1:-1, 1:-1]
    for r in res:
        assert np.abs(r[4]["temp"][1:-1, 1:-1, 1:-1] - ref).max() < 1e-10 * np.abs(ref).max()


@pytest.mark.parametrize("cpu_dim", [(2, 1, 1), (2, 2, 1)])
def test_ranks_ttm_decomposed_lattice(cpu_dim):
    """The scalable TTM mode: every rank diffuses only its block of the electron lattice; the ghost faces go through
    the host shared-memory mailbox, written and polled by the sub-step kernels themselves (imdgpu_ttm_mailbox_*).
    The blocks, put together, must equal the reference's lattice after the sub-steps of step 2."""
    res = _run("ttm_2048", cpu_dim, 3, True, decomposed=True)
    G = gu.load("ttm_2048")
    # ranks in different PROCESSES: the ghost-record buffers are opened through hipIpc handles; steps 1 and 2 ran in the
    # persistent kernel with the faces written straight into the neighbours' buffers (step 0: the mailbox)
    assert all(r[6][0] >= 1 and r[6][1] == 2 and r[6][2] >= 1 for r in res), [r[6] for r in res]
    print("ttm_remote (2 = fine-grained memory behind the IPC handle, 1 = coarse), persist mode, runs:", [r[6] for r in res])
    ids = np.concatenate([r[1]["nummer"] for r in res]); pos = np.concatenate([r[1]["ort"] for r in res])
    o, go = np.argsort(ids), np.argsort(G["s3_nummer"])
    assert np.abs(pos[o] - G["s3_ort"][go]).max() < 1e-9
    ref = G["t2_substeps_done_temp"][1:-1, 1:-1, 1:-1]
    full = np.zeros_like(ref)
    for r in res:
        blk = r[4]["temp"][1:-1, 1:-1, 1:-1]
        c, n = r[5], blk.shape
        full[c[0] * n[0]:(c[0] + 1) * n[0], c[1] * n[1]:(c[1] + 1) * n[1], c[2] * n[2]:(c[2] + 1) * n[2]] = blk
    assert np.abs(full - ref).max() < 1e-10 * np.abs(ref).max()


def test_two_ranks_ttm_with_advection():
    """do_ADV with two ranks along x, like the reference run behind tests/golden/ttm_adv.npz: the flux counters travel
    in the all-gathered update_fd blocks, every rank advects its replica of the lattice.  After 49 MD steps (list
    rebuilds with atoms changing FD cells -- and ranks -- at steps 41 and 48) every rank holds the reference's lattice
    of step 48: flux counters exact, temp/U of both lattices within 2e-8 (inverse-EOS tolerance, see test_gpu_ttm)."""
    res = _run("ttm_adv", (2, 1, 1), 49, True, adv=True)
    G = gu.load("ttm_adv")
    inner = (slice(1, -1),) * 3
    assert int(G["t48_flux"].sum()) > 0
    for r in res:
        assert np.array_equal(r[4]["flux"][inner], G["t48_flux"][inner])
        assert np.array_equal(r[4]["natoms"][inner], G["t48_substeps_done_natoms"][inner])
        for f in ("temp", "U", "temp2", "U2", "xi"):
            a, b = r[4][f][inner], G["t48_substeps_done_" + f][inner]
            assert np.abs(a - b).max() <= 2e-8 * np.abs(b).max(), f


def test_two_ranks_ttm_with_advection_on_the_decomposed_lattice():
    """The same run with one lattice block per rank (option ttm_decomposed): do_COMMFLUX's exchange of the boundary
    columns of {flux[8], U} (imd_ttm_3D.c:667-995; dist_commflux inside imdgpu_dist_md_step, no wrap-around at
    the ends of the x range) feeds do_ADV; the per-atom FD cell index is global, so atoms that change ranks are counted
    like in the reference.  Every rank's block equals its part of the reference lattice of step 48."""
    res = _run("ttm_adv", (2, 1, 1), 49, True, decomposed=True, adv=True)
    G = gu.load("ttm_adv")
    nx = (int(G["ttm_fd_dim"][0]) - 2) // 2
    for r in res:
        cx = r[5][0]
        sl = (slice(1 + cx * nx, 1 + (cx + 1) * nx), slice(1, -1), slice(1, -1))
        inner = (slice(1, -1),) * 3
        assert r[4]["temp"].shape[0] == nx + 2
        assert np.array_equal(r[4]["flux"][inner], G["t48_flux"][sl])
        assert np.array_equal(r[4]["natoms"][inner], G["t48_substeps_done_natoms"][sl])
        for f in ("temp", "U", "temp2", "U2", "xi"):
            a, b = r[4][f][inner], G["t48_substeps_done_" + f][sl]
            assert np.abs(a - b).max() <= 2e-8 * np.abs(b).max(), f


def _worker_vacuum(rank, world, port, q):
    try:
        import torch
        import torch.distributed as dist
        from imd_amd import lattice, potential
        from imd_amd.parallel import Decomposition, DistributedMD, GpuOps
        from imd_amd.system import GpuSystem
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        C = _vacuum_case()
        dec = Decomposition(rank, (2, 1, 1), C["gdim"], C["box"], (0, 1, 1))
        mine = (C["pos"][:, 0] >= np.diag(C["box"])[0] / 2) == (rank == 1)
        s = GpuSystem(C["box"], (0, 1, 1), dec.dim, 1, C["cellsz"], 0.5, 0.1, device=0, global_cell_dim=dec.gdim,
                      cpu_dim=(2, 1, 1), my_coord=dec.coord)
        s.upload_tables(*C["tabs"])
        s.upload_atoms(C["pos"][mine], C["mom"][mine], 26.9815, None, np.arange(len(C["pos"]), dtype=np.int32)[mine])
        ops = GpuOps(s, 0)
        md = DistributedMD(dec, ops)
        for st in range(C["nsteps"]):
            md.md_step(st)
        sums = md.global_sums()
        d = s.download(fields=("ort", "impuls", "nummer"))
        q.put((rank, d, sums, int(mine.sum())))
        s.close()
        dist.destroy_process_group()
    except BaseException:
        import traceback
        q.put((rank, "ERROR", traceback.format_exc(), 0))


def _vacuum_case():
    """a slab in the low half of a box twice as long in x (open x): rank 1 of a 2x1x1 grid owns only vacuum"""
    from imd_amd import lattice, potential
    from imd_amd.system import global_cell_dim
    nb = (5, 6, 6)
    pair, rho, emb, r2cut = potential.synthetic_eam_tables()
    cellsz = (np.sqrt(r2cut) + 0.5) ** 2
    box = lattice.fcc_box(nb, 4.05).copy()
    box[0, 0] = 2 * 32.4                                        # slab 2.0 .. 22.3 A, rank boundary at 32.4 A
    pos = lattice.fcc_positions(nb, 4.05) + np.array([2.0, 0.0, 0.0])
    rng = np.random.default_rng(5)
    pos = pos + rng.normal(scale=0.04, size=pos.shape)
    pos[:, 1:] %= np.diag(box)[1:]
    mom = lattice.maxwell_momenta(len(pos), 26.9815, 0.05, seed=5)
    gdim = global_cell_dim(box, cellsz, (2, 1, 1))
    return dict(box=box, pos=pos, mom=mom, gdim=gdim, cellsz=cellsz, tabs=(pair, rho, emb), nsteps=12)


def test_two_ranks_one_in_vacuum():
    """the normal case of an ablation slab decomposed along x -- one rank holds every atom, the other none.
    12 steps of the two-rank run against the CPU oracle's single-domain loop on the same atoms.
    (No mpiexec drop-in twin of this case: the reference's own MPI + NBL build, oracle/_ref/imd_mpi_eam_nve_nbl, dies with
    a segmentation fault on such an input -- pbc 0 1 1, cpu_dim 2 1 1, the second rank all vacuum -- so there is no CPU
    binary to compare a shim run with.)"""
    from oracle import pyoracle as po
    C = _vacuum_case()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_vacuum, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in range(2):
            r = q.get(timeout=300)
            assert r[1] != "ERROR", "rank %d failed:\n%s" % (r[0], r[2])
            res.append(r)
    finally:
        for p in procs:
            p.join(timeout=20)
            if p.is_alive():
                p.kill()
    res = sorted(res, key=lambda r: r[0])
    assert res[1][3] == 0 and res[0][3] == len(C["pos"])          # rank 1 really had nothing
    o = po.OracleSystem([g + 2 for g in C["gdim"]], 64, 1, C["box"], (0, 1, 1), C["cellsz"], 0.5, 0.1)
    for w, t in enumerate(C["tabs"]):
        o.set_table(w, t.as_dict())
    o.fill_from_positions(C["pos"], C["mom"], 26.9815)
    for st in range(C["nsteps"]):
        o.calc_forces(st); o.move_atoms_nve(); o.check_nblist()
    ids = np.concatenate([r[1]["nummer"] for r in res]); pos = np.concatenate([r[1]["ort"] for r in res])
    mom = np.concatenate([r[1]["impuls"] for r in res])
    a, b = np.argsort(ids), np.argsort(o.gather_inner("nummer"))
    assert len(ids) == len(C["pos"])
    dx = pos[a] - o.gather_inner("ort")[b]
    L = np.diag(C["box"]); dx[:, 1:] -= np.round(dx[:, 1:] / L[1:]) * L[1:]
    assert np.abs(dx).max() < 1e-9
    assert np.abs(mom[a] - o.gather_inner("impuls")[b]).max() < 1e-9 * np.abs(mom).max()
