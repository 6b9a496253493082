"""N>1 path on CPUs: the product's host-side decomposition / halo / migration logic
(imd_amd.parallel) driven over torch.distributed gloo by 2 and 4 processes, each with a numpy
stand-in device; the distributed trajectory must reproduce the single-domain reference
trajectory (golden fixture from the real reference binary)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from tests import golden_util as gu


def _run_ranks(target, world, args, timeout=None):
    """Spawn `world` workers, collect one result each; on a time-out (scaled with the cores available) or a dead
    worker the remaining ones are killed, the queue is drained and the test FAILS instead of hanging in the exit handler."""
    import queue as _q
    import time
    # one thread per rank: dense numpy / torch thread pools spin in sched_yield when ranks outnumber cores
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
        os.environ[v] = "1"
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    if timeout is None:
        timeout = 240.0 * max(1.0, world / max(ncpu, 1)) + 120.0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + tuple(args) + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    res, deadline, err = [], time.time() + timeout, None
    while len(res) < world:
        try:
            res.append(q.get(timeout=2.0))
        except _q.Empty:
            dead = [p for p in procs if not p.is_alive() and p.exitcode not in (0, None)]
            if dead:
                err = "worker died with exit code %s" % dead[0].exitcode
                break
            if time.time() > deadline:
                err = "no result after %.0f s" % timeout
                break
    for p in procs:
        p.join(timeout=30 if err is None else 0.1)
        if p.is_alive():
            p.kill()
            p.join(timeout=10)
    try:                                   # drain: a worker blocked in q.put must not keep the feeder thread alive
        while True:
            q.get_nowait()
    except Exception:
        pass
    q.close(); q.join_thread()
    if err:
        pytest.fail("multi-rank CPU run (%d ranks): %s" % (world, err))
    return res


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, cpu_dim, nsteps, q):
    import torch
    import torch.distributed as dist
    torch.set_num_threads(1)
    from imd_amd.parallel import Decomposition, DistributedMD
    from tests.numpy_ops import NumpyOps
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    G = gu.load("eam_nbl_2048")
    dec = Decomposition(rank, cpu_dim, G["global_cell_dim"], G["box"], G["pbc_dirs"])
    pos, mom, mass, ident = G["s0_ort"], G["s0_impuls"], G["s0_masse"], G["s0_nummer"]
    t = np.linalg.inv(G["box"]).T
    c = np.clip((np.array(dec.gdim) * (pos @ t.T)).astype(int), 0, np.array(dec.gdim) - 1)
    mine = np.all(c // (np.array(dec.dim) - 2) == np.array(dec.coord), axis=1)
    ops = NumpyOps(dec, gu.table(G, "pair_pot"), gu.table(G, "rho_h_tab"), gu.table(G, "embed_pot"), float(G["cellsz"]),
                   float(G["nbl_margin"]), float(G["timestep"]))
    ops.upload(pos[mine], mom[mine], mass[mine], ident[mine])
    md = DistributedMD(dec, ops)
    for st in range(nsteps):
        md.md_step(st)
    md.calc_forces()
    sums = md.global_sums()
    q.put((rank, ops.id.copy(), ops.pos[:ops.n, :3].copy(), ops.p.copy(), ops.f.copy(), sums, md.nrebuild))
    dist.destroy_process_group()


@pytest.mark.parametrize("cpu_dim", [(2, 1, 1), (2, 2, 1), (2, 2, 2)])      # 2x2x2 is the 8-GPU layout of bench.py
def test_distributed_trajectory_matches_reference(cpu_dim):
    world = int(np.prod(cpu_dim))
    nsteps = 20
    res = _run_ranks(_worker, world, (cpu_dim, nsteps))
    G = gu.load("eam_nbl_2048")
    ids = np.concatenate([r[1] for r in res]); pos = np.concatenate([r[2] for r in res])
    mom = np.concatenate([r[3] for r in res]); frc = np.concatenate([r[4] for r in res])
    assert len(ids) == len(G["s20_nummer"]) and len(np.unique(ids)) == len(ids)        # no atom lost or duplicated
    o, go = np.argsort(ids), np.argsort(G["s20_nummer"])
    box = np.diag(G["box"])
    dx = pos[o] - G["s20_ort"][go]; dx -= np.round(dx / box) * box
    assert np.abs(dx).max() < 1e-9
    assert np.abs(mom[o] - G["s20_impuls"][go]).max() < 1e-9
    assert np.abs(frc[o] - G["s20_kraft"][go]).max() < 1e-9 * np.abs(G["s20_kraft"]).max() + 1e-10
    sums = res[0][5]
    assert abs(sums["tot_pot_energy"] - float(G["s20_tot_pot_energy"])) < 1e-9 * abs(float(G["s20_tot_pot_energy"]))
    assert sums["natoms"] == len(ids)
    assert all(r[6] >= 2 for r in res)          # at least one rebuild with migration took place


class _FakeTtmOps:
    """TTM stand-ins for the numpy device: every rank's update_fd block is a known function of the GLOBAL FD cell index,
    so the host logic under test (block offsets per rank, rank order of the all-gather, fork/join bookkeeping) is right
    exactly when the replicated lattice of every rank equals that function everywhere."""

    def _ttm_setup(self):
        self.calls = []
        self.bdim = tuple(int(v) - 2 for v in self.dec.dim)                      # fd_ext 1 1 1
        self.gfd = tuple(self.bdim[d] * self.dec.cpu_dim[d] for d in range(3))
        self.lat = np.full(self.gfd + (6,), np.nan)

    @staticmethod
    def expected(gfd, step):
        lin = np.arange(int(np.prod(gfd))).reshape(gfd)
        return 1000.0 * step + lin[..., None] + 0.1 * np.arange(6)

    def ttm_block_dims(self): return self.bdim

    def ttm_local_block(self, step):
        import torch
        o = [self.dec.coord[d] * self.bdim[d] for d in range(3)]
        blk = self.expected(self.gfd, step)[o[0]:o[0] + self.bdim[0], o[1]:o[1] + self.bdim[1], o[2]:o[2] + self.bdim[2]]
        return torch.from_numpy(np.ascontiguousarray(blk).reshape(-1))

    def ttm_apply_blocks(self, step, allb, nblocks, offs):
        a = allb.numpy().reshape((nblocks,) + self.bdim + (6,))
        offs = offs.numpy().reshape(nblocks, 3)
        for b in range(nblocks):
            o = offs[b]
            self.lat[o[0]:o[0] + self.bdim[0], o[1]:o[1] + self.bdim[1], o[2]:o[2] + self.bdim[2]] = a[b]

    def ttm_lattice_step(self, step): self.calls.append("lattice_step")


def _worker_ttm(rank, world, port, cpu_dim, q):
    import torch
    import torch.distributed as dist
    torch.set_num_threads(1)
    from imd_amd.parallel import Decomposition, DistributedMD
    from tests.numpy_ops import NumpyOps
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    G = gu.load("eam_nbl_2048")
    dec = Decomposition(rank, cpu_dim, G["global_cell_dim"], G["box"], G["pbc_dirs"])
    pos, mom, mass, ident = G["s0_ort"], G["s0_impuls"], G["s0_masse"], G["s0_nummer"]
    t = np.linalg.inv(G["box"]).T
    c = np.clip((np.array(dec.gdim) * (pos @ t.T)).astype(int), 0, np.array(dec.gdim) - 1)
    mine = np.all(c // (np.array(dec.dim) - 2) == np.array(dec.coord), axis=1)
    Ops = type("Ops", (_FakeTtmOps, NumpyOps), {})
    ops = Ops(dec, gu.table(G, "pair_pot"), gu.table(G, "rho_h_tab"), gu.table(G, "embed_pot"), float(G["cellsz"]),
              float(G["nbl_margin"]), float(G["timestep"]))
    ops._ttm_setup()
    ops.upload(pos[mine], mom[mine], mass[mine], ident[mine])
    md = DistributedMD(dec, ops, ttm=True)
    for st in range(3):
        md.md_step(st)
    q.put((rank, ops.lat.copy(), list(ops.calls), ops.gfd))
    dist.destroy_process_group()


@pytest.mark.parametrize("cpu_dim", [(2, 1, 1), (2, 2, 1)])
def test_ttm_blocks_are_gathered_into_the_replicated_lattice(cpu_dim):
    """The TTM part of the multi-rank step on CPUs (gloo): update_fd block of every rank -> all-gather -> fold into the
    replicated lattice at the rank's FD-cell offset (the offsets DistributedMD hands to imdgpu_dist_set_ttm_blocks) ->
    lattice step."""
    world = int(np.prod(cpu_dim))
    res = _run_ranks(_worker_ttm, world, (cpu_dim,))
    for rank, lat, calls, gfd in res:
        assert np.array_equal(lat, _FakeTtmOps.expected(gfd, 2)), rank            # the lattice of the last step, complete
        assert calls == ["lattice_step"] * 3, calls
