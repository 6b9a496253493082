#!/usr/bin/env python3
"""Per-sub-step cost of the decomposed TTM lattice's faces on ONE GPU: 2 (or more) ranks as threads of this process
(imd_amd.parallel.LocalGroup), each with its own context, lattice block and persistent / chained sub-step kernels.
  mode "mailbox": faces through the host shared-memory mailbox, one launch chain per sub-step (round 2's path, TTM_REMOTE=0 in the environment of this tool)
  mode "remote" : ghost records in the neighbours' device memory, written from inside the persistent kernel (round 3)
Prints the chain's duration per MD step and per sub-step (HIP events around the lattice work on its stream) for rank 0.
usage: tools/ttm_decomposed_timing.py [cx cy cz] [steps]      (run once per mode: TTM_REMOTE=0|1 in the environment of this tool -> option ttm_remote)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    cpu_dim = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (2, 1, 1)
    nsteps = int(sys.argv[4]) if len(sys.argv) >= 5 else 40
    from imd_amd import lattice, potential, ttm as ttm_mod
    from imd_amd.parallel import Decomposition, DistributedMD, GpuOps, LocalGroup
    from imd_amd.system import GpuSystem, global_cell_dim
    world = int(np.prod(cpu_dim))
    per = (35, 35, 35)                                  # unit cells per rank: 20 MD cells = 10 FD cells per direction (27 blocks)
    nb = tuple(per[d] * cpu_dim[d] for d in range(3))
    box = lattice.fcc_box(nb)
    pair, rho, emb, r2cut = potential.synthetic_eam_tables()
    cellsz = (np.sqrt(r2cut) + 0.5) ** 2
    gdim = global_cell_dim(box, cellsz, tuple(2 * c for c in cpu_dim))
    pos = lattice.fcc_positions(nb)
    rng = np.random.default_rng(3)
    pos = (pos + rng.normal(scale=0.03, size=pos.shape)) % np.diag(box)
    mom = lattice.maxwell_momenta(len(pos), 26.9815, 0.0258, seed=3)
    ident = np.arange(len(pos), dtype=np.int32)
    group = LocalGroup(world)
    t = np.linalg.inv(box).T

    def body(rank):
        dec = Decomposition(rank, cpu_dim, gdim, box, (1, 1, 1))
        c = np.clip((np.array(dec.gdim) * (pos @ t.T)).astype(int), 0, np.array(dec.gdim) - 1)
        mine = np.all(c // (np.array(dec.dim) - 2) == np.array(dec.coord), axis=1)
        s = GpuSystem(box, (1, 1, 1), dec.dim, 1, cellsz, 0.5, 0.1, device=0, global_cell_dim=dec.gdim, cpu_dim=cpu_dim,
                      my_coord=dec.coord, ttm_decomposed=True, count_ttm_neighbours=True)
        if os.environ.get("TTM_REMOTE") == "0":
            s.set_option("ttm_remote", 0)
        s.upload_tables(pair, rho, emb)
        ttm_mod.attach_default(s, fd_ext=(2, 2, 2), diff_substeps=100, fd_min_atoms=5)
        s.upload_atoms(pos[mine], mom[mine], 26.9815, None, ident[mine])
        md = DistributedMD(dec, GpuOps(s, 0), ttm=True, local_group=group)
        for st in range(10):
            md.md_step(st)
        s.synchronize(); group.barrier()
        s.profile(True); s.profile_reset()
        for st in range(10, 10 + nsteps):
            md.md_step(st)
        s.synchronize()
        n, ms = s.profile_get("ttm_diff")
        n3, ms3 = s.profile_get("pass3")
        out = dict(rank=rank, fd_dim=[int(v) for v in s.fd_dim], remote=s.get_info("ttm_remote"), persist_mode=s.get_info("ttm_persist_mode"),
                   chain_ms_per_step=ms / max(n, 1), per_substep_us=1e3 * ms / max(n, 1) / 100.0, pass3_ms=ms3 / max(n3, 1), natoms=int(mine.sum()))
        group.barrier()
        s.close()
        return out
    res = group.run(body, timeout=600)
    print(json.dumps(dict(cpu_dim=cpu_dim, atoms=len(pos), env_remote=os.environ.get("TTM_REMOTE", "default"), ranks=res)))
    group.close()


if __name__ == "__main__":
    main()
