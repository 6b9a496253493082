"""Cost of the 1D two-temperature model's lattice launch (k_ttm1d_calc) with and without the transfer-matrix laser source:
a slab of nbx x nby x nby FCC cells, ttmdimx FD cells along x, vacuum on both sides; ms per calc_ttm and per sub-step.
usage: python tools/ttm1d_tmm_timing.py [nbx nby ttmdimx steps]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from imd_amd import lattice, potential, tmm, ttm as ttm_mod  # noqa: E402
from imd_amd.system import GpuSystem, global_cell_dim  # noqa: E402


def main():
    nbx, nby, nx, steps = (int(v) for v in (sys.argv[1:5] + ["160", "24", "400", "12"][len(sys.argv) - 1:]))
    a = 4.05
    pair, rho, emb, r2cut = potential.synthetic_eam_tables()
    margin = 0.5
    cellsz = (np.sqrt(r2cut) + margin) ** 2
    vac = 60.0
    box = np.diag([nbx * a + 2 * vac, nby * a, nby * a])
    pos = lattice.fcc_positions((nbx, nby, nby), a) + np.array([vac, 0.0, 0.0])
    rng = np.random.default_rng(1)
    pos = pos + rng.normal(scale=0.02, size=pos.shape); pos[:, 1:] %= nby * a
    mom = lattice.maxwell_momenta(len(pos), 26.9815, 0.03, seed=1)
    gd = global_cell_dim(box, cellsz)
    out = {"atoms": len(pos), "ttmdimx": nx, "fd_h_x_A": box[0, 0] / nx}
    for mode in ("no_laser", "tmm"):
        g = GpuSystem(box, (0, 1, 1), tuple(v + 2 for v in gd), 1, cellsz, margin, 0.1, count_ttm_neighbours=True)
        g.upload_tables(pair, rho, emb)
        g.upload_atoms(pos, mom, 26.9815)
        cve, ee = ttm_mod.synthetic_eos()
        g.init_ttm1d(nx, 20, 100, 8, 1, 2700.0, 26.9815, 13.0, cve, ee, ref_ranks=1, adv=True)
        if mode == "tmm":
            phase = dict(x=cve["x"], y=cve["y"], z=np.ones((len(cve["x"]), len(cve["y"]))))
            g.init_ttm1d_tmm(2e15, 800e-9, 50e-15, 10e-15, tmm.synthetic_eps_bb(), tmm.synthetic_K12(), phase, tmm_threshold=8.0)
        for st in range(3):
            g.md_step(st)
        g.synchronize(); t0 = time.perf_counter()
        for st in range(3, 3 + steps):
            g.md_step(st)
        g.synchronize(); ms_step = (time.perf_counter() - t0) / steps * 1e3
        g.calc_forces(3 + steps)
        g.synchronize(); t0 = time.perf_counter()
        for q in range(steps):
            g.ttm1d_calc(3 + steps)
        g.synchronize(); ms_calc = (time.perf_counter() - t0) / steps * 1e3
        nsub = int(g.ttm1d_download()["scalars"][2])
        g.synchronize(); t0 = time.perf_counter()
        for q in range(steps):
            g.ttm1d_stage(3 + steps, 64)                 # the sub-step loop alone (do_tmm, do_DIFF, do_FILLMESH per sub-step)
        g.synchronize(); ms_loop = (time.perf_counter() - t0) / steps * 1e3
        g.synchronize(); t0 = time.perf_counter()
        for q in range(steps):
            g.ttm1d_stage(3 + steps, 1 | 2 | 4 | 8 | 128)   # everything of calc_ttm but the loop (update_fd sweeps, do_ADV's inverse EOS, ...)
        g.synchronize(); ms_rest = (time.perf_counter() - t0) / steps * 1e3
        out[mode] = {"ms_per_md_step": ms_step, "ms_per_calc_ttm_alone": ms_calc, "substeps": nsub, "ms_loop_alone": ms_loop, "ms_rest_alone": ms_rest,
                     "us_per_substep": ms_loop / nsub * 1e3}
        if mode == "tmm":
            st = g.ttm1d_tmm_state()
            out[mode].update(laser_active=st["laser_active"], absorption_depth_cells=st["ecut"], refl=st["refl"], too_short=st["too_short"],
                             active_cells=int(st["active"].sum()), eps_mid=[float(st["eps_re"][nx // 2]), float(st["eps_im"][nx // 2])],
                             I_t=st["I_t"], tmm_time=st["tmm_time"])
        g.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
