"""The reference's own Makefile configuration measured at scale: a slab of nbx x nby x nby FCC cells (default 10 027 008 atoms),
vacuum on both x sides, imd_ttm.c's 1D model with its transfer-matrix laser source active during the timed steps --
MD-steps/s on one MI355X (imdgpu_md_step with a 1D lattice) and the reference's own 1D program (oracle/_ref/
imd_mpi_eam_ttm1d_tmm_nbl: calc_ttm's sequence with do_tmm in every sub-step, move_atoms_ttm, list maintenance, MPI halo) on
the host cores, on a bounded sample of cells-per-rank^3 unit cells per rank.  One JSON line.
usage: python tools/ttm1d_bench.py [--box nbx nby] [--steps K] [--warmup W] [--cpu-ranks N] [--cpu-cells-per-rank C] [--cpu-steps S] [--no-cpu]"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LASER = dict(I0=2e15, lam=800e-9, sigma_t=60e-15, t0=9e-14, thr=8.0)     # peak about 90 steps in: on during warm-up and timing


def gpu_run(a):
    from imd_amd import lattice, potential, tmm, ttm as ttm_mod
    from imd_amd.system import GpuSystem, global_cell_dim
    al = 4.05
    nbx, nby = a.box
    pair, rho, emb, r2cut = potential.synthetic_eam_tables()
    margin = 0.5
    cellsz = (np.sqrt(r2cut) + margin) ** 2
    vac = 12 * (np.sqrt(r2cut) + margin)
    box = np.diag([nbx * al + 2 * vac, nby * al, nby * al])
    pos = lattice.fcc_positions((nbx, nby, nby), al) + np.array([vac, 0.0, 0.0])
    mom = lattice.maxwell_momenta(len(pos), 26.9815, 0.03, seed=1)
    gd = global_cell_dim(box, cellsz)
    nx = int(round(box[0, 0] / 4.8))
    g = GpuSystem(box, (0, 1, 1), tuple(v + 2 for v in gd), 1, cellsz, margin, 0.1, count_ttm_neighbours=True)
    g.upload_tables(pair, rho, emb)
    g.upload_atoms(pos, mom, 26.9815)
    del pos, mom
    cve, ee = ttm_mod.synthetic_eos()
    g.init_ttm1d(nx, 20, 100, 8, 1, 2700.0, 26.9815, 13.0, cve, ee, ref_ranks=1, adv=True)
    phase = dict(x=cve["x"], y=cve["y"], z=np.ones((len(cve["x"]), len(cve["y"]))))
    g.init_ttm1d_tmm(LASER["I0"], LASER["lam"], LASER["sigma_t"], LASER["t0"], tmm.synthetic_eps_bb(), tmm.synthetic_K12(), phase,
                     tmm_threshold=LASER["thr"])
    step = 0
    for _ in range(a.warmup):
        g.md_step(step); step += 1
    g.synchronize()
    g.profile(True); g.profile_reset()
    nbl0 = g.get_sums()["nbl_count"]
    t0 = time.perf_counter()
    for _ in range(a.steps):
        g.md_step(step); step += 1
    g.synchronize()
    dt = time.perf_counter() - t0
    prof = {k: g.profile_get(k) for k in ("pass1", "pass3", "nblist", "fix_cells", "integrate", "ttm_update_fd", "ttm_diff")}
    st = g.ttm1d_tmm_state()
    lat = g.ttm1d_download()
    natoms = 4 * nbx * nby * nby
    out = {"atoms": natoms, "ttmdimx": nx, "fd_h_x_A": box[0, 0] / nx, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
           "md_steps_per_s": a.steps / dt, "atom_steps_per_s": natoms * a.steps / dt,
           "kernel_ms_per_step": {k: v[1] / a.steps for k, v in prof.items() if v[0]},
           "list_rebuilds_in_timed_region": g.get_sums()["nbl_count"] - nbl0, "substeps": int(lat["scalars"][2]),
           "laser_active": st["laser_active"], "absorption_depth_cells": st["ecut"], "reflectivity": st["refl"],
           "Te_max_eV": float(lat["temp"].max()), "active_cells": int((lat["natoms"] >= 20).sum())}
    g.close()
    return out


def cpu_run(a):
    from imd_amd import potential, tmm, ttm as ttm_mod
    import bench
    exe = os.path.join(ROOT, "oracle", "_ref", "imd_mpi_eam_ttm1d_tmm_nbl")
    mpiexec = bench._find_mpiexec()
    if not os.path.exists(exe) or mpiexec is None:
        return None
    nranks = a.cpu_ranks or min(bench._host_cores(), bench._cpu_quota() or 10 ** 9)
    cpu_dim = bench._rank_grid(nranks)
    nranks = cpu_dim[0] * cpu_dim[1] * cpu_dim[2]
    c = a.cpu_cells_per_rank
    nb = [c * v for v in cpu_dim]
    natoms = 4 * nb[0] * nb[1] * nb[2]
    shift = 80.0
    nx = int(round((nb[0] * 4.05 + 2 * shift) / 4.8 / nranks)) * nranks           # ttmdimx: a multiple of the ranks (imd_ttm.c:969)
    with tempfile.TemporaryDirectory() as top:
        d = os.path.join(top, "run"); os.mkdir(d)
        potential.write_synthetic_eam_files(d)
        cve, ee = ttm_mod.synthetic_eos()
        ttm_mod.write_eos_table(os.path.join(top, "EOS_cve_from_r_te.txt"), cve)
        ttm_mod.write_eos_table(os.path.join(top, "EOS_ee_from_r_tesqrt.txt"), ee)
        ttm_mod.write_eos_table(os.path.join(top, "EOS_phase_from_r_ti.txt"), dict(x=cve["x"], y=cve["y"], z=np.ones((len(cve["x"]), len(cve["y"])))))
        tmm.write_arr(os.path.join(top, "alu_eps_bb.dat"), tmm.synthetic_eps_bb()); tmm.write_arr(os.path.join(top, "K12.dat"), tmm.synthetic_K12())
        open(os.path.join(d, "b.param"), "w").write(
            "coordname _fcc\noutfiles out\nensemble ttm\nmaxsteps %d\ntimestep 0.1\nntypes 1\ntotal_types 1\nmasses 26.9815\ntypes 0\n"
            "box_param %d %d %d\nbox_unit 4.05\npbc_dirs 0 1 1\nstarttemp 0.03\nseed 12345\ncore_potential_file phi.pt\n"
            "embedding_energy_file F.pt\natomic_e-density_file rho.pt\neng_int 100000\ncheckpt_int 100000\nnbl_margin 0.5\nnbl_size 1.5\n"
            "cpu_dim %d %d %d\nshiftx_front %g\nshiftx_rear %g\nmsgbuf_size 3.0\nfd_n_timesteps 100\nfd_min_atoms 20\natomic_weight 26.9815\n"
            "atomic_charge 13\nttm_int 100000\nttmdimx %d\nvlatdim 8\nvlatbuffer 1\nI0 %r\nlambda %r\nlaser_sigma_t %r\nlaser_t_0 %r\ntmm_threshold %r\n"
            % ((a.cpu_steps, nb[0], nb[1], nb[2]) + cpu_dim + (shift, shift, nx, LASER["I0"], LASER["lam"], LASER["sigma_t"], 2e-14, LASER["thr"])))
        env = bench._cpu_env()
        env["IMD_PROBE_ADV"] = "1"; env["IMD_PROBE_TMM"] = "1"        # the probe-driven calc_ttm calls the reference's do_ADV and do_tmm (ref_probe_ttm1d.c)
        t0 = time.time()
        try:
            r = subprocess.run([mpiexec, "-n", str(nranks), exe, "-p", "b.param"], cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                               text=True, timeout=a.cpu_timeout)
        except subprocess.TimeoutExpired:
            return {"value": None, "cores": nranks, "sample": "gave up after %.0f s" % a.cpu_timeout}
        wall = time.time() - t0
    if r.returncode != 0:
        return {"value": None, "cores": nranks, "sample": "reference run failed: " + r.stdout[-400:]}
    secs = None
    for line in r.stdout.splitlines():
        if "seconds excluding setup time" in line:
            secs = float(line.split()[0])
    if not secs:
        return None
    return {"value": natoms * (a.cpu_steps + 1) / secs, "unit": "atom-steps/s", "cores": nranks, "kind": "reference",
            "sample": "reference binary imd_mpi_eam_ttm1d_tmm_nbl (gcc -O2, MPI, cpu_dim %d %d %d), %d atoms (%d per rank) x %d steps, EAM + 1D TTM "
                      "(ttmdimx %d, >= 100 sub-steps, do_ADV, do_tmm in every sub-step from step 2 on), %.1f s main loop, %.1f s wall"
                      % (cpu_dim + (natoms, natoms // nranks, a.cpu_steps + 1, nx, secs, wall))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--box", type=int, nargs=2, default=[272, 96])
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--cpu-ranks", type=int, default=0)
    ap.add_argument("--cpu-cells-per-rank", type=int, default=32)
    ap.add_argument("--cpu-steps", type=int, default=20)
    ap.add_argument("--cpu-timeout", type=float, default=400.0)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    out = {"workload": "FCC Al slab, synthetic tabulated EAM, imd_ttm.c's 1D TTM (-DTTM1D) + transfer-matrix laser source (imd_tmm.c), do_ADV on",
           "gpu": gpu_run(a)}
    if not a.no_cpu:
        out["cpu_reference"] = cpu_run(a)
        if out["cpu_reference"] and out["cpu_reference"].get("value"):
            out["gpu_over_cpu_atom_steps"] = out["gpu"]["atom_steps_per_s"] / out["cpu_reference"]["value"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
